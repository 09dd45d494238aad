/*
 * h2_oracle.c -- CPU restatement (plain C, 4x64-bit Montgomery limbs, pthreads) of the
 * halo2 prover hot path.  TEST INFRASTRUCTURE ONLY: linked/loaded solely by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under halo2_amd/
 * may call into it.
 *
 * Follows, function by function (reference = zcash/halo2, halo2_proofs 0.3.2):
 *   orc_best_multiexp      halo2_proofs/src/arithmetic.rs:143-180  (Buckets :29-112)
 *   orc_best_fft           halo2_proofs/src/arithmetic.rs:192-295
 *   orc_ifft               halo2_proofs/src/poly/domain.rs:375-383
 *   orc_coeff_to_extended  halo2_proofs/src/poly/domain.rs:241-255, :357-373
 *   orc_extended_to_coeff  halo2_proofs/src/poly/domain.rs:303-325
 *   orc_divide_by_vanishing_poly  halo2_proofs/src/poly/domain.rs:329-348
 *   orc_commit             halo2_proofs/src/poly/commitment.rs:119-150
 *   orc_lagrange_basis     halo2_proofs/src/poly/commitment.rs:77-100 (point FFT of Params::new)
 *   orc_generator_collapse halo2_proofs/src/poly/commitment/prover.rs:154-166
 *   orc_fold_scalars       halo2_proofs/src/poly/commitment/prover.rs:128-131
 *   orc_eval_polynomial    halo2_proofs/src/arithmetic.rs:298-303
 *   orc_inner_product      halo2_proofs/src/arithmetic.rs:308-318
 *   orc_kate_division      halo2_proofs/src/arithmetic.rs:322-341
 *   orc_powers             halo2_proofs/src/poly/commitment/prover.rs:90-97
 *   orc_scale_add          halo2_proofs/src/poly/commitment/prover.rs:70 (Polynomial * F, + : poly.rs)
 *   orc_batch_invert       ff 0.13 BatchInvert (Cargo.lock:628), call sites plonk/permutation/prover.rs:118
 *   orc_grand_product      halo2_proofs/src/plonk/permutation/prover.rs:147-153
 * Field and curve arithmetic (pasta_curves 0.5.1, Cargo.lock:1303, not vendored in the
 * reference tree) is restated from the definition: p, q below; y^2 = x^3 + 5;
 * Montgomery form with R = 2^256; Jacobian projective coordinates.
 *
 * Pinning: field layer pinned by the reference's Poseidon KATs and omega constants
 * (tests/test_oracle.py); MSM numeric outputs have no reachable golden vector in the
 * reference tree -> "MSM golden parity unpinned", anchored instead on oracle/pasta.py's
 * naive sum_i [s_i]P_i.
 *
 * Memory layout everywhere: field element = 4 x uint64 little-endian limbs, Montgomery
 * form unless a function says "canonical"; affine point = {x, y} (identity = all zero);
 * Jacobian point = {x, y, z} (identity: z = 0).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

typedef struct {
    u64 p[4];   /* modulus */
    u64 inv;    /* -p^{-1} mod 2^64 */
    u64 r[4];   /* R mod p  (Montgomery one) */
    u64 r2[4];  /* R^2 mod p */
} field_t;

/* constants: SURVEY.md section 8c (computed + verified against the Python oracle) */
static const field_t FIELDS[2] = {
    {/* Fp: Pallas base / Vesta scalar */
     {0x992d30ed00000001ULL, 0x224698fc094cf91bULL, 0, 0x4000000000000000ULL},
     0x992d30ecffffffffULL,
     {0x34786d38fffffffdULL, 0x992c350be41914adULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL},
     {0x8c78ecb30000000fULL, 0xd7d30dbd8b0de0e7ULL, 0x7797a99bc3c95d18ULL, 0x096d41af7b9cb714ULL}},
    {/* Fq: Pallas scalar / Vesta base */
     {0x8c46eb2100000001ULL, 0x224698fc0994a8ddULL, 0, 0x4000000000000000ULL},
     0x8c46eb20ffffffffULL,
     {0x5b2b3e9cfffffffdULL, 0x992c350be3420567ULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL},
     {0xfc9678ff0000000fULL, 0x67bb433d891a16e3ULL, 0x7fae231004ccf590ULL, 0x096d41af7ccfdaa9ULL}},
};

/* curve id 0 = Pallas (base Fp, scalar Fq); 1 = Vesta (base Fq, scalar Fp) */
static const field_t *base_field(int curve) { return &FIELDS[curve ? 1 : 0]; }
static const field_t *scalar_field(int curve) { return &FIELDS[curve ? 0 : 1]; }

static int g_threads = 0;
void orc_set_threads(int t) { g_threads = t; }
int orc_get_threads(void) {
    if (g_threads > 0) return g_threads;
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? (int)n : 1;
}

/* ------------------------------------------------------------------ field */
static inline int geq(const u64 a[4], const u64 b[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static inline void sub_nb(u64 r[4], const u64 a[4], const u64 b[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a[i] - b[i] - (u64)br;
        r[i] = (u64)t;
        br = (t >> 64) & 1;
    }
}
static inline void f_add(const field_t *f, u64 r[4], const u64 a[4], const u64 b[4]) {
    u128 c = 0;
    u64 t[4];
    for (int i = 0; i < 4; i++) {
        c += (u128)a[i] + b[i];
        t[i] = (u64)c;
        c >>= 64;
    }
    /* p < 2^255 so a+b < 2^256: no carry out */
    if (geq(t, f->p)) sub_nb(r, t, f->p); else memcpy(r, t, 32);
}
static inline void f_sub(const field_t *f, u64 r[4], const u64 a[4], const u64 b[4]) {
    u64 t[4];
    if (geq(a, b)) {
        sub_nb(r, a, b);
    } else {
        sub_nb(t, b, a);
        sub_nb(r, f->p, t);
    }
}
static inline void f_neg(const field_t *f, u64 r[4], const u64 a[4]) {
    static const u64 z[4] = {0, 0, 0, 0};
    f_sub(f, r, z, a);
}
static inline void f_dbl(const field_t *f, u64 r[4], const u64 a[4]) { f_add(f, r, a, a); }
static inline int f_is_zero(const u64 a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }
static inline int f_eq(const u64 a[4], const u64 b[4]) {
    return a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3];
}

/* Montgomery product a*b*R^-1 mod p (CIOS) */
static inline void f_mul(const field_t *f, u64 r[4], const u64 a[4], const u64 b[4]) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a[j] * b[i] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (u64)c;
        t[5] = (u64)(c >> 64);
        u64 m = t[0] * f->inv;
        c = (u128)m * f->p[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * f->p[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (u64)c;
        t[4] = t[5] + (u64)(c >> 64);
    }
    if (t[4] || geq(t, f->p)) sub_nb(r, t, f->p); else memcpy(r, t, 32);
}
static inline void f_sqr(const field_t *f, u64 r[4], const u64 a[4]) { f_mul(f, r, a, a); }

static void f_from_mont(const field_t *f, u64 r[4], const u64 a[4]) {
    static const u64 one[4] = {1, 0, 0, 0};
    f_mul(f, r, a, one);
}
static void f_to_mont(const field_t *f, u64 r[4], const u64 a[4]) { f_mul(f, r, a, f->r2); }

static void f_pow(const field_t *f, u64 r[4], const u64 a[4], const u64 e[4]) {
    u64 acc[4], base[4];
    memcpy(acc, f->r, 32);
    memcpy(base, a, 32);
    for (int i = 0; i < 256; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) f_mul(f, acc, acc, base);
        f_sqr(f, base, base);
    }
    memcpy(r, acc, 32);
}
static void f_inv(const field_t *f, u64 r[4], const u64 a[4]) {
    u64 e[4];
    static const u64 two[4] = {2, 0, 0, 0};
    sub_nb(e, f->p, two);
    f_pow(f, r, a, e);
}

/* exported scalar helpers (used by the tests to pin the field layer) */
void orc_f_mul(int field, u64 *r, const u64 *a, const u64 *b) { f_mul(&FIELDS[field], r, a, b); }
void orc_f_add(int field, u64 *r, const u64 *a, const u64 *b) { f_add(&FIELDS[field], r, a, b); }
void orc_f_sub(int field, u64 *r, const u64 *a, const u64 *b) { f_sub(&FIELDS[field], r, a, b); }
void orc_f_inv(int field, u64 *r, const u64 *a) { f_inv(&FIELDS[field], r, a); }
void orc_to_mont(int field, u64 *a, size_t n) {
    for (size_t i = 0; i < n; i++) f_to_mont(&FIELDS[field], a + 4 * i, a + 4 * i);
}
void orc_from_mont(int field, u64 *a, size_t n) {
    for (size_t i = 0; i < n; i++) f_from_mont(&FIELDS[field], a + 4 * i, a + 4 * i);
}

/* ------------------------------------------------------------------ curve */
typedef struct { u64 x[4], y[4]; } aff_t;
typedef struct { u64 x[4], y[4], z[4]; } jac_t;

static inline int aff_is_id(const aff_t *a) { return f_is_zero(a->x) && f_is_zero(a->y); }
static inline int jac_is_id(const jac_t *a) { return f_is_zero(a->z); }
static inline void jac_set_id(jac_t *a) { memset(a, 0, sizeof *a); }
static inline void jac_from_aff(const field_t *f, jac_t *r, const aff_t *a) {
    if (aff_is_id(a)) { jac_set_id(r); return; }
    memcpy(r->x, a->x, 32);
    memcpy(r->y, a->y, 32);
    memcpy(r->z, f->r, 32);
}

static void jac_double(const field_t *f, jac_t *r, const jac_t *p) {
    if (jac_is_id(p)) { jac_set_id(r); return; }
    u64 a[4], b[4], c[4], d[4], e[4], ff[4], t[4], z3[4];
    f_sqr(f, a, p->x);
    f_sqr(f, b, p->y);
    f_sqr(f, c, b);
    f_add(f, d, p->x, b);
    f_sqr(f, d, d);
    f_sub(f, d, d, a);
    f_sub(f, d, d, c);
    f_dbl(f, d, d);
    f_dbl(f, e, a);
    f_add(f, e, e, a);
    f_sqr(f, ff, e);
    f_mul(f, z3, p->y, p->z);
    f_dbl(f, z3, z3);
    f_dbl(f, t, d);
    f_sub(f, r->x, ff, t);
    f_sub(f, t, d, r->x);
    f_mul(f, t, e, t);
    f_dbl(f, c, c);
    f_dbl(f, c, c);
    f_dbl(f, c, c);
    f_sub(f, r->y, t, c);
    memcpy(r->z, z3, 32);
}

static void jac_add(const field_t *f, jac_t *r, const jac_t *p, const jac_t *q) {
    if (jac_is_id(p)) { *r = *q; return; }
    if (jac_is_id(q)) { *r = *p; return; }
    u64 z1z1[4], z2z2[4], u1[4], u2[4], s1[4], s2[4];
    f_sqr(f, z1z1, p->z);
    f_sqr(f, z2z2, q->z);
    f_mul(f, u1, p->x, z2z2);
    f_mul(f, u2, q->x, z1z1);
    f_mul(f, s1, p->y, z2z2);
    f_mul(f, s1, s1, q->z);
    f_mul(f, s2, q->y, z1z1);
    f_mul(f, s2, s2, p->z);
    if (f_eq(u1, u2)) {
        if (f_eq(s1, s2)) jac_double(f, r, p); else jac_set_id(r);
        return;
    }
    u64 h[4], i[4], j[4], rr[4], v[4], t[4], x3[4], y3[4], z3[4];
    f_sub(f, h, u2, u1);
    f_dbl(f, i, h);
    f_sqr(f, i, i);
    f_mul(f, j, h, i);
    f_sub(f, rr, s2, s1);
    f_dbl(f, rr, rr);
    f_mul(f, v, u1, i);
    f_sqr(f, x3, rr);
    f_sub(f, x3, x3, j);
    f_sub(f, x3, x3, v);
    f_sub(f, x3, x3, v);
    f_sub(f, t, v, x3);
    f_mul(f, y3, rr, t);
    f_mul(f, t, s1, j);
    f_dbl(f, t, t);
    f_sub(f, y3, y3, t);
    f_add(f, z3, p->z, q->z);
    f_sqr(f, z3, z3);
    f_sub(f, z3, z3, z1z1);
    f_sub(f, z3, z3, z2z2);
    f_mul(f, z3, z3, h);
    memcpy(r->x, x3, 32);
    memcpy(r->y, y3, 32);
    memcpy(r->z, z3, 32);
}

static void jac_add_mixed(const field_t *f, jac_t *r, const jac_t *p, const aff_t *q) {
    if (aff_is_id(q)) { *r = *p; return; }
    if (jac_is_id(p)) { jac_from_aff(f, r, q); return; }
    u64 z1z1[4], u2[4], s2[4];
    f_sqr(f, z1z1, p->z);
    f_mul(f, u2, q->x, z1z1);
    f_mul(f, s2, q->y, z1z1);
    f_mul(f, s2, s2, p->z);
    if (f_eq(p->x, u2)) {
        if (f_eq(p->y, s2)) jac_double(f, r, p); else jac_set_id(r);
        return;
    }
    u64 h[4], hh[4], i[4], j[4], rr[4], v[4], t[4], x3[4], y3[4], z3[4];
    f_sub(f, h, u2, p->x);
    f_sqr(f, hh, h);
    f_dbl(f, i, hh);
    f_dbl(f, i, i);
    f_mul(f, j, h, i);
    f_sub(f, rr, s2, p->y);
    f_dbl(f, rr, rr);
    f_mul(f, v, p->x, i);
    f_sqr(f, x3, rr);
    f_sub(f, x3, x3, j);
    f_sub(f, x3, x3, v);
    f_sub(f, x3, x3, v);
    f_sub(f, t, v, x3);
    f_mul(f, y3, rr, t);
    f_mul(f, t, p->y, j);
    f_dbl(f, t, t);
    f_sub(f, y3, y3, t);
    f_add(f, z3, p->z, h);
    f_sqr(f, z3, z3);
    f_sub(f, z3, z3, z1z1);
    f_sub(f, z3, z3, hh);
    memcpy(r->x, x3, 32);
    memcpy(r->y, y3, 32);
    memcpy(r->z, z3, 32);
}

static void jac_to_affine(const field_t *f, aff_t *r, const jac_t *p) {
    if (jac_is_id(p)) { memset(r, 0, sizeof *r); return; }
    u64 zi[4], zi2[4], zi3[4];
    f_inv(f, zi, p->z);
    f_sqr(f, zi2, zi);
    f_mul(f, zi3, zi2, zi);
    f_mul(f, r->x, p->x, zi2);
    f_mul(f, r->y, p->y, zi3);
}

/* [k]P with k canonical 4x64; used to build synthetic bases and by tests */
static void jac_mul(const field_t *f, jac_t *r, const aff_t *p, const u64 k[4]) {
    jac_t acc;
    jac_set_id(&acc);
    for (int i = 255; i >= 0; i--) {
        jac_double(f, &acc, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) jac_add_mixed(f, &acc, &acc, p);
    }
    *r = acc;
}

void orc_point_to_affine(int curve, u64 *out_xy, const u64 *in_xyz) {
    jac_to_affine(base_field(curve), (aff_t *)out_xy, (const jac_t *)in_xyz);
}
void orc_point_add(int curve, u64 *out_xyz, const u64 *a_xyz, const u64 *b_xyz) {
    jac_t r;
    jac_add(base_field(curve), &r, (const jac_t *)a_xyz, (const jac_t *)b_xyz);
    memcpy(out_xyz, &r, sizeof r);
}
void orc_point_mul(int curve, u64 *out_xyz, const u64 *p_xy, const u64 *k_canonical) {
    jac_t r;
    jac_mul(base_field(curve), &r, (const aff_t *)p_xy, k_canonical);
    memcpy(out_xyz, &r, sizeof r);
}
int orc_point_on_curve(int curve, const u64 *p_xy) {
    const field_t *f = base_field(curve);
    const aff_t *p = (const aff_t *)p_xy;
    if (aff_is_id(p)) return 1;
    u64 l[4], r[4], five[4] = {5, 0, 0, 0};
    f_to_mont(f, five, five);
    f_sqr(f, l, p->y);
    f_sqr(f, r, p->x);
    f_mul(f, r, r, p->x);
    f_add(f, r, r, five);
    return f_eq(l, r);
}

/* Montgomery batch inversion + normalise: C::Curve::batch_normalize equivalent */
void orc_batch_to_affine(int curve, u64 *out_xy, const u64 *in_xyz, size_t n) {
    for (size_t i = 0; i < n; i++) orc_point_to_affine(curve, out_xy + 8 * i, in_xyz + 12 * i);
}

/* ------------------------------------------------------------------ generic parallel-for */
typedef void (*task_fn)(void *ctx, size_t idx);
typedef struct { task_fn fn; void *ctx; size_t n; size_t next; pthread_mutex_t mu; } pool_t;
static void *pool_worker(void *arg) {
    pool_t *p = (pool_t *)arg;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        size_t i = p->next++;
        pthread_mutex_unlock(&p->mu);
        if (i >= p->n) break;
        p->fn(p->ctx, i);
    }
    return NULL;
}
static void parallel_for(size_t n, task_fn fn, void *ctx) {
    int nt = orc_get_threads();
    if ((size_t)nt > n) nt = (int)n;
    pool_t p = {fn, ctx, n, 0, PTHREAD_MUTEX_INITIALIZER};
    if (nt <= 1) { pool_worker(&p); return; }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nt);
    for (int i = 0; i < nt; i++) pthread_create(&th[i], NULL, pool_worker, &p);
    for (int i = 0; i < nt; i++) pthread_join(th[i], NULL);
    free(th);
}

/* ------------------------------------------------------------------ best_multiexp */
/* arithmetic.rs:146-152 */
int orc_window_bits(size_t n) {
    if (n < 4) return 1;
    if (n < 32) return 3;
    return (int)ceil(log((double)(uint32_t)n));
}

/* Buckets::get_at, arithmetic.rs:95-111; repr = canonical LE bytes */
static inline size_t get_at(int c, size_t segment, const uint8_t repr[32]) {
    size_t skip_bits = segment * (size_t)c;
    size_t skip_bytes = skip_bits / 8;
    if (skip_bytes >= 32) return 0;
    uint8_t v[8] = {0};
    size_t avail = 32 - skip_bytes;
    memcpy(v, repr + skip_bytes, avail < 8 ? avail : 8);
    u64 tmp;
    memcpy(&tmp, v, 8);
    tmp >>= skip_bits - skip_bytes * 8;
    return (size_t)(tmp % ((u64)1 << c));
}

/* Bucket enum, arithmetic.rs:29-58 */
typedef struct { int tag; /* 0 None, 1 Affine, 2 Projective */ aff_t a; jac_t p; } bucket_t;

typedef struct {
    int curve, c;
    const u64 *scalars;
    const u64 *bases;
    size_t n;
    jac_t *window_out;
} msm_ctx;

/* Buckets::sum (arithmetic.rs:74-93) + the c*i doublings of the parallel branch (:163) */
static void msm_window_task(void *vctx, size_t seg) {
    msm_ctx *m = (msm_ctx *)vctx;
    const field_t *bf = base_field(m->curve), *sf = scalar_field(m->curve);
    size_t nb = ((size_t)1 << m->c) - 1;
    bucket_t *bk = (bucket_t *)calloc(nb, sizeof(bucket_t));
    for (size_t i = 0; i < m->n; i++) {
        u64 repr[4];
        f_from_mont(sf, repr, m->scalars + 4 * i);           /* coeff.to_repr(), :77 */
        size_t d = get_at(m->c, seg, (const uint8_t *)repr);
        if (d == 0) continue;
        bucket_t *b = &bk[d - 1];
        const aff_t *base = (const aff_t *)(m->bases + 8 * i);
        if (b->tag == 0) {
            b->tag = 1; b->a = *base;
        } else if (b->tag == 1) {
            jac_t t; jac_from_aff(bf, &t, &b->a);
            jac_add_mixed(bf, &b->p, &t, base);
            b->tag = 2;
        } else {
            jac_add_mixed(bf, &b->p, &b->p, base);
        }
    }
    jac_t acc, sum;
    jac_set_id(&acc);
    jac_set_id(&sum);
    for (size_t k = nb; k-- > 0;) {
        if (bk[k].tag == 1) jac_add_mixed(bf, &sum, &sum, &bk[k].a);
        else if (bk[k].tag == 2) jac_add(bf, &sum, &sum, &bk[k].p);
        jac_add(bf, &acc, &acc, &sum);
    }
    free(bk);
    m->window_out[seg] = acc;
}

/* scalars: n x 4 u64 Montgomery (scalar field of `curve`); bases: n x 8 u64 affine Montgomery;
 * out: Jacobian 12 u64 Montgomery.  Returns 0, or -1 on argument error. */
int orc_best_multiexp(int curve, const u64 *scalars, const u64 *bases, size_t n, u64 *out_xyz) {
    if (curve < 0 || curve > 1) return -1;
    const field_t *bf = base_field(curve);
    int c = orc_window_bits(n);
    size_t nw = 256 / c + 1;
    jac_t *win = (jac_t *)calloc(nw, sizeof(jac_t));
    msm_ctx m = {curve, c, scalars, bases, n, win};
    jac_t total;
    jac_set_id(&total);
    if (n > (size_t)orc_get_threads()) {
        /* parallel branch :156-167: one task per window, doubled c*i times, reduced by add */
        parallel_for(nw, msm_window_task, &m);
        for (size_t i = 0; i < nw; i++) {
            jac_t acc = win[i];
            if (!jac_is_id(&acc))
                for (size_t d = 0; d < (size_t)c * i; d++) jac_double(bf, &acc, &acc);
            jac_add(bf, &total, &total, &acc);
        }
    } else {
        /* serial branch :169-178: Horner from the top window */
        for (size_t i = nw; i-- > 0;) {
            msm_window_task(&m, i);
            for (int d = 0; d < c; d++) jac_double(bf, &total, &total);
            jac_add(bf, &total, &total, &win[i]);
        }
    }
    free(win);
    memcpy(out_xyz, &total, sizeof total);
    return 0;
}

/* Params::commit / commit_lagrange (poly/commitment.rs:119-150): copies poly + blind and
 * g + w into fresh vectors of n+1 entries, then best_multiexp. */
int orc_commit(int curve, const u64 *g, const u64 *w, const u64 *poly, const u64 *blind, size_t n,
               u64 *out_xyz) {
    u64 *s = (u64 *)malloc((n + 1) * 32), *b = (u64 *)malloc((n + 1) * 64);
    memcpy(s, poly, n * 32);
    memcpy(s + 4 * n, blind, 32);
    memcpy(b, g, n * 64);
    memcpy(b + 8 * n, w, 64);
    int rc = orc_best_multiexp(curve, s, b, n + 1, out_xyz);
    free(s);
    free(b);
    return rc;
}

/* ------------------------------------------------------------------ best_fft */
static size_t bitreverse(size_t n, unsigned l) {
    size_t r = 0;
    for (unsigned i = 0; i < l; i++) { r = (r << 1) | (n & 1); n >>= 1; }
    return r;
}
static unsigned log2_floor(size_t num) {
    unsigned pow = 0;
    while (((size_t)1 << (pow + 1)) <= num) pow++;
    return pow;
}

typedef struct {
    const field_t *f; u64 *a; size_t n; size_t twiddle_chunk; const u64 *tw; int depth;
} fft_rec;

/* recursive_butterfly_arithmetic, arithmetic.rs:258-295; `join` = spawn one half on a
 * new thread while depth allows (rayon would steal; same work split). */
static void *fft_rec_run(void *arg);
static void fft_recursive(const field_t *f, u64 *a, size_t n, size_t twiddle_chunk, const u64 *tw, int depth) {
    if (n == 2) {
        u64 t[4];
        memcpy(t, a + 4, 32);
        memcpy(a + 4, a, 32);
        f_add(f, a, a, t);
        f_sub(f, a + 4, a + 4, t);
        return;
    }
    u64 *left = a, *right = a + 4 * (n / 2);
    if (depth > 0) {
        fft_rec l = {f, left, n / 2, twiddle_chunk * 2, tw, depth - 1};
        pthread_t th;
        pthread_create(&th, NULL, fft_rec_run, &l);
        fft_recursive(f, right, n / 2, twiddle_chunk * 2, tw, depth - 1);
        pthread_join(th, NULL);
    } else {
        fft_recursive(f, left, n / 2, twiddle_chunk * 2, tw, 0);
        fft_recursive(f, right, n / 2, twiddle_chunk * 2, tw, 0);
    }
    u64 t[4];
    /* twiddle factor one */
    memcpy(t, right, 32);
    memcpy(right, left, 32);
    f_add(f, left, left, t);
    f_sub(f, right, right, t);
    for (size_t i = 1; i < n / 2; i++) {
        u64 *x = left + 4 * i, *y = right + 4 * i;
        f_mul(f, t, y, tw + 4 * (i * twiddle_chunk));
        memcpy(y, x, 32);
        f_add(f, x, x, t);
        f_sub(f, y, y, t);
    }
}
static void *fft_rec_run(void *arg) {
    fft_rec *r = (fft_rec *)arg;
    fft_recursive(r->f, r->a, r->n, r->twiddle_chunk, r->tw, r->depth);
    return NULL;
}

/* a: n x 4 u64 Montgomery, in place; omega Montgomery.  Returns 0 / -1. */
int orc_best_fft(int field, u64 *a, const u64 *omega, unsigned log_n) {
    if (field < 0 || field > 1 || log_n > 32) return -1;
    const field_t *f = &FIELDS[field];
    size_t n = (size_t)1 << log_n;
    unsigned log_threads = log2_floor((size_t)orc_get_threads());
    for (size_t k = 0; k < n; k++) {                        /* :207-212 */
        size_t rk = bitreverse(k, log_n);
        if (k < rk) {
            u64 t[4];
            memcpy(t, a + 4 * rk, 32);
            memcpy(a + 4 * rk, a + 4 * k, 32);
            memcpy(a + 4 * k, t, 32);
        }
    }
    size_t half = n / 2;
    u64 *tw = (u64 *)malloc((half ? half : 1) * 32);       /* :215-221 */
    u64 w[4];
    memcpy(w, f->r, 32);
    for (size_t j = 0; j < half; j++) {
        memcpy(tw + 4 * j, w, 32);
        f_mul(f, w, w, omega);
    }
    if (log_n <= log_threads) {                             /* :223-251 */
        size_t chunk = 2, twiddle_chunk = n / 2;
        for (unsigned s = 0; s < log_n; s++) {
            for (size_t st = 0; st < n; st += chunk) {
                u64 *left = a + 4 * st, *right = a + 4 * (st + chunk / 2);
                u64 t[4];
                memcpy(t, right, 32);
                memcpy(right, left, 32);
                f_add(f, left, left, t);
                f_sub(f, right, right, t);
                for (size_t i = 1; i < chunk / 2; i++) {
                    u64 *x = left + 4 * i, *y = right + 4 * i;
                    f_mul(f, t, y, tw + 4 * (i * twiddle_chunk));
                    memcpy(y, x, 32);
                    f_add(f, x, x, t);
                    f_sub(f, y, y, t);
                }
            }
            chunk *= 2;
            twiddle_chunk /= 2;
        }
    } else if (n >= 2) {
        fft_recursive(f, a, n, 1, tw, (int)log_threads);    /* :253 */
    }
    free(tw);
    return 0;
}

/* parallelize (arithmetic.rs:345-362): chunked elementwise multiply helpers */
typedef struct { const field_t *f; u64 *a; size_t n, chunk; const u64 *k; size_t nk; int mode; } ew_ctx;
static void ew_task(void *vctx, size_t ci) {
    ew_ctx *e = (ew_ctx *)vctx;
    size_t lo = ci * e->chunk, hi = lo + e->chunk;
    if (hi > e->n) hi = e->n;
    for (size_t i = lo; i < hi; i++) {
        if (e->mode == 0) {
            f_mul(e->f, e->a + 4 * i, e->a + 4 * i, e->k);                       /* scale */
        } else if (e->mode == 1) {
            size_t r = i % 3;                                                     /* zeta powers */
            if (r) f_mul(e->f, e->a + 4 * i, e->a + 4 * i, e->k + 4 * (r - 1));
        } else {
            f_mul(e->f, e->a + 4 * i, e->a + 4 * i, e->k + 4 * (i % e->nk));     /* periodic table */
        }
    }
}
static void elementwise(const field_t *f, u64 *a, size_t n, const u64 *k, size_t nk, int mode) {
    size_t nt = (size_t)orc_get_threads();
    size_t chunk = n / nt;
    if (chunk < nt) chunk = n ? n : 1;
    ew_ctx e = {f, a, n, chunk, k, nk, mode};
    parallel_for((n + chunk - 1) / chunk, ew_task, &e);
}

/* EvaluationDomain::ifft, domain.rs:375-383 */
int orc_ifft(int field, u64 *a, const u64 *omega_inv, unsigned log_n, const u64 *divisor) {
    int rc = orc_best_fft(field, a, omega_inv, log_n);
    if (rc) return rc;
    elementwise(&FIELDS[field], a, (size_t)1 << log_n, divisor, 1, 0);
    return 0;
}
/* distribute_powers_zeta, domain.rs:357-373; coset_powers = {first, second} */
void orc_distribute_powers_zeta(int field, u64 *a, size_t n, const u64 *coset_powers2) {
    elementwise(&FIELDS[field], a, n, coset_powers2, 2, 1);
}
/* coeff_to_extended, domain.rs:241-255. a_ext has 2^ext_k entries; first 2^k hold the input. */
int orc_coeff_to_extended(int field, u64 *a_ext, unsigned k, unsigned ext_k, const u64 *g_coset,
                          const u64 *g_coset_inv, const u64 *extended_omega) {
    u64 cp[8];
    memcpy(cp, g_coset, 32);
    memcpy(cp + 4, g_coset_inv, 32);
    orc_distribute_powers_zeta(field, a_ext, (size_t)1 << k, cp);
    memset(a_ext + 4 * ((size_t)1 << k), 0, 32 * (((size_t)1 << ext_k) - ((size_t)1 << k)));
    return orc_best_fft(field, a_ext, extended_omega, ext_k);
}
/* extended_to_coeff, domain.rs:303-325 (caller truncates) */
int orc_extended_to_coeff(int field, u64 *a_ext, unsigned ext_k, const u64 *g_coset, const u64 *g_coset_inv,
                          const u64 *extended_omega_inv, const u64 *extended_ifft_divisor) {
    int rc = orc_ifft(field, a_ext, extended_omega_inv, ext_k, extended_ifft_divisor);
    if (rc) return rc;
    u64 cp[8];
    memcpy(cp, g_coset_inv, 32);
    memcpy(cp + 4, g_coset, 32);
    orc_distribute_powers_zeta(field, a_ext, (size_t)1 << ext_k, cp);
    return 0;
}
/* divide_by_vanishing_poly, domain.rs:329-348 */
void orc_divide_by_vanishing_poly(int field, u64 *a_ext, unsigned ext_k, const u64 *t_evals, size_t nt) {
    elementwise(&FIELDS[field], a_ext, (size_t)1 << ext_k, t_evals, nt, 2);
}

/* ------------------------------------------------------------------ synthetic inputs */
/* SplitMix64 -> 512 bits -> mod p (mirrors oracle/pasta.py SplitMix64.field), output Montgomery. */
static u64 sm64(u64 *s) {
    u64 z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static void field_from_512(const field_t *f, u64 out[4], const u64 w[8]) {
    /* value = lo + hi*2^256; Montgomery form = lo*R + hi*R^2:  mont(lo)=lo*R2*R^-1, hi*R2 -> *R2 again */
    u64 lo[4], hi[4], t[4];
    memcpy(lo, w, 32);
    memcpy(hi, w + 4, 32);
    /* reduce lo, hi below p first (they are < 2^256 < 4p+...) via Montgomery identities */
    f_mul(f, lo, lo, f->r2);          /* lo*R            */
    f_mul(f, t, hi, f->r2);           /* hi*R            */
    f_mul(f, t, t, f->r2);            /* hi*R*R = hi*2^256 in Montgomery form */
    f_add(f, out, lo, t);
}
void orc_random_field(int field, u64 seed, u64 *out, size_t n) {
    const field_t *f = &FIELDS[field];
    u64 s = seed;
    for (size_t i = 0; i < n; i++) {
        u64 w[8];
        for (int k = 0; k < 8; k++) w[k] = sm64(&s);
        field_from_512(f, out + 4 * i, w);
    }
}

/* n distinct-looking affine bases: P_0 = [r0]G, P_{i+1} = P_i + D  with D = [r1]G; chain of mixed adds,
 * normalised in blocks.  G supplied by caller (a pinned on-curve point).  Cheap (1 add/point), and the
 * points are in general position w.r.t. a random scalar vector. */
typedef struct { int curve; const aff_t *g; u64 seed; u64 *out; size_t n, blk; } gen_ctx;
static void gen_task(void *vctx, size_t bi) {
    gen_ctx *g = (gen_ctx *)vctx;
    const field_t *bf = base_field(g->curve), *sf = scalar_field(g->curve);
    size_t lo = bi * g->blk, hi = lo + g->blk;
    if (hi > g->n) hi = g->n;
    u64 s = g->seed + 0x1000003ULL * (bi + 1), k[8], r0[4], r1[4];
    for (int i = 0; i < 8; i++) k[i] = sm64(&s);
    field_from_512(sf, r0, k);
    f_from_mont(sf, r0, r0);
    for (int i = 0; i < 8; i++) k[i] = sm64(&s);
    field_from_512(sf, r1, k);
    f_from_mont(sf, r1, r1);
    jac_t cur, d;
    aff_t da;
    jac_mul(bf, &cur, g->g, r0);
    jac_mul(bf, &d, g->g, r1);
    jac_to_affine(bf, &da, &d);
    /* batch-normalise the block with Montgomery's trick */
    size_t m = hi - lo;
    jac_t *pts = (jac_t *)malloc(m * sizeof(jac_t));
    u64 *pre = (u64 *)malloc(m * 32);
    for (size_t i = 0; i < m; i++) {
        pts[i] = cur;
        jac_add_mixed(bf, &cur, &cur, &da);
    }
    u64 acc[4];
    memcpy(acc, bf->r, 32);
    for (size_t i = 0; i < m; i++) {
        memcpy(pre + 4 * i, acc, 32);
        if (!jac_is_id(&pts[i])) f_mul(bf, acc, acc, pts[i].z);
    }
    f_inv(bf, acc, acc);
    for (size_t i = m; i-- > 0;) {
        aff_t *o = (aff_t *)(g->out + 8 * (lo + i));
        if (jac_is_id(&pts[i])) { memset(o, 0, sizeof *o); continue; }
        u64 zi[4], zi2[4];
        f_mul(bf, zi, acc, pre + 4 * i);
        f_mul(bf, acc, acc, pts[i].z);
        f_sqr(bf, zi2, zi);
        f_mul(bf, o->x, pts[i].x, zi2);
        f_mul(bf, zi2, zi2, zi);
        f_mul(bf, o->y, pts[i].y, zi2);
    }
    free(pts);
    free(pre);
}
void orc_generate_bases(int curve, const u64 *g_xy, u64 seed, u64 *out_xy, size_t n) {
    gen_ctx g = {curve, (const aff_t *)g_xy, seed, out_xy, n, 4096};
    parallel_for((n + g.blk - 1) / g.blk, gen_task, &g);
}

/* naive sum_i [s_i]P_i : the definition used by the reference's test_multiexp (arithmetic.rs:448-455) */
int orc_msm_naive(int curve, const u64 *scalars, const u64 *bases, size_t n, u64 *out_xyz) {
    const field_t *bf = base_field(curve), *sf = scalar_field(curve);
    jac_t acc;
    jac_set_id(&acc);
    for (size_t i = 0; i < n; i++) {
        u64 k[4];
        jac_t t;
        f_from_mont(sf, k, scalars + 4 * i);
        jac_mul(bf, &t, (const aff_t *)(bases + 8 * i), k);
        jac_add(bf, &acc, &acc, &t);
    }
    memcpy(out_xyz, &acc, sizeof acc);
    return 0;
}

/* ------------------------------------------------------------------ IPA round pieces */
/* parallel_generator_collapse, poly/commitment/prover.rs:154-166: g_lo[i] = g_lo[i] + g_hi[i] * challenge, normalised.
 * g: 2*half affine points (Montgomery); challenge: scalar-field element (Montgomery). */
typedef struct { int curve; u64 *g; size_t half; u64 k[4]; } collapse_ctx;
static void collapse_task(void *vctx, size_t i) {
    collapse_ctx *c = (collapse_ctx *)vctx;
    const field_t *bf = base_field(c->curve);
    jac_t t, lo;
    jac_mul(bf, &t, (const aff_t *)(c->g + 8 * (c->half + i)), c->k);
    jac_from_aff(bf, &lo, (const aff_t *)(c->g + 8 * i));
    jac_add(bf, &t, &lo, &t);
    jac_to_affine(bf, (aff_t *)(c->g + 8 * i), &t);
}
void orc_generator_collapse(int curve, u64 *g, size_t half, const u64 *challenge) {
    collapse_ctx c = {curve, g, half, {0, 0, 0, 0}};
    f_from_mont(scalar_field(curve), c.k, challenge);
    parallel_for(half, collapse_task, &c);
}
/* prover.rs:128-131: a[i] = a[i] + a[i + half] * factor */
void orc_fold_scalars(int field, u64 *a, size_t half, const u64 *factor) {
    const field_t *f = &FIELDS[field];
    for (size_t i = 0; i < half; i++) {
        u64 t[4];
        f_mul(f, t, a + 4 * (half + i), factor);
        f_add(f, a + 4 * i, a + 4 * i, t);
    }
}

/* ------------------------------------------------------------------ polynomial helpers (sequential, as written) */
/* arithmetic.rs:298-303: fold from the top coefficient, acc = acc * point + coeff */
void orc_eval_polynomial(int field, const u64 *poly, size_t n, const u64 *point, u64 *out) {
    const field_t *f = &FIELDS[field];
    u64 acc[4] = {0, 0, 0, 0};
    for (size_t i = n; i-- > 0;) {
        f_mul(f, acc, acc, point);
        f_add(f, acc, acc, poly + 4 * i);
    }
    memcpy(out, acc, 32);
}
/* arithmetic.rs:308-318 */
void orc_inner_product(int field, const u64 *a, const u64 *b, size_t n, u64 *out) {
    const field_t *f = &FIELDS[field];
    u64 acc[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        u64 t[4];
        f_mul(f, t, a + 4 * i, b + 4 * i);
        f_add(f, acc, acc, t);
    }
    memcpy(out, acc, 32);
}
/* arithmetic.rs:322-341: b = -b; from the top: lead = a_i - tmp; q = lead; tmp = lead * b.  q has n - 1 entries */
void orc_kate_division(int field, const u64 *a, size_t n, const u64 *point, u64 *q) {
    const field_t *f = &FIELDS[field];
    u64 nb[4], tmp[4] = {0, 0, 0, 0};
    f_neg(f, nb, point);
    for (size_t i = n; i-- > 1;) {               /* q.iter_mut().rev().zip(a.rev()): q[i-1] pairs with a[i] */
        u64 lead[4];
        f_sub(f, lead, a + 4 * i, tmp);
        memcpy(q + 4 * (i - 1), lead, 32);
        f_mul(f, tmp, lead, nb);
    }
}
/* prover.rs:90-97: cur = 1; push(cur); cur *= x */
void orc_powers(int field, const u64 *x, size_t n, u64 *out) {
    const field_t *f = &FIELDS[field];
    u64 cur[4];
    memcpy(cur, f->r, 32);
    for (size_t i = 0; i < n; i++) {
        memcpy(out + 4 * i, cur, 32);
        f_mul(f, cur, cur, x);
    }
}
/* prover.rs:70: s_poly * xi + p_poly, coefficient-wise */
void orc_scale_add(int field, u64 *a, const u64 *x, const u64 *b, size_t n) {
    const field_t *f = &FIELDS[field];
    for (size_t i = 0; i < n; i++) {
        f_mul(f, a + 4 * i, a + 4 * i, x);
        f_add(f, a + 4 * i, a + 4 * i, b + 4 * i);
    }
}
/* ff::BatchInvert: running product over the non-zero entries, one inversion, walk back; zeros untouched */
void orc_batch_invert(int field, u64 *a, size_t n) {
    const field_t *f = &FIELDS[field];
    u64 *pre = (u64 *)malloc((n ? n : 1) * 32);
    u64 acc[4];
    memcpy(acc, f->r, 32);
    for (size_t i = 0; i < n; i++) {
        memcpy(pre + 4 * i, acc, 32);
        if (!f_is_zero(a + 4 * i)) f_mul(f, acc, acc, a + 4 * i);
    }
    f_inv(f, acc, acc);
    for (size_t i = n; i-- > 0;) {
        if (f_is_zero(a + 4 * i)) continue;
        u64 t[4];
        f_mul(f, t, acc, pre + 4 * i);
        f_mul(f, acc, acc, a + 4 * i);
        memcpy(a + 4 * i, t, 32);
    }
    free(pre);
}
/* permutation/prover.rs:147-153: z = [init]; for row in 1..n: z.push(z[row-1] * m[row-1]) */
void orc_grand_product(int field, const u64 *m, size_t n, const u64 *init, u64 *z) {
    const field_t *f = &FIELDS[field];
    if (!n) return;
    memcpy(z, init, 32);
    for (size_t row = 1; row < n; row++) f_mul(f, z + 4 * row, z + 4 * (row - 1), m + 4 * (row - 1));
}

/* ------------------------------------------------------------------ Params::new: Lagrange basis */
/* poly/commitment.rs:77-100: best_fft over curve points with alpha_inv = ROOT_OF_UNITY_INV^(2^(S-k)), then every point
 * times 2^-k, then normalise.  best_fft's generic G = C::Curve instance (arithmetic.rs:192-255, iterative form). */
typedef struct { int curve; jac_t *a; size_t n; size_t chunk, twiddle_chunk; const u64 *tw_canon; } ecfft_ctx;
static void ecfft_block_task(void *vctx, size_t blk) {
    ecfft_ctx *c = (ecfft_ctx *)vctx;
    const field_t *bf = base_field(c->curve);
    jac_t *left = c->a + blk * c->chunk, *right = left + c->chunk / 2;
    for (size_t i = 0; i < c->chunk / 2; i++) {
        jac_t t = right[i], neg;
        if (i != 0) {                                   /* twiddle factor one for i == 0 (:232-238) */
            aff_t ta;
            jac_to_affine(bf, &ta, &right[i]);
            jac_mul(bf, &t, &ta, c->tw_canon + 4 * (i * c->twiddle_chunk));
        }
        neg = t;
        f_neg(bf, neg.y, t.y);
        right[i] = left[i];
        jac_add(bf, &left[i], &left[i], &t);
        jac_add(bf, &right[i], &right[i], &neg);
    }
}
int orc_lagrange_basis(int curve, const u64 *g, unsigned k, u64 *out) {
    if (k >= 32) return -1;
    const field_t *bf = base_field(curve), *sf = scalar_field(curve);
    size_t n = (size_t)1 << k;
    /* alpha_inv */
    u64 five[4] = {5, 0, 0, 0}, e[4], root[4], inv[4];
    f_to_mont(sf, five, five);
    e[0] = (sf->p[0] >> 32) | (sf->p[1] << 32); e[1] = (sf->p[1] >> 32) | (sf->p[2] << 32);
    e[2] = (sf->p[2] >> 32) | (sf->p[3] << 32); e[3] = sf->p[3] >> 32;     /* (p - 1) / 2^32 */
    f_pow(sf, root, five, e);
    for (unsigned i = k; i < 32; i++) f_sqr(sf, root, root);
    f_inv(sf, inv, root);
    jac_t *a = (jac_t *)malloc(n * sizeof(jac_t));
    for (size_t i = 0; i < n; i++) jac_from_aff(bf, &a[i], (const aff_t *)(g + 8 * i));
    for (size_t x = 0; x < n; x++) {                                       /* bit reversal :207-212 */
        size_t rx = bitreverse(x, k);
        if (x < rx) { jac_t t = a[x]; a[x] = a[rx]; a[rx] = t; }
    }
    size_t half = n / 2;
    u64 *tw = (u64 *)malloc((half ? half : 1) * 32), w[4];
    memcpy(w, sf->r, 32);
    for (size_t j = 0; j < half; j++) { f_from_mont(sf, tw + 4 * j, w); f_mul(sf, w, w, inv); }
    ecfft_ctx c = {curve, a, n, 2, n / 2, tw};
    for (unsigned s = 0; s < k; s++) {
        parallel_for(n / c.chunk, ecfft_block_task, &c);
        c.chunk *= 2;
        c.twiddle_chunk /= 2;
    }
    /* g *= TWO_INV^k (:83-88), normalise (:90-100) */
    u64 two[4] = {2, 0, 0, 0}, minv[4], minv_c[4], acc[4];
    f_to_mont(sf, two, two);
    f_inv(sf, two, two);
    memcpy(acc, sf->r, 32);
    for (unsigned i = 0; i < k; i++) f_mul(sf, acc, acc, two);
    memcpy(minv, acc, 32);
    f_from_mont(sf, minv_c, minv);
    for (size_t i = 0; i < n; i++) {
        aff_t p;
        jac_t r;
        jac_to_affine(bf, &p, &a[i]);
        jac_mul(bf, &r, &p, minv_c);
        jac_to_affine(bf, (aff_t *)(out + 8 * i), &r);
    }
    free(tw);
    free(a);
    return 0;
}
