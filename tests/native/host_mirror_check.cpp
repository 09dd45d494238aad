// C++ parity test of the host mirror (halo2_amd/host/halo2_host.hpp) through the C ABI, against the C oracle.
// Written the way the reference's own tests read (arithmetic.rs:440 test_multiexp, commitment.rs:258
// test_commit_lagrange, domain.rs iFFT checks).  TEST CODE: links oracle/h2_oracle.c as the checker.
// Build: g++ -O2 -std=c++17 tests/native/host_mirror_check.cpp -Lhalo2_amd -lhalo2_mi355x -Loracle -lh2oracle (see __graft_entry__.build)
#include <chrono>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <vector>

#include "../../halo2_amd/host/halo2_host.hpp"

extern "C" {
void orc_random_field(int field, uint64_t seed, uint64_t *out, size_t n);
void orc_generate_bases(int curve, const uint64_t *g_xy, uint64_t seed, uint64_t *out_xy, size_t n);
int orc_best_multiexp(int curve, const uint64_t *scalars, const uint64_t *bases, size_t n, uint64_t *out_xyz);
int orc_commit(int curve, const uint64_t *g, const uint64_t *w, const uint64_t *poly, const uint64_t *blind, size_t n, uint64_t *out_xyz);
void orc_point_to_affine(int curve, uint64_t *out_xy, const uint64_t *in_xyz);
int orc_best_fft(int field, uint64_t *a, const uint64_t *omega, unsigned log_n);
int orc_ifft(int field, uint64_t *a, const uint64_t *omega_inv, unsigned log_n, const uint64_t *divisor);
int orc_coeff_to_extended(int field, uint64_t *a_ext, unsigned k, unsigned ext_k, const uint64_t *g_coset, const uint64_t *g_coset_inv, const uint64_t *extended_omega);
void orc_to_mont(int field, uint64_t *a, size_t n);
void orc_eval_polynomial(int field, const uint64_t *poly, size_t n, const uint64_t *point, uint64_t *out);
void orc_inner_product(int field, const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out);
void orc_kate_division(int field, const uint64_t *a, size_t n, const uint64_t *point, uint64_t *q);
void orc_divide_by_vanishing_poly(int field, uint64_t *a_ext, unsigned ext_k, const uint64_t *t_evals, size_t nt);
}
using namespace halo2;

static int fails = 0;
#define EXPECT(cond, msg) do { if (!(cond)) { printf("FAIL: %s\n", msg); fails++; } else printf("ok: %s\n", msg); } while (0)

static bool same_point(int curve, const Jacobian &a, const uint64_t *b_xyz) {
    uint64_t x[8], y[8];
    orc_point_to_affine(curve, x, a.data());
    orc_point_to_affine(curve, y, b_xyz);
    return memcmp(x, y, 64) == 0;
}

// `host_mirror_check opening <k>`: the reference's test_opening_proof (poly/commitment.rs:305-379) through the C++ mirror -- Params::new(k),
// a_i = i, a counter-driven rng -- printing the transcript bytes in hex; tests/test_gpu_opening.py compares them with the Python
// mirror's for the same inputs (which is in turn pinned to the sequential restatement of the reference prover).
static int opening_mode(uint32_t k) {
    constexpr int CURVE = H2_VESTA, SF = H2_FP;
    Params<CURVE> params = Params<CURVE>::new_params(k);
    std::vector<Fe> px(params.n);
    for (size_t i = 0; i < params.n; i++) px[i] = field::from_u64(SF, i);
    const Blind<CURVE> blind{field::from_u64(SF, 7)};
    uint64_t ctr = 0;
    auto rng = [&]() { ++ctr; return field::from_u64(SF, ctr * 0x9E3779B97F4A7C15ULL + 1); };
    Blake2bWrite<CURVE> tr;
    tr.write_point(to_affine<CURVE>(params.commit(px, blind)));
    const Fe x = tr.squeeze_challenge_scalar();
    tr.write_scalar(eval_polynomial<SF>(px, x));
    create_proof<CURVE>(params, rng, tr, px, blind, x);
    // the step-by-step form (host-pointer calls around h2_ipa_rounds) must write the same bytes as the one-call form (h2_open)
    ctr = 0;
    Blake2bWrite<CURVE> tr2;
    tr2.write_point(to_affine<CURVE>(params.commit(px, blind)));
    const Fe x2 = tr2.squeeze_challenge_scalar();
    tr2.write_scalar(eval_polynomial<SF>(px, x2));
    create_proof_stepwise<CURVE>(params, rng, tr2, px, blind, x2);
    const std::vector<uint8_t> bytes = tr.finalize();
    if (bytes != tr2.finalize()) { printf("FAIL: h2_open and the step-by-step argument wrote different bytes\n"); return 1; }
    for (uint8_t b : bytes) printf("%02x", b);
    printf("\n");
    return 0;
}

// `host_mirror_check opening-time <k> [reps] [stepwise]`: the same opening argument timed natively (no Python in the process: a fresh GPU
// box is measuring within a second) -- Params::new(k) once, then `reps` arguments over the same polynomial; prints the wall time of each and
// whether every repetition wrote the same bytes.  The n + 1 + 2k random scalars are drawn into a pool BEFORE the clock starts (the rng
// is the caller's in the reference too; this counter rng costs ~35 ns a draw, 37 ms at k = 20, which round 4's 68-71 ms included): the
// timed region is create_proof with host vectors in and the transcript out.  `stepwise`: the form before h2_open.  Not run by the test suite.
static int opening_time_mode(uint32_t k, int reps, bool stepwise) {
    constexpr int CURVE = H2_VESTA, SF = H2_FP;
    const auto t_p = std::chrono::steady_clock::now();
    Params<CURVE> params = Params<CURVE>::new_params(k);
    printf("Params::new(%u): %.1f ms\n", k, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_p).count());
    std::vector<Fe> px(params.n);
    for (size_t i = 0; i < params.n; i++) px[i] = field::from_u64(SF, i);
    const Blind<CURVE> blind{field::from_u64(SF, 7)};
    std::vector<uint8_t> first;
    bool same = true;
    const auto t_r = std::chrono::steady_clock::now();
    std::vector<Fe> pool(params.n + 1 + 2 * (size_t)k);
    for (size_t i = 0; i < pool.size(); i++) pool[i] = field::from_u64(SF, (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ULL + 1);
    printf("%zu random scalars drawn in %.1f ms (outside the timed region); %s\n", pool.size(),
           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_r).count(),
           stepwise ? "step-by-step form (host-pointer calls around h2_ipa_rounds)" : "one-call form (h2_open)");
    {   // the mirror's own share of the timed region: a fresh Vec of n scalars filled from the rng, as create_proof does (:44-47)
        const auto t_f = std::chrono::steady_clock::now();
        size_t c0 = 0;
        std::vector<Fe> s_poly(params.n);
        for (Fe &c : s_poly) c = pool[c0++];
        printf("of which the mirror's allocation and fill of s_poly (host): %.3f ms (checksum %llu)\n",
               std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_f).count(), (unsigned long long)s_poly[params.n - 1][0]);
    }
    for (int rep = 0; rep < reps; ++rep) {
        size_t ctr = 0;
        auto rng = [&]() { return pool[ctr++]; };
        Blake2bWrite<CURVE> tr;
        tr.write_point(to_affine<CURVE>(params.commit(px, blind)));
        const Fe x = tr.squeeze_challenge_scalar();
        tr.write_scalar(eval_polynomial<SF>(px, x));
        const auto t0 = std::chrono::steady_clock::now();
        if (stepwise) create_proof_stepwise<CURVE>(params, rng, tr, px, blind, x);
        else create_proof<CURVE>(params, rng, tr, px, blind, x);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const std::vector<uint8_t> bytes = tr.finalize();
        if (rep == 0) first = bytes;
        same = same && bytes == first;
        printf("opening argument k = %u, repetition %d: %.3f ms (%zu bytes)\n", k, rep, ms, bytes.size());
    }
    printf(same ? "every repetition wrote the same bytes\n" : "FAIL: repetitions differ\n");
    return same ? 0 : 1;
}

int main(int argc, char **argv) {
    if (h2_device_count() <= 0) { printf("no GPU: host mirror check needs an MI355X\n"); return 2; }
    if (argc == 3 && std::string(argv[1]) == "opening") return opening_mode((uint32_t)atoi(argv[2]));
    if (argc >= 3 && std::string(argv[1]) == "opening-time") return opening_time_mode((uint32_t)atoi(argv[2]), argc > 3 ? atoi(argv[3]) : 5, argc > 4 && std::string(argv[4]) == "stepwise");
    constexpr int CURVE = H2_VESTA, FIELD = H2_FP;   // every proof in the reference runs on Vesta / Fp
    const uint32_t k = 8;
    const size_t n = (size_t)1 << k;
    // generators: seeded multiples of (-1, 2) (pinned on-curve point, poly/commitment/msm.rs:181)
    uint64_t gen[8] = {0}; {
        const auto &Fq = field::params(H2_FQ);
        uint64_t m1[4] = {Fq.p[0] - 1, Fq.p[1], Fq.p[2], Fq.p[3]}, two[4] = {2, 0, 0, 0};
        memcpy(gen, m1, 32); memcpy(gen + 4, two, 32);
        orc_to_mont(H2_FQ, gen, 2);
    }
    std::vector<Affine> g(n), gl(n);
    orc_generate_bases(CURVE, gen, 11, g[0].data(), n);
    orc_generate_bases(CURVE, gen, 12, gl[0].data(), n);
    Affine w, u;
    orc_generate_bases(CURVE, gen, 13, w.data(), 1);
    orc_generate_bases(CURVE, gen, 14, u.data(), 1);

    // test_multiexp (arithmetic.rs:440-458)
    std::vector<Fe> coeffs(n);
    orc_random_field(FIELD, 21, coeffs[0].data(), n);
    Jacobian got = best_multiexp<CURVE>(coeffs, g);
    uint64_t want[12];
    orc_best_multiexp(CURVE, coeffs[0].data(), g[0].data(), n, want);
    EXPECT(same_point(CURVE, got, want), "best_multiexp == oracle (k = 8, Vesta)");
    bool threw = false;
    try { std::vector<Fe> c2(n - 1); best_multiexp<CURVE>(c2, g); } catch (const std::invalid_argument &) { threw = true; }
    EXPECT(threw, "best_multiexp rejects mismatched lengths (arithmetic.rs:144)");

    // Params::commit / commit_lagrange with a blind (commitment.rs:119-150)
    Params<CURVE> params(k, g, gl, w, u);
    Blind<CURVE> r{field::from_u64(FIELD, 987654321)};
    orc_commit(CURVE, g[0].data(), w.data(), coeffs[0].data(), r.value.data(), n, want);
    EXPECT(same_point(CURVE, params.commit(coeffs, r), want), "Params::commit == oracle");
    orc_commit(CURVE, gl[0].data(), w.data(), coeffs[0].data(), r.value.data(), n, want);
    EXPECT(same_point(CURVE, params.commit_lagrange(coeffs, r), want), "Params::commit_lagrange == oracle");

    // EvaluationDomain (domain.rs): constants, iFFT, coset FFT, round trip
    EvaluationDomain<FIELD> dom(5, k);     // benches/plonk.rs: cs_degree 5 -> extended_k = k + 2
    EXPECT(dom.extended_k == 10 && dom.t_evaluations.size() == 4, "EvaluationDomain::new(5, 8): extended_k = 10, 4 t_evaluations");
    Fe w3 = field::mul(FIELD, field::mul(FIELD, dom.g_coset, dom.g_coset), dom.g_coset);
    EXPECT(w3 == field::one(FIELD) && dom.g_coset != field::one(FIELD), "zeta^3 = 1, zeta != 1");
    std::vector<Fe> a(n);
    orc_random_field(FIELD, 31, a[0].data(), n);
    std::vector<Fe> ref = a;
    orc_ifft(FIELD, ref[0].data(), dom.omega_inv.data(), k, dom.ifft_divisor.data());
    std::vector<Fe> coeff = dom.lagrange_to_coeff(a);
    EXPECT(coeff == ref, "lagrange_to_coeff == oracle ifft");
    std::vector<Fe> ext_ref(dom.extended_len());
    memcpy(ext_ref[0].data(), ref[0].data(), n * 32);
    orc_coeff_to_extended(FIELD, ext_ref[0].data(), k, dom.extended_k, dom.g_coset.data(), dom.g_coset_inv.data(), dom.extended_omega.data());
    std::vector<Fe> ext = dom.coeff_to_extended(coeff);
    EXPECT(ext == ext_ref, "coeff_to_extended == oracle");
    std::vector<Fe> back = dom.extended_to_coeff(ext);
    bool rt = back.size() == n * dom.quotient_poly_degree && std::equal(coeff.begin(), coeff.end(), back.begin());
    for (size_t i = n; i < back.size(); i++) rt = rt && back[i] == Fe{0, 0, 0, 0};
    EXPECT(rt, "extended_to_coeff(coeff_to_extended(p)) == p, zero padded to n * (degree - 1)");
    std::vector<Fe> f = a;
    best_fft<FIELD>(f, dom.omega, k);
    std::vector<Fe> fref = a;
    orc_best_fft(FIELD, fref[0].data(), dom.omega.data(), k);
    EXPECT(f == fref, "best_fft == oracle");
    // arithmetic.rs:298-341 helpers and divide_by_vanishing_poly (domain.rs:329)
    Fe pt = field::from_u64(FIELD, 0x1234567), ev_ref, ip_ref;
    orc_eval_polynomial(FIELD, a[0].data(), n, pt.data(), ev_ref.data());
    EXPECT(eval_polynomial<FIELD>(a, pt) == ev_ref, "eval_polynomial == oracle");
    orc_inner_product(FIELD, a[0].data(), coeffs[0].data(), n, ip_ref.data());
    EXPECT(compute_inner_product<FIELD>(a, coeffs) == ip_ref, "compute_inner_product == oracle");
    threw = false;
    try { std::vector<Fe> c2(n - 1); compute_inner_product<FIELD>(a, c2); } catch (const std::invalid_argument &) { threw = true; }
    EXPECT(threw, "compute_inner_product rejects mismatched lengths (arithmetic.rs:311)");
    std::vector<Fe> q_ref(n - 1);
    orc_kate_division(FIELD, a[0].data(), n, pt.data(), q_ref[0].data());
    EXPECT(kate_division<FIELD>(a, pt) == q_ref, "kate_division == oracle");
    std::vector<Fe> dv_ref = ext;
    orc_divide_by_vanishing_poly(FIELD, dv_ref[0].data(), dom.extended_k, dom.t_evaluations[0].data(), dom.t_evaluations.size());
    EXPECT(dom.divide_by_vanishing_poly(ext) == dv_ref, "divide_by_vanishing_poly == oracle");

    // Params::write -> Params::read (commitment.rs:323-326 in test_opening_proof)
    std::stringstream file;
    params.write(file);
    EXPECT(file.str().size() == 4 + 32 * (2 * n + 2), "Params::write: 4 + 32 (2n + 2) bytes");
    Params<CURVE> again = Params<CURVE>::read(file);
    EXPECT(again.k == k && again.g == g && again.g_lagrange == gl && again.w == w && again.u == u, "Params::read(Params::write(p)) == p");
    orc_commit(CURVE, g[0].data(), w.data(), coeffs[0].data(), r.value.data(), n, want);
    EXPECT(same_point(CURVE, again.commit(coeffs, r), want), "re-read Params commit == oracle");
    std::string bad = file.str();
    bad[4 + 31] = (char)0x7f; for (int i = 0; i < 31; i++) bad[4 + i] = (char)0xff;     // x >= p
    std::stringstream bad_file(bad);
    threw = false;
    try { Params<CURVE>::read(bad_file); } catch (const std::runtime_error &) { threw = true; }
    EXPECT(threw, "Params::read rejects a non-canonical x");
    // ---- values the REFERENCE pins (halo2_proofs/tests/plonk_api.rs:958-981): Params::new(5) on Vesta -----------------------
    {
        Params<CURVE> p5 = Params<CURVE>::new_params(5);
        // fixed_commitments[0] commits to a never-assigned column with Blind::default() = 1: it is w = hasher(&[1]) (commitment.rs:102-103)
        const uint64_t wx[4] = {0x17d45c3e6fd92075ULL, 0x82548b6051713660ULL, 0xf24f9a4b0cc18318ULL, 0x2bbc94ef7b22aebeULL};
        const uint64_t wy[4] = {0xaf559fde4e3abd97ULL, 0xf47a5c8cc4aa7fa0ULL, 0x943bfb759fb02138ULL, 0x082b801a6e176239ULL};
        uint64_t pinned[8];
        memcpy(pinned, wx, 32); memcpy(pinned + 4, wy, 32);
        orc_to_mont(H2_FQ, pinned, 2);
        EXPECT(memcmp(p5.w.data(), pinned, 64) == 0, "Params::new(5).w == fixed_commitments[0] of tests/plonk_api.rs:959");
        auto hasher = hash_to_curve<CURVE>("Halo2-Parameters");
        EXPECT(hasher({1}) == p5.w && hasher({2}) == p5.u && hasher({0, 3, 0, 0, 0}) == p5.g[3], "hash_to_curve closure == Params::new's generators");
        // the all-zero Lagrange column with blind 1 commits to w through the registered table; and fixed_commitments[5] (sp: row 0 = 1)
        EvaluationDomain<FIELD> d5(4, 5);
        Polynomial<FIELD, LagrangeCoeff> zero = d5.lagrange_from_vec(std::vector<Fe>(32, Fe{0, 0, 0, 0}));
        Blind<CURVE> one{field::one(FIELD)};
        Jacobian cz = p5.commit_lagrange(zero, one);
        uint64_t waff[12];
        memcpy(waff, p5.w.data(), 64); memcpy(waff + 8, field::one(H2_FQ).data(), 32);
        EXPECT(same_point(CURVE, cz, waff), "commit_lagrange(0, Blind::default()) == w (typed Polynomial<LagrangeCoeff>)");
        Polynomial<FIELD, LagrangeCoeff> sp = zero;
        sp[0] = field::one(FIELD);
        const uint64_t sx[4] = {0xa89736f5c4b5ae9bULL, 0xbddd35e5929a90a4ULL, 0xd3ee48fb8d769da5ULL, 0x224ef42758215157ULL};
        const uint64_t sy[4] = {0x9a30ad2febb511c1ULL, 0x6d71e996e2165f7aULL, 0xde764f1492ecef95ULL, 0x11bc3a1e08eb320cULL};
        uint64_t sp_pinned[12];
        memcpy(sp_pinned, sx, 32); memcpy(sp_pinned + 4, sy, 32);
        orc_to_mont(H2_FQ, sp_pinned, 2);
        memcpy(sp_pinned + 8, field::one(H2_FQ).data(), 32);
        EXPECT(same_point(CURVE, p5.commit_lagrange(sp, one), sp_pinned), "commit_lagrange(sp column) == fixed_commitments[5] of tests/plonk_api.rs:964");
        // commit(iFFT(a)) == commit_lagrange(a) (commitment.rs:258-302) with the typed API
        std::vector<Fe> av(32);
        orc_random_field(FIELD, 41, av[0].data(), 32);
        Polynomial<FIELD, LagrangeCoeff> al = d5.lagrange_from_vec(av);
        Polynomial<FIELD, Coeff> ac = d5.lagrange_to_coeff(al);
        Jacobian c1 = p5.commit(ac, r), c2 = p5.commit_lagrange(al, r);
        uint64_t c2a[12];
        memcpy(c2a, c2.data(), 96);
        EXPECT(same_point(CURVE, c1, c2a), "commit(lagrange_to_coeff(a)) == commit_lagrange(a) over Params::new(5)");
        // benches/arithmetic.rs:15-33: small_multiexp on the 16 (g_lo, g_hi) pairs
        bool sm_ok = true;
        std::vector<Fe> two_c(2);
        orc_random_field(FIELD, 42, two_c[0].data(), 2);
        for (int i = 0; i < 16; i++) {
            std::vector<Affine> pair = {p5.g[i], p5.g[16 + i]};
            orc_best_multiexp(CURVE, two_c[0].data(), pair[0].data(), 2, want);
            sm_ok = sm_ok && same_point(CURVE, small_multiexp<CURVE>(two_c, pair), want);
        }
        EXPECT(sm_ok, "small_multiexp on the 16 generator pairs of benches/arithmetic.rs == oracle");
        // the column loop over "two GPUs" (the one device twice: two host threads inside the library)
        std::vector<std::vector<Fe>> cols(5, std::vector<Fe>(32));
        std::vector<Blind<CURVE>> bls(5);
        for (int i = 0; i < 5; i++) { orc_random_field(FIELD, 50 + i, cols[i][0].data(), 32); orc_random_field(FIELD, 60 + i, bls[i].value.data(), 1); }
        std::vector<Jacobian> multi = commit_columns_multi<CURVE>({p5.handle_g(), p5.handle_g()}, {0, 0}, cols, p5.w, bls);
        bool multi_ok = true;
        for (int i = 0; i < 5; i++) {
            orc_commit(CURVE, p5.g[0].data(), p5.w.data(), cols[i][0].data(), bls[i].value.data(), 32, want);
            multi_ok = multi_ok && same_point(CURVE, multi[i], want);
        }
        EXPECT(multi_ok, "commit_columns_multi over devices {0, 0} == oracle commits");
    }
    printf(fails ? "HOST MIRROR CHECK FAILED (%d)\n" : "HOST MIRROR CHECK OK\n", fails);
    return fails ? 1 : 0;
}
