// halo2_host.hpp -- C++ mirror of the reference's interface for the hot path, on top of the C ABI
// (include/halo2_mi355x.h).  The reference's host language is Rust (no toolchain in this image), so this
// header plays the role the Rust shim of INTEGRATION.md would: same names, argument meaning and error
// behaviour as
//     arithmetic::best_multiexp / best_fft            halo2_proofs/src/arithmetic.rs:143, :192
//     EvaluationDomain::{new, lagrange_to_coeff, coeff_to_extended, extended_to_coeff}
//                                                     halo2_proofs/src/poly/domain.rs:40, :227, :241, :303
//     arithmetic::{eval_polynomial, compute_inner_product, kate_division}   arithmetic.rs:298, :308, :322
//     EvaluationDomain::divide_by_vanishing_poly      halo2_proofs/src/poly/domain.rs:329
//     Params::{new, commit, commit_lagrange, write, read}, Blind   halo2_proofs/src/poly/commitment.rs:38, :119, :135, :169, :184, :208
//     CurveExt::hash_to_curve (pasta_curves, as called at commitment.rs:52, :102), arithmetic::small_multiexp (arithmetic.rs:116)
//     Polynomial<F, B> with the basis markers Coeff / LagrangeCoeff / ExtendedLagrangeCoeff   halo2_proofs/src/poly.rs:30-57
//     commit_columns_multi: the column loop of a prover phase (plonk/prover.rs:93-101, 301-313) over several GPUs
//     Blake2bWrite + Challenge255 (transcript.rs:150-300) and poly::commitment::create_proof, the opening argument
//                                                     halo2_proofs/src/poly/commitment/prover.rs:26-151
// Where the reference panics (assert_eq! on lengths) these throw std::invalid_argument; HIP / device
// failures throw std::runtime_error with h2_last_error().  All compute happens in libhalo2_mi355x.so.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <functional>
#include <cstring>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/halo2_mi355x.h"

namespace halo2 {

using Fe = std::array<uint64_t, 4>;        // field element, Montgomery limbs (what Rust's Fp/Fq holds)
using Affine = std::array<uint64_t, 8>;    // {x, y}; identity = all zero
using Jacobian = std::array<uint64_t, 12>; // {X, Y, Z}; identity: Z = 0

inline void check(int rc, const char *what) {
    if (rc == H2_OK) return;
    if (rc == H2_ERR_ARGS) throw std::invalid_argument(std::string(what) + ": bad arguments");
    throw std::runtime_error(std::string(what) + ": " + h2_last_error());
}

// ---- host-side Pasta field arithmetic (constants only; pasta_curves supplies these to the reference) ------
namespace field {
typedef unsigned __int128 u128;
struct Params { uint64_t p[4], inv, r2[4], one[4]; };
inline const Params &params(int f) {
    static const Params P[2] = {
        {{0x992d30ed00000001ULL, 0x224698fc094cf91bULL, 0, 0x4000000000000000ULL}, 0x992d30ecffffffffULL,
         {0x8c78ecb30000000fULL, 0xd7d30dbd8b0de0e7ULL, 0x7797a99bc3c95d18ULL, 0x096d41af7b9cb714ULL},
         {0x34786d38fffffffdULL, 0x992c350be41914adULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL}},
        {{0x8c46eb2100000001ULL, 0x224698fc0994a8ddULL, 0, 0x4000000000000000ULL}, 0x8c46eb20ffffffffULL,
         {0xfc9678ff0000000fULL, 0x67bb433d891a16e3ULL, 0x7fae231004ccf590ULL, 0x096d41af7ccfdaa9ULL},
         {0x5b2b3e9cfffffffdULL, 0x992c350be3420567ULL, 0xffffffffffffffffULL, 0x3fffffffffffffffULL}}};
    return P[f];
}
inline Fe mul(int f, const Fe &a, const Fe &b) {   // Montgomery product
    const Params &F = params(f);
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F.inv;
        c = ((u128)m * F.p[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * F.p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    bool ge = t[4] != 0;
    if (!ge) { ge = true; for (int i = 3; i >= 0; i--) { if (t[i] > F.p[i]) break; if (t[i] < F.p[i]) { ge = false; break; } } }
    if (ge) { u128 br = 0; for (int i = 0; i < 4; i++) { u128 d = (u128)t[i] - F.p[i] - (uint64_t)br; t[i] = (uint64_t)d; br = (d >> 64) & 1; } }
    return Fe{t[0], t[1], t[2], t[3]};
}
inline Fe one(int f) { const Params &F = params(f); return Fe{F.one[0], F.one[1], F.one[2], F.one[3]}; }
inline Fe from_u64(int f, uint64_t v) { const Params &F = params(f); return mul(f, Fe{v, 0, 0, 0}, Fe{F.r2[0], F.r2[1], F.r2[2], F.r2[3]}); }
inline Fe pow(int f, Fe base, const uint64_t e[4]) {
    Fe acc = one(f);
    for (int i = 0; i < 256; i++) { if ((e[i / 64] >> (i % 64)) & 1) acc = mul(f, acc, base); base = mul(f, base, base); }
    return acc;
}
inline Fe inv(int f, const Fe &a) {   // a^(p-2)
    const Params &F = params(f);
    uint64_t e[4] = {F.p[0] - 2, F.p[1], F.p[2], F.p[3]};
    return pow(f, a, e);
}
inline Fe sub(int f, const Fe &a, const Fe &b) {
    const Params &F = params(f);
    uint64_t t[4]; u128 br = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - (uint64_t)br; t[i] = (uint64_t)d; br = (d >> 64) & 1; }
    if (br) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)t[i] + F.p[i]; t[i] = (uint64_t)c; c >>= 64; } }
    return Fe{t[0], t[1], t[2], t[3]};
}
inline Fe add(int f, const Fe &a, const Fe &b) {
    const Params &F = params(f);
    uint64_t t[4]; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; t[i] = (uint64_t)c; c >>= 64; }
    Fe r{t[0], t[1], t[2], t[3]};
    bool ge = true;
    for (int i = 3; i >= 0; i--) { if (t[i] > F.p[i]) break; if (t[i] < F.p[i]) { ge = false; break; } }
    return ge ? sub(f, r, Fe{F.p[0], F.p[1], F.p[2], F.p[3]}) : r;
}
inline Fe from_mont(int f, const Fe &a) { return mul(f, a, Fe{1, 0, 0, 0}); }            // canonical limbs
inline Fe to_mont(int f, const Fe &raw) { const Params &F = params(f); return mul(f, raw, Fe{F.r2[0], F.r2[1], F.r2[2], F.r2[3]}); }   // any raw < 2^256
// ROOT_OF_UNITY = 5^((p-1)/2^32): multiplicative generator 5, S = 32 for both fields
inline Fe root_of_unity(int f) {
    const Params &F = params(f);
    uint64_t e[4] = {(F.p[0] >> 32) | (F.p[1] << 32), (F.p[1] >> 32) | (F.p[2] << 32), (F.p[2] >> 32) | (F.p[3] << 32), F.p[3] >> 32};
    return pow(f, from_u64(f, 5), e);   // (p - 1) >> 32 == p >> 32 because p = 1 mod 2^32
}
// ZETA: primitive cube root of unity (pasta_curves): Fp (5^((p-1)/3))^2, Fq 5^((q-1)/3)
inline Fe zeta(int f) {
    const Params &F = params(f);
    // (p - 1) / 3 by long division on the limbs
    uint64_t n[4] = {F.p[0] - 1, F.p[1], F.p[2], F.p[3]}, q[4];
    u128 rem = 0;
    for (int i = 3; i >= 0; i--) { u128 cur = (rem << 64) | n[i]; q[i] = (uint64_t)(cur / 3); rem = cur % 3; }
    Fe z = pow(f, from_u64(f, 5), q);
    return f == H2_FP ? mul(f, z, z) : z;
}
}  // namespace field

// ---- arithmetic.rs ----------------------------------------------------------------------------------------
template <int CURVE>
inline Jacobian best_multiexp(const std::vector<Fe> &coeffs, const std::vector<Affine> &bases) {
    if (coeffs.size() != bases.size()) throw std::invalid_argument("best_multiexp: coeffs.len() != bases.len()");   // arithmetic.rs:144
    Jacobian out{};
    check(h2_msm(CURVE, coeffs.empty() ? nullptr : coeffs[0].data(), bases.empty() ? nullptr : bases[0].data(), coeffs.size(),
                 H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, out.data()), "h2_msm");
    return out;
}
template <int FIELD> inline void best_fft(std::vector<Fe> &a, const Fe &omega, uint32_t log_n) {
    if (a.size() != ((size_t)1 << log_n)) throw std::invalid_argument("best_fft: a.len() != 1 << log_n");       // arithmetic.rs:205
    check(h2_ntt(FIELD, a[0].data(), log_n, omega.data(), H2_FORM_MONTGOMERY), "h2_ntt");
}

// small_multiexp (arithmetic.rs:116-136): the same sum; on the device both are one kernel path
template <int CURVE>
inline Jacobian small_multiexp(const std::vector<Fe> &coeffs, const std::vector<Affine> &bases) { return best_multiexp<CURVE>(coeffs, bases); }

// C::CurveExt::hash_to_curve(domain_prefix): returns the hasher closure, as pasta_curves does (commitment.rs:52, :102)
template <int CURVE> inline std::function<Affine(const std::vector<uint8_t> &)> hash_to_curve(const std::string &domain_prefix) {
    return [domain_prefix](const std::vector<uint8_t> &message) {
        Affine out{};
        check(h2_hash_to_curve(CURVE, domain_prefix.c_str(), message.empty() ? nullptr : message.data(), message.size(), 1, H2_FORM_MONTGOMERY,
                               out.data()), "h2_hash_to_curve");
        return out;
    };
}

// ---- poly.rs: Polynomial<F, B>; the basis is a type, as in the reference, so mixing bases does not compile -----------------
struct Coeff {};
struct LagrangeCoeff {};
struct ExtendedLagrangeCoeff {};
template <int FIELD, typename Basis> struct Polynomial {
    std::vector<Fe> values;
    size_t len() const { return values.size(); }
    Fe &operator[](size_t i) { return values[i]; }
    const Fe &operator[](size_t i) const { return values[i]; }
};

template <int FIELD> inline Fe eval_polynomial(const std::vector<Fe> &poly, const Fe &point) {                  // arithmetic.rs:298
    Fe out{};
    check(h2_eval_polynomial(FIELD, poly.empty() ? nullptr : poly[0].data(), poly.size(), point.data(), H2_FORM_MONTGOMERY, out.data()),
          "h2_eval_polynomial");
    return out;
}
template <int FIELD> inline Fe compute_inner_product(const std::vector<Fe> &a, const std::vector<Fe> &b) {       // arithmetic.rs:308
    if (a.size() != b.size()) throw std::invalid_argument("compute_inner_product: a.len() != b.len()");          // :311
    Fe out{};
    check(h2_inner_product(FIELD, a.empty() ? nullptr : a[0].data(), b.empty() ? nullptr : b[0].data(), a.size(), H2_FORM_MONTGOMERY,
                           out.data()), "h2_inner_product");
    return out;
}
template <int FIELD> inline std::vector<Fe> kate_division(const std::vector<Fe> &a, const Fe &b) {              // arithmetic.rs:322
    if (a.empty()) throw std::invalid_argument("kate_division: empty polynomial");
    std::vector<Fe> q(a.size() - 1);
    check(h2_kate_division(FIELD, a[0].data(), a.size(), b.data(), H2_FORM_MONTGOMERY, q.empty() ? nullptr : q[0].data()), "h2_kate_division");
    return q;
}

// ---- poly/domain.rs ------------------------------------------------------------------------------------------
template <int FIELD> class EvaluationDomain {
  public:
    uint32_t k, extended_k;
    uint64_t n, quotient_poly_degree;
    Fe omega, omega_inv, extended_omega, extended_omega_inv, g_coset, g_coset_inv, ifft_divisor, extended_ifft_divisor;
    std::vector<Fe> t_evaluations;

    EvaluationDomain(uint32_t j, uint32_t k_) : k(k_), n((uint64_t)1 << k_), quotient_poly_degree(j - 1) {   // domain.rs:40
        extended_k = k;
        while (((uint64_t)1 << extended_k) < n * quotient_poly_degree) extended_k++;
        if (extended_k > 32) throw std::invalid_argument("EvaluationDomain: extended_k > S");                // domain.rs:56
        extended_omega = field::root_of_unity(FIELD);
        for (uint32_t i = extended_k; i < 32; i++) extended_omega = field::mul(FIELD, extended_omega, extended_omega);
        omega = extended_omega;
        for (uint32_t i = k; i < extended_k; i++) omega = field::mul(FIELD, omega, omega);
        omega_inv = field::inv(FIELD, omega);
        extended_omega_inv = field::inv(FIELD, extended_omega);
        g_coset = field::zeta(FIELD);
        g_coset_inv = field::mul(FIELD, g_coset, g_coset);
        ifft_divisor = field::inv(FIELD, field::from_u64(FIELD, (uint64_t)1 << k));
        extended_ifft_divisor = field::inv(FIELD, field::from_u64(FIELD, (uint64_t)1 << extended_k));
        uint64_t e[4] = {n, 0, 0, 0};
        Fe orig = field::pow(FIELD, g_coset, e), step = field::pow(FIELD, extended_omega, e), cur = orig;
        do {                                                                                                   // domain.rs:85-110
            t_evaluations.push_back(field::inv(FIELD, field::sub(FIELD, cur, field::one(FIELD))));
            cur = field::mul(FIELD, cur, step);
        } while (cur != orig);
    }
    size_t extended_len() const { return (size_t)1 << extended_k; }

    std::vector<Fe> lagrange_to_coeff(std::vector<Fe> a) const {                                                // domain.rs:227
        if (a.size() != n) throw std::invalid_argument("lagrange_to_coeff: wrong length");
        check(h2_ifft(FIELD, a[0].data(), k, omega_inv.data(), ifft_divisor.data(), H2_FORM_MONTGOMERY), "h2_ifft");
        return a;
    }
    std::vector<Fe> coeff_to_extended(const std::vector<Fe> &a) const {                                         // domain.rs:241
        if (a.size() != n) throw std::invalid_argument("coeff_to_extended: wrong length");
        std::vector<Fe> out(extended_len());
        check(h2_coeff_to_extended(FIELD, a[0].data(), out[0].data(), k, extended_k, g_coset.data(), g_coset_inv.data(),
                                   extended_omega.data(), H2_FORM_MONTGOMERY), "h2_coeff_to_extended");
        return out;
    }
    std::vector<Fe> extended_to_coeff(std::vector<Fe> a) const {                                                // domain.rs:303
        if (a.size() != extended_len()) throw std::invalid_argument("extended_to_coeff: wrong length");
        check(h2_extended_to_coeff(FIELD, a[0].data(), extended_k, g_coset.data(), g_coset_inv.data(), extended_omega_inv.data(),
                                   extended_ifft_divisor.data(), H2_FORM_MONTGOMERY), "h2_extended_to_coeff");
        a.resize(n * quotient_poly_degree);
        return a;
    }
    std::vector<Fe> divide_by_vanishing_poly(std::vector<Fe> a) const {                                         // domain.rs:329
        if (a.size() != extended_len()) throw std::invalid_argument("divide_by_vanishing_poly: wrong length");  // :333
        check(h2_divide_by_vanishing_poly(FIELD, a[0].data(), extended_k, t_evaluations[0].data(), t_evaluations.size(), H2_FORM_MONTGOMERY),
              "h2_divide_by_vanishing_poly");
        return a;
    }
    // the same transforms on tagged polynomials (domain.rs:151-237 signatures)
    Polynomial<FIELD, LagrangeCoeff> lagrange_from_vec(std::vector<Fe> v) const {
        if (v.size() != n) throw std::invalid_argument("lagrange_from_vec: wrong length");
        return {std::move(v)};
    }
    Polynomial<FIELD, Coeff> coeff_from_vec(std::vector<Fe> v) const {
        if (v.size() != n) throw std::invalid_argument("coeff_from_vec: wrong length");
        return {std::move(v)};
    }
    Polynomial<FIELD, Coeff> lagrange_to_coeff(Polynomial<FIELD, LagrangeCoeff> a) const { return {lagrange_to_coeff(std::move(a.values))}; }
    Polynomial<FIELD, ExtendedLagrangeCoeff> coeff_to_extended(const Polynomial<FIELD, Coeff> &a) const { return {coeff_to_extended(a.values)}; }
    std::vector<Fe> extended_to_coeff(Polynomial<FIELD, ExtendedLagrangeCoeff> a) const { return extended_to_coeff(std::move(a.values)); }
    Polynomial<FIELD, ExtendedLagrangeCoeff> divide_by_vanishing_poly(Polynomial<FIELD, ExtendedLagrangeCoeff> a) const {
        return {divide_by_vanishing_poly(std::move(a.values))};
    }
};

// ---- poly/commitment.rs ------------------------------------------------------------------------------------------
template <int CURVE> struct Blind { Fe value; };

template <int CURVE> class Params {
  public:
    uint32_t k;
    uint64_t n;
    std::vector<Affine> g, g_lagrange;
    Affine w, u;
    // what Params::read (commitment.rs:184) ends with: generators in hand; `new`'s hash-to-curve is upstream of the hot path
    Params(uint32_t k_, std::vector<Affine> g_, std::vector<Affine> g_lagrange_, const Affine &w_, const Affine &u_)
        : k(k_), n((uint64_t)1 << k_), g(std::move(g_)), g_lagrange(std::move(g_lagrange_)), w(w_), u(u_) {
        if (g.size() != n || g_lagrange.size() != n) throw std::invalid_argument("Params: need 2^k generators");
        const int wb = h2_commit_column_window_bits(n);        // tables of column commits: 17-bit windows from 2^18 points on
        check(h2_bases_register_ex(CURVE, g[0].data(), n, H2_FORM_MONTGOMERY, wb, &h_g), "h2_bases_register_ex");
        check(h2_bases_register_ex(CURVE, g_lagrange[0].data(), n, H2_FORM_MONTGOMERY, wb, &h_gl), "h2_bases_register_ex");
        // `w` is a field of Params (commitment.rs:26-33): installed once per table; a commit then passes only its blind scalar
        check(h2_bases_set_blind_base(h_g, w.data(), H2_FORM_MONTGOMERY), "h2_bases_set_blind_base");
        check(h2_bases_set_blind_base(h_gl, w.data(), H2_FORM_MONTGOMERY), "h2_bases_set_blind_base");
    }
    ~Params() { for (h2_bases_t h : {h_g, h_gl, h_open}) if (h) h2_bases_free(h); }
    // Params::new (commitment.rs:38-114): g_i = hasher({0, i as LE u32}), g_lagrange by the point iFFT, w = hasher({1}), u = hasher({2});
    // the 2^k + 2 hashes and the point FFT run on the device
    static Params new_params(uint32_t k_) {
        if (k_ >= 32) throw std::invalid_argument("Params::new: k < 32");                                              // :41
        const size_t n_ = (size_t)1 << k_;
        std::vector<uint8_t> msgs(5 * n_, 0);
        for (size_t i = 0; i < n_; i++) { const uint32_t le = (uint32_t)i; memcpy(&msgs[5 * i + 1], &le, 4); }         // :57-58 (little-endian host)
        std::vector<Affine> g_(n_), gl_(n_);
        check(h2_hash_to_curve(CURVE, "Halo2-Parameters", msgs.data(), 5, n_, H2_FORM_MONTGOMERY, g_[0].data()), "h2_hash_to_curve");
        check(h2_lagrange_basis(CURVE, g_[0].data(), gl_[0].data(), k_, H2_FORM_MONTGOMERY), "h2_lagrange_basis");       // :77-100
        auto hasher = hash_to_curve<CURVE>("Halo2-Parameters");
        return Params(k_, std::move(g_), std::move(gl_), hasher({1}), hasher({2}));
    }
    // typed commits: the basis of the polynomial selects the generator set at compile time
    Jacobian commit(const Polynomial<CURVE == H2_PALLAS ? H2_FQ : H2_FP, Coeff> &poly, const Blind<CURVE> &r) const { return run(h_g, poly.values, r); }
    Jacobian commit_lagrange(const Polynomial<CURVE == H2_PALLAS ? H2_FQ : H2_FP, LagrangeCoeff> &poly, const Blind<CURVE> &r) const {
        return run(h_gl, poly.values, r);
    }
    h2_bases_t handle_g() const { return h_g; }
    h2_bases_t handle_g_lagrange() const { return h_gl; }
    // the registered basis of the opening argument's commits (prover.rs:107-114): g || u || u || w || w when one paired commit per
    // round is possible (h2_commit_pair_supported), else g || u || w; registered on first use
    h2_bases_t opening_basis(bool *paired) const {
        *paired = h2_commit_pair_supported(n + 4) != 0;
        if (!h_open) {
            std::vector<Affine> basis(g);
            if (*paired) basis.insert(basis.end(), {u, u, w, w});
            else basis.insert(basis.end(), {u, w});
            check(h2_bases_register(CURVE, basis[0].data(), basis.size(), H2_FORM_MONTGOMERY, &h_open), "h2_bases_register");
        }
        return h_open;
    }
    Params(const Params &) = delete;
    Params &operator=(const Params &) = delete;

    Jacobian commit(const std::vector<Fe> &poly, const Blind<CURVE> &r) const { return run(h_g, poly, r); }            // :119
    Jacobian commit_lagrange(const std::vector<Fe> &poly, const Blind<CURVE> &r) const { return run(h_gl, poly, r); }  // :135
    std::vector<Affine> get_g() const { return g; }

    void write(std::ostream &writer) const {                                                                            // :169-181
        const uint32_t k_le = k;                      // little-endian hosts only, like the GPU
        writer.write(reinterpret_cast<const char *>(&k_le), 4);
        std::vector<uint8_t> buf(n * 32);
        for (const std::vector<Affine> *v : {&g, &g_lagrange}) {
            check(h2_points_compress(CURVE, (*v)[0].data(), n, H2_FORM_MONTGOMERY, buf.data()), "h2_points_compress");
            writer.write(reinterpret_cast<const char *>(buf.data()), (std::streamsize)buf.size());
        }
        const Affine wu[2] = {w, u};
        check(h2_points_compress(CURVE, wu[0].data(), 2, H2_FORM_MONTGOMERY, buf.data()), "h2_points_compress");
        writer.write(reinterpret_cast<const char *>(buf.data()), 64);
    }
    static Params read(std::istream &reader) {                                                                          // :184-205
        uint32_t k_le = 0;
        if (!reader.read(reinterpret_cast<char *>(&k_le), 4) || k_le >= 32) throw std::runtime_error("Params::read: bad header");
        const size_t n_ = (size_t)1 << k_le;
        std::vector<uint8_t> buf(32 * (2 * n_ + 2));
        if (!reader.read(reinterpret_cast<char *>(buf.data()), (std::streamsize)buf.size())) throw std::runtime_error("Params::read: truncated");
        std::vector<Affine> pts(2 * n_ + 2);
        int rc = h2_points_decompress(CURVE, buf.data(), pts.size(), H2_FORM_MONTGOMERY, pts[0].data());
        if (rc == H2_ERR_DECODE) throw std::runtime_error("Params::read: invalid point encoding");                      // io::Error in the reference
        check(rc, "h2_points_decompress");
        return Params(k_le, std::vector<Affine>(pts.begin(), pts.begin() + n_), std::vector<Affine>(pts.begin() + n_, pts.begin() + 2 * n_),
                      pts[2 * n_], pts[2 * n_ + 1]);
    }
    Params(Params &&o) noexcept
        : k(o.k), n(o.n), g(std::move(o.g)), g_lagrange(std::move(o.g_lagrange)), w(o.w), u(o.u), h_g(o.h_g), h_gl(o.h_gl), h_open(o.h_open) {
        o.h_g = o.h_gl = o.h_open = 0;
    }

  private:
    h2_bases_t h_g = 0, h_gl = 0;
    mutable h2_bases_t h_open = 0;
    Jacobian run(h2_bases_t h, const std::vector<Fe> &poly, const Blind<CURVE> &r) const {
        if (poly.size() != n) throw std::invalid_argument("commit: poly.len() != n");
        Jacobian out{};
        check(h2_commit(h, poly[0].data(), n, nullptr, r.value.data(), H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, out.data()), "h2_commit");
        return out;
    }
};

// ---- transcript.rs -----------------------------------------------------------------------------------------------
// Blake2bWrite<_, C, Challenge255<C>> (transcript.rs:150-198, 286-296) over the library's transcript object (h2_transcript_*:
// BLAKE2b-512 personalised "Halo2-Transcript"; points absorbed as canonical (x, y) and written compressed; a challenge is the
// 64-byte digest of a copy of the state reduced into the scalar field).  Keeping the state in the library lets the opening
// argument's round loop (h2_ipa_rounds) absorb L_j, R_j and squeeze its challenges without a callback into this header.
template <int CURVE> class Blake2bWrite {
  public:
    Blake2bWrite() { check(h2_transcript_new(CURVE, &h), "h2_transcript_new"); }
    ~Blake2bWrite() { if (h) h2_transcript_free(h); }
    Blake2bWrite(const Blake2bWrite &) = delete;
    Blake2bWrite &operator=(const Blake2bWrite &) = delete;
    void common_point(const Affine &p) { point(h2_transcript_common_point(h, p.data(), 0)); }                          // :206-219
    void write_point(const Affine &p) { point(h2_transcript_write_point(h, p.data(), 0)); }                            // :183-187
    void write_point(const Jacobian &p) { point(h2_transcript_write_point(h, p.data(), 1)); }                          // .to_affine() first
    void common_scalar(const Fe &s) { check(h2_transcript_common_scalar(h, s.data()), "h2_transcript_common_scalar"); } // :221-227
    void write_scalar(const Fe &s) { check(h2_transcript_write_scalar(h, s.data()), "h2_transcript_write_scalar"); }    // :188-192
    Fe squeeze_challenge_scalar() {                                                                                    // :200-205, :286-296
        Fe u{};
        check(h2_transcript_squeeze_challenge(h, u.data()), "h2_transcript_squeeze_challenge");
        return u;
    }
    std::vector<uint8_t> finalize() const {
        size_t len = 0;
        check(h2_transcript_bytes(h, nullptr, 0, &len), "h2_transcript_bytes");
        std::vector<uint8_t> out(len);
        check(h2_transcript_bytes(h, out.data(), out.size(), &len), "h2_transcript_bytes");
        return out;
    }
    h2_transcript_t handle() const { return h; }

  private:
    h2_transcript_t h = 0;
    static void point(int rc) {
        if (rc == H2_ERR_ARGS) throw std::invalid_argument("cannot write points at infinity to the transcript");           // :209-214
        check(rc, "h2_transcript_write_point");
    }
};

template <int CURVE> inline Affine to_affine(const Jacobian &p) {
    constexpr int BF = CURVE == H2_PALLAS ? H2_FP : H2_FQ;
    const Fe X{p[0], p[1], p[2], p[3]}, Y{p[4], p[5], p[6], p[7]}, Z{p[8], p[9], p[10], p[11]};
    if (!(Z[0] | Z[1] | Z[2] | Z[3])) return Affine{};
    const Fe zi = field::inv(BF, Z), zi2 = field::mul(BF, zi, zi);
    const Fe x = field::mul(BF, X, zi2), y = field::mul(BF, Y, field::mul(BF, zi2, zi));
    return Affine{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
}

// ---- poly/commitment/prover.rs:26-151 ------------------------------------------------------------------------------
// create_proof: the opening argument for `p_poly` at `x_3`, written to `transcript`.  rng() -> one uniformly random scalar
// (C::Scalar::random); it is drawn n + 1 + 2k times in the reference's order (s_poly coefficients, s_poly_blind, then l_j, r_j
// per round).  Between the rng and the transcript the function is ONE library call (h2_open): the two host vectors cross PCIe
// once each and nothing comes back but c and f.  (Through round 4 only the round loop was native, h2_ipa_rounds, with host-pointer
// calls and two host Horner evaluations around it: 68-71 ms at k = 20 against ~19 now; `create_proof_stepwise` keeps that form.)
// The proof bytes are the reference's for the same randomness (tests/test_gpu_parity.py::test_native_drivers compares them with
// the Python mirror's).
template <int CURVE, class Rng>
inline void create_proof(const Params<CURVE> &params, Rng &&rng, Blake2bWrite<CURVE> &transcript, const std::vector<Fe> &p_poly,
                         const Blind<CURVE> &p_blind, const Fe &x_3) {
    const size_t n = params.n;
    const uint32_t k = params.k;
    if (p_poly.size() != n) throw std::invalid_argument("create_proof: p_poly.len() != params.n");                      // :41
    std::vector<Fe> s_poly(n);                                                                                         // :44-47
    for (Fe &c : s_poly) c = rng();
    const Blind<CURVE> s_poly_blind{rng()};                                                                            // :53
    std::vector<Fe> rands(2 * (size_t)k);                                                                              // :111-112
    for (Fe &r : rands) r = rng();
    bool paired = false;
    const h2_bases_t basis = params.opening_basis(&paired);
    const Affine uw[2] = {params.u, params.w};
    Fe c_final{}, f_final{};
    check(h2_open(CURVE, k, params.handle_g(), basis, paired ? 1 : 0, H2_IPA_SWITCH_DEFAULT, uw[0].data(), p_poly[0].data(), p_blind.value.data(),
                  x_3.data(), s_poly[0].data(), s_poly_blind.value.data(), rands[0].data(), h2_transcript_cb_write_point, h2_transcript_cb_squeeze,
                  (void *)(uintptr_t)transcript.handle(), c_final.data(), f_final.data()), "h2_open");                  // :49-142
    transcript.write_scalar(c_final);                                                                                  // :146-148
    transcript.write_scalar(f_final);
}

// the same argument step by step through the host-pointer entry points, the round loop alone native (the form before h2_open; kept
// as the A/B arm of `host_mirror_check opening-time` and as a second path to the same bytes)
template <int CURVE, class Rng>
inline void create_proof_stepwise(const Params<CURVE> &params, Rng &&rng, Blake2bWrite<CURVE> &transcript, const std::vector<Fe> &p_poly,
                                  const Blind<CURVE> &p_blind, const Fe &x_3) {
    constexpr int SF = CURVE == H2_PALLAS ? H2_FQ : H2_FP;
    const size_t n = params.n;
    const uint32_t k = params.k;
    if (p_poly.size() != n) throw std::invalid_argument("create_proof: p_poly.len() != params.n");                      // :41
    // a random polynomial with a root at x_3, and its commitment (:43-57)
    std::vector<Fe> s_poly(n);
    for (Fe &c : s_poly) c = rng();
    const Fe s_at_x3 = eval_polynomial<SF>(s_poly, x_3);
    s_poly[0] = field::sub(SF, s_poly[0], s_at_x3);
    const Blind<CURVE> s_poly_blind{rng()};
    transcript.write_point(params.commit(s_poly, s_poly_blind));
    const Fe xi = transcript.squeeze_challenge_scalar();                                                              // :62
    const Fe z = transcript.squeeze_challenge_scalar();                                                               // :66
    // P' = P - [v] G_0 + [xi] S (:70-78)
    std::vector<Fe> p_prime(std::move(s_poly));
    check(h2_scale_add(SF, p_prime[0].data(), xi.data(), p_poly[0].data(), n, H2_FORM_MONTGOMERY), "h2_scale_add");
    const Fe v = eval_polynomial<SF>(p_prime, x_3);
    p_prime[0] = field::sub(SF, p_prime[0], v);
    Fe f = field::add(SF, field::mul(SF, s_poly_blind.value, xi), p_blind.value);
    std::vector<Fe> b(n);                                                                                              // :86-97
    check(h2_powers(SF, x_3.data(), n, H2_FORM_MONTGOMERY, b[0].data()), "h2_powers");
    std::vector<Fe> rands(2 * (size_t)k);                                                                              // :111-112
    for (Fe &r : rands) r = rng();
    bool paired = false;
    const h2_bases_t basis = params.opening_basis(&paired);
    const Affine uw[2] = {params.u, params.w};
    Fe c_final{}, f_delta{};
    check(h2_ipa_rounds(CURVE, k, H2_IPA_SWITCH_DEFAULT, basis, paired ? 1 : 0, p_prime[0].data(), b[0].data(), z.data(), rands[0].data(),
                        uw[0].data(), h2_transcript_cb_write_point, h2_transcript_cb_squeeze, (void *)(uintptr_t)transcript.handle(),
                        c_final.data(), f_delta.data()), "h2_ipa_rounds");                                             // :104-142
    transcript.write_scalar(c_final);                                                                                  // :146-148
    transcript.write_scalar(field::add(SF, f, f_delta));
}

// The independent column commits of a prover phase (plonk/prover.rs:93-101, 301-313; vanishing/prover.rs:96-108) spread over the
// GPUs of one node from one process: column i -> devices[i % ndev], which holds handles[i % ndev] (the same bases registered on
// each device: h2_init(d), then construct a Params there).  Blocking; one host thread and three streams per device inside.
template <int CURVE>
inline std::vector<Jacobian> commit_columns_multi(const std::vector<h2_bases_t> &handles, const std::vector<int> &devices, const std::vector<std::vector<Fe>> &columns,
                                                  const Affine &w, const std::vector<Blind<CURVE>> &blinds) {
    if (handles.size() != devices.size() || columns.size() != blinds.size()) throw std::invalid_argument("commit_columns_multi: sizes differ");
    std::vector<Jacobian> out(columns.size());
    if (columns.empty()) return out;
    const size_t n = columns[0].size();
    std::vector<const uint64_t *> sc, bl;
    std::vector<uint64_t *> op;
    for (size_t i = 0; i < columns.size(); i++) {
        if (columns[i].size() != n) throw std::invalid_argument("commit_columns_multi: ragged columns");
        sc.push_back(columns[i][0].data());
        bl.push_back(blinds[i].value.data());
        op.push_back(out[i].data());
    }
    check(h2_commit_batch_multi(handles.data(), devices.data(), (int)devices.size(), sc.data(), columns.size(), n, w.data(), bl.data(),
                                H2_FORM_MONTGOMERY, H2_OUT_JACOBIAN, op.data()), "h2_commit_batch_multi");
    return out;
}

}  // namespace halo2
