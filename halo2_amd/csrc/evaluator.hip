// Expression evaluation over registered polynomials: `poly::Evaluator::evaluate`
// (halo2_proofs/src/poly/evaluator.rs:129-228), the step between the coset FFTs and the quotient iFFT of `create_proof`
// (plonk/prover.rs:340-520 builds the trees for custom gates, permutation and lookup arguments).  SURVEY.md section 8f-3.
//
// The reference walks the tree once per chunk and materialises a vector per node.  Here the host flattens the tree into
// post-order bytecode and ONE kernel evaluates it per element with a small stack: the top of the stack lives in registers,
// the rest in LDS (word-major, so a wave's accesses are conflict-free).  Every leaf is one coalesced 32-byte load per lane,
// every node at most one modular multiplication: the kernel is HBM-bound (32 B per leaf per element).
//
// Node semantics, element i of a vector of 2^log_len elements (evaluator.rs:153-209, 330-614):
//   POLY p, r       polys[p][(i + r) mod len]      r = rotation * step, step = 1 (Lagrange) or 2^(ext_k - k) (extended)
//   CONST c         Lagrange / extended: c at every i;  coefficient basis: c at i = 0, else 0
//   LINEAR c        Lagrange / extended: c * omega^i (the caller folds ZETA into c for the extended basis);  coefficient basis: c at i = 1
//   ADD, MUL        element-wise (MUL: not in the coefficient basis)
//   SCALE c         x * c
//   MULADD b        DistributePowers fold: acc * b + term
#include <vector>

#include "common.h"
#include "field.cuh"
#include "host_field.h"

namespace h2 {

enum : u32 { EV_POLY = 1, EV_CONST = 2, EV_LINEAR = 3, EV_ADD = 4, EV_MUL = 5, EV_SCALE = 6, EV_MULADD = 7 };
constexpr int kEvalDepth = 8;      // stack slots below the register-resident top
constexpr int kEvalThreads = 256;

__device__ __forceinline__ void ev_spill(u32 *lds, int level, const fe &v) {
#pragma unroll
    for (int w = 0; w < 8; ++w) lds[(level * 8 + w) * kEvalThreads + threadIdx.x] = v.v[w];
}
__device__ __forceinline__ fe ev_fill(const u32 *lds, int level) {
    fe v;
#pragma unroll
    for (int w = 0; w < 8; ++w) v.v[w] = lds[(level * 8 + w) * kEvalThreads + threadIdx.x];
    return v;
}

// basis: 0 coefficient, 1 Lagrange, 2 extended Lagrange.  tw: omega^0 .. omega^(len/2 - 1) (Montgomery) or null when the
// program has no LINEAR node.
template <int F>
__global__ void __launch_bounds__(kEvalThreads) ev_run(const u32 *__restrict__ program, u32 n_words, const u32 *__restrict__ consts,
                                                       const u32 *const *__restrict__ polys, unsigned log_len, int basis,
                                                       const u32 *__restrict__ tw, u32 *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const size_t len = (size_t)1 << log_len, i = (size_t)blockIdx.x * kEvalThreads + threadIdx.x;
    if (i >= len) return;
    fe top = fe_zero();
    int depth = 0;                      // values on the stack, the last one in `top`
    for (u32 pc = 0; pc < n_words; ++pc) {
        const u32 word = program[pc], op = word & 0xFFu, arg = word >> 8;
        if (op == EV_POLY || op == EV_CONST || op == EV_LINEAR) {
            if (depth > 0) ev_spill(lds, depth - 1, top);
            ++depth;
            if (op == EV_POLY) {
                const int shift = (int)program[++pc];
                const size_t j = (i + (size_t)((long long)shift + (long long)len)) & (len - 1);   // |shift| < len
                top = fe_load(polys[arg] + 8 * j);
            } else if (op == EV_CONST) {
                top = (basis != 0 || i == 0) ? fe_load(consts + 8 * (size_t)arg) : fe_zero();
            } else {
                if (basis == 0) {
                    top = i == 1 ? fe_load(consts + 8 * (size_t)arg) : fe_zero();
                } else {
                    const size_t halfn = len >> 1;
                    fe w = halfn ? fe_load(tw + 8 * (i & (halfn - 1))) : fe_one<F>();             // omega^(len/2) = -1
                    if (halfn && i >= halfn) w = fe_neg<F>(w);
                    top = fe_mulx<F>(w, fe_load(consts + 8 * (size_t)arg));
                }
            }
        } else if (op == EV_SCALE) {
            top = fe_mulx<F>(top, fe_load(consts + 8 * (size_t)arg));
        } else {
            const fe below = ev_fill(lds, depth - 2);
            --depth;
            if (op == EV_ADD) top = fe_add<F>(below, top);
            else if (op == EV_MUL) top = fe_mulx<F>(below, top);
            else top = fe_add<F>(fe_mulx<F>(below, fe_load(consts + 8 * (size_t)arg)), top);       // MULADD
        }
    }
    fe_store(out + 8 * i, top);
}

namespace {
struct EvalContext {
    std::mutex mu;
    DevBuf prog, consts, ptrs;
    void release_all() {
        prog.release();
        consts.release();
        ptrs.release();
    }
};
StreamContexts<EvalContext> g_eval_ctxs;
}  // namespace
void eval_release_workspaces() { g_eval_ctxs.release_current_device(); }

}  // namespace h2

using namespace h2;

extern "C" int h2_evaluate_device(int field, int basis, const uint32_t *program, size_t n_words, const uint64_t *consts, size_t n_consts,
                                  const void *const *d_polys, size_t n_polys, unsigned log_len, const uint64_t *omega, void *d_out,
                                  void *stream) {
    if ((field != H2_FP && field != H2_FQ) || basis < 0 || basis > 2 || !program || n_words == 0 || n_words > (1u << 20) || !d_out ||
        log_len > 30 || (n_consts && !consts) || (n_polys && !d_polys))
        return H2_ERR_ARGS;
    // validate the program: operands in range, stack discipline, rotations and products only where the reference allows them
    const size_t len = (size_t)1 << log_len;
    int depth = 0, max_depth = 0;
    bool has_linear = false;
    for (size_t pc = 0; pc < n_words; ++pc) {
        const uint32_t op = program[pc] & 0xFFu, arg = program[pc] >> 8;
        switch (op) {
            case EV_POLY: {
                if (arg >= n_polys || pc + 1 >= n_words || !d_polys[arg]) return H2_ERR_ARGS;
                const long long shift = (int)program[++pc];
                if (shift <= -(long long)len || shift >= (long long)len) return H2_ERR_ARGS;
                if (basis == 0 && shift != 0) return H2_ERR_ARGS;      // "Can't rotate polynomials in the standard basis" (:519)
                ++depth;
                break;
            }
            case EV_LINEAR:
                has_linear = true;
                [[fallthrough]];
            case EV_CONST:
                if (arg >= n_consts) return H2_ERR_ARGS;
                ++depth;
                break;
            case EV_SCALE:
                if (arg >= n_consts || depth < 1) return H2_ERR_ARGS;
                break;
            case EV_MUL:
                // Mul exists for Ast<_, _, LagrangeCoeff> and Ast<_, _, ExtendedLagrangeCoeff> (evaluator.rs:370-418)
                if (basis == 0) return H2_ERR_ARGS;
                [[fallthrough]];
            case EV_ADD:
                if (depth < 2) return H2_ERR_ARGS;
                --depth;
                break;
            case EV_MULADD:
                if (arg >= n_consts || depth < 2) return H2_ERR_ARGS;
                --depth;
                break;
            default:
                return H2_ERR_ARGS;
        }
        max_depth = std::max(max_depth, depth);
    }
    if (depth != 1 || max_depth > kEvalDepth + 1) return H2_ERR_ARGS;
    if (has_linear && basis != 0 && !omega) return H2_ERR_ARGS;
    int rc = ensure_device();
    if (rc != H2_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const uint32_t *d_tw = nullptr;
    if (has_linear && basis != 0 && log_len >= 1) {
        if ((rc = ntt_twiddle_table(field, (int)log_len, omega, st, &d_tw)) != H2_OK) return rc;
    }
    EvalContext &cx = g_eval_ctxs.get(st);
    std::lock_guard<std::mutex> lk(cx.mu);
    if ((rc = cx.prog.reserve(n_words * 4)) != H2_OK || (rc = cx.consts.reserve(n_consts * 32 + 32)) != H2_OK ||
        (rc = cx.ptrs.reserve(n_polys * 8 + 8)) != H2_OK)
        return rc;
    // the staging buffers belong to this (device, stream): the copies are stream-ordered after the kernel that last read them
    H2_HIP(hipMemcpyAsync(cx.prog.ptr, program, n_words * 4, hipMemcpyHostToDevice, st));
    if (n_consts) H2_HIP(hipMemcpyAsync(cx.consts.ptr, consts, n_consts * 32, hipMemcpyHostToDevice, st));
    if (n_polys) H2_HIP(hipMemcpyAsync(cx.ptrs.ptr, d_polys, n_polys * 8, hipMemcpyHostToDevice, st));
    const dim3 grid((unsigned)((len + kEvalThreads - 1) / kEvalThreads)), block(kEvalThreads);
    const size_t lds = (size_t)kEvalDepth * 8 * kEvalThreads * 4;
    if (field == H2_FP)
        hipLaunchKernelGGL((ev_run<FP>), grid, block, lds, st, cx.prog.as<u32>(), (u32)n_words, cx.consts.as<u32>(),
                           (const u32 *const *)cx.ptrs.ptr, log_len, basis, d_tw, (u32 *)d_out);
    else
        hipLaunchKernelGGL((ev_run<FQ>), grid, block, lds, st, cx.prog.as<u32>(), (u32)n_words, cx.consts.as<u32>(),
                           (const u32 *const *)cx.ptrs.ptr, log_len, basis, d_tw, (u32 *)d_out);
    H2_HIP(hipGetLastError());
    return H2_OK;
}
