// Square roots in the Pasta base fields (Tonelli-Shanks: p - 1 = 2^32 T for both) and the curve constant b = 5, shared by the
// point codec (points.hip: `from_bytes`) and the hash-to-curve map (h2c.hip: simplified SWU).
#pragma once
#include "field.cuh"

namespace h2 {

template <int F> __device__ __forceinline__ fe sqrt_root_of_unity() {   // 5^T, order 2^32, Montgomery
    if (F == FP) return fe{{0xbad6dbf0u, 0xa28db849u, 0xd3b539dfu, 0x9083cd03u, 0x9dc8448eu, 0xfba6b9cau, 0x7b89c6dau, 0x3ec92874u}};
    return fe{{0x8c9942deu, 0x21807742u, 0x21b60494u, 0xcc495789u, 0xb2efbee2u, 0xac2e5d27u, 0x7f2db056u, 0x0b79fa89u}};
}
template <int F> __device__ __forceinline__ fe curve_b() {              // 5, Montgomery (y^2 = x^3 + 5 on both curves)
    if (F == FP) return fe{{0xffffffedu, 0xa1a55e68u, 0x4f4982f3u, 0x74c2a54bu, 0xfffffffdu, 0xffffffffu, 0xffffffffu, 0x3fffffffu}};
    return fe{{0xffffffedu, 0x96bc8c8cu, 0x49f7778eu, 0x74c2a54bu, 0xfffffffdu, 0xffffffffu, 0xffffffffu, 0x3fffffffu}};
}

// square root of a Montgomery element; false when `a` is a non-residue
template <int F> __device__ bool fe_sqrt(const fe &a, fe &out) {
    if (fe_is_zero(a)) {
        out = a;
        return true;
    }
    // (T - 1) / 2, T = (p - 1) / 2^32
    const u32 e[8] = {F == FP ? 0xcc969876u : 0xc6237590u, F == FP ? 0x04a67c8du : 0x04ca546eu, 0x11234c7eu, 0, 0, 0, 0x20000000u, 0};
    const fe one = fe_one<F>();
    fe w = one;
    for (int i = 221; i >= 0; --i) {
        w = fe_sqr<F>(w);
        if ((e[i >> 5] >> (i & 31)) & 1) w = fe_mulx<F>(w, a);
    }
    fe x = fe_mulx<F>(a, w);   // a^((T+1)/2)
    fe b = fe_mulx<F>(x, w);   // a^T, in the 2^32-torsion
    fe z = sqrt_root_of_unity<F>();
    int v = 32;
    while (!fe_eq(b, one)) {
        int k = 0;
        fe t = b;
        while (!fe_eq(t, one)) {           // least k with b^(2^k) = 1
            t = fe_sqr<F>(t);
            if (++k == v) return false;    // order 2^v: a is not a square
        }
        fe ww = z;
        for (int i = 0; i < v - k - 1; ++i) ww = fe_sqr<F>(ww);
        z = fe_sqr<F>(ww);
        b = fe_mulx<F>(b, z);
        x = fe_mulx<F>(x, ww);
        v = k;
    }
    out = x;
    return true;
}

}  // namespace h2
