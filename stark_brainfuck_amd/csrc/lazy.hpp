// lazy.hpp -- unreduced dot products  sum_k a_k * b_k  (a_k: one residue per lane, b_k: wave-uniform) for the weighted sums of
// the non-linear combination (/root/reference/code/brainfuck_stark.py:236-300: sum over terms of weight * codeword).
// The reference reduces after every multiplication and every addition because its elements are boxed integers mod p; the sum
// is the same field element whenever the reduction happens, so here the 64 x 64 products are accumulated as three 64-bit
// COLUMNS (a_lo*b_lo | a_lo*b_hi + a_hi*b_lo | a_hi*b_hi) plus a carry counter per column, and reduced once per sum:
// 8 VALU instructions per product (4 multiply-adds + 4 add-with-carry) against 19 + 6 for gl_mul + gl_add.
#pragma once
#include "gl.hpp"

namespace bfs {

struct LazyAcc {
    u64 c0, c1, c2;      // column sums mod 2^64
    u32 t0, t1, t2;      // how often each column wrapped (up to 2^32 - 1 products: far more than a combination has terms)
};

BFS_HD LazyAcc lazy_zero() { return LazyAcc{0, 0, 0, 0, 0, 0}; }

#if defined(__HIP_DEVICE_COMPILE__)
// acc += a * b.  b must be wave-uniform (it is read through the scalar unit: "s" operands).  The four carries live in four
// SGPR pairs and every carry has three VALU instructions between its producer and its consumer, which covers the two wait
// states gfx950 wants between a VALU write of an SGPR and a VALU read of it as carry-in -- no s_nop in the sequence.
__device__ __forceinline__ void lazy_mac(LazyAcc& L, u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 k0, k1, k2, k3;
    asm("v_mad_u64_u32 %0, %6, %10, %12, %0\n\t"
        "v_mad_u64_u32 %1, %7, %10, %13, %1\n\t"
        "v_mad_u64_u32 %2, %8, %11, %13, %2\n\t"
        "v_mad_u64_u32 %1, %9, %11, %12, %1\n\t"
        "v_addc_co_u32_e64 %3, %6, 0, %3, %6\n\t"
        "v_addc_co_u32_e64 %4, %7, 0, %4, %7\n\t"
        "v_addc_co_u32_e64 %5, %8, 0, %5, %8\n\t"
        "v_addc_co_u32_e64 %4, %9, 0, %4, %9"
        : "+v"(L.c0), "+v"(L.c1), "+v"(L.c2), "+v"(L.t0), "+v"(L.t1), "+v"(L.t2), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3)
        : "v"(a0), "v"(a1), "s"(b0), "s"(b1));
}
// the same with b in vector registers: a weight read from LDS (every lane reads the same word: a broadcast), or a per-lane factor
__device__ __forceinline__ void lazy_mac_v(LazyAcc& L, u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 k0, k1, k2, k3;
    asm("v_mad_u64_u32 %0, %6, %10, %12, %0\n\t"
        "v_mad_u64_u32 %1, %7, %10, %13, %1\n\t"
        "v_mad_u64_u32 %2, %8, %11, %13, %2\n\t"
        "v_mad_u64_u32 %1, %9, %11, %12, %1\n\t"
        "v_addc_co_u32_e64 %3, %6, 0, %3, %6\n\t"
        "v_addc_co_u32_e64 %4, %7, 0, %4, %7\n\t"
        "v_addc_co_u32_e64 %5, %8, 0, %5, %8\n\t"
        "v_addc_co_u32_e64 %4, %9, 0, %4, %9"
        : "+v"(L.c0), "+v"(L.c1), "+v"(L.c2), "+v"(L.t0), "+v"(L.t1), "+v"(L.t2), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
}
#else
BFS_HD void lazy_mac_v(LazyAcc& L, u64 a, u64 b);
BFS_HD void lazy_mac(LazyAcc& L, u64 a, u64 b) {
    const u64 a0 = (u32)a, a1 = a >> 32, b0 = (u32)b, b1 = b >> 32;
    u64 s;
    s = L.c0 + a0 * b0; L.t0 += s < L.c0; L.c0 = s;
    s = L.c1 + a0 * b1; L.t1 += s < L.c1; L.c1 = s;
    s = L.c2 + a1 * b1; L.t2 += s < L.c2; L.c2 = s;
    s = L.c1 + a1 * b0; L.t1 += s < L.c1; L.c1 = s;
}
BFS_HD void lazy_mac_v(LazyAcc& L, u64 a, u64 b) { lazy_mac(L, a, b); }
#endif

// the canonical residue of  c0 + c1 2^32 + c2 2^64 + t0 2^64 + t1 2^96 + t2 2^128:  assembled as five 32-bit words W0..W4 plus
// what is above them, then 2^128 = -2^32 (mod p) for the top.  Needs the sum to stay below 2^160, i.e. fewer than 2^31 products.
BFS_HD u64 lazy_reduce(const LazyAcc& L) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 k, k2, k3, k4;
    const u32 w0 = (u32)L.c0;
    const u32 w1 = __builtin_addc((u32)(L.c0 >> 32), (u32)L.c1, 0u, &k);
    u32 w2 = __builtin_addc((u32)(L.c1 >> 32), (u32)L.c2, k, &k2);
    u32 w3 = __builtin_addc((u32)(L.c2 >> 32), 0u, k2, &k3);
    u32 w4 = k3;
    w2 = __builtin_addc(w2, L.t0, 0u, &k);
    w3 = __builtin_addc(w3, L.t1, k, &k4);
    w4 = w4 + L.t2 + k4;
    const u64 r = gl_reduce128(((u64)w3 << 32) | w2, ((u64)w1 << 32) | w0);
    return gl_sub(r, (u64)w4 << 32);
#else
    u128 v = (u128)L.c0 + ((u128)L.c1 << 32);                       // < 2^97
    u128 hi = (u128)L.c2 + L.t0 + ((u128)L.t1 << 32) + (u64)(v >> 64);   // weight 2^64, < 2^66
    const u64 lo = (u64)v;
    hi += (u128)L.t2 << 64;
    const u64 w4 = (u64)(hi >> 64);
    const u64 r = gl_reduce128((u64)hi, lo);
    return gl_sub(r, gl_reduce128(0, w4 << 32));
#endif
}

// three accumulators = one extension-field sum
struct LazyX {
    LazyAcc r[3];
};
BFS_HD LazyX lazyx_zero() { return LazyX{{lazy_zero(), lazy_zero(), lazy_zero()}}; }
BFS_HD Xfe lazyx_reduce(const LazyX& L) { return Xfe{{lazy_reduce(L.r[0]), lazy_reduce(L.r[1]), lazy_reduce(L.r[2])}}; }

// Multiplication by a fixed extension element w is a linear map of the three limbs of the other operand:
//     w * (v0 + v1 X + v2 X^2) = [ w0 -w2 -w1 ; w1 w0+w2 w1-w2 ; w2 w1 w0+w2 ] (v0 v1 v2)^T        (X^3 = X - 1)
// so the host lays a weight out as the seven residues below and the lanes only multiply and accumulate -- no subtraction,
// no fold of the degree-3 and degree-4 coefficients.  (Check: row 0 is d0 - d3, row 1 d1 + d3 - d4, row 2 d2 + d4 of xfe_mul.)
constexpr int LAZY_W_BASE = 3;    // weight times a BASE value: the first column of the matrix = the weight's own limbs
constexpr int LAZY_W_EXT = 7;     // w0 w1 w2 -w2 -w1 w0+w2 w1-w2
BFS_HD void lazy_weight_matrix(const Xfe& w, u64* out7) {
    out7[0] = w.c[0]; out7[1] = w.c[1]; out7[2] = w.c[2];
    out7[3] = gl_neg(w.c[2]); out7[4] = gl_neg(w.c[1]);
    out7[5] = gl_add(w.c[0], w.c[2]); out7[6] = gl_sub(w.c[1], w.c[2]);
}
// L += w * v for a base value v (w: 3 words) and for an extension value v (w: the 7 words above); `w` must be wave-uniform
// (the weights are read from LDS -- see air.hip: as kernel arguments the compiler fetched all ~600 words up front and spilled them)
template <class W>
BFS_HD void lazyx_mac_base(LazyX& L, W w, u64 v) {
    lazy_mac_v(L.r[0], v, w[0]);
    lazy_mac_v(L.r[1], v, w[1]);
    lazy_mac_v(L.r[2], v, w[2]);
}
template <class W>
BFS_HD void lazyx_mac_ext(LazyX& L, W w, const Xfe& v) {
    lazy_mac_v(L.r[0], v.c[0], w[0]); lazy_mac_v(L.r[0], v.c[1], w[3]); lazy_mac_v(L.r[0], v.c[2], w[4]);
    lazy_mac_v(L.r[1], v.c[0], w[1]); lazy_mac_v(L.r[1], v.c[1], w[5]); lazy_mac_v(L.r[1], v.c[2], w[6]);
    lazy_mac_v(L.r[2], v.c[0], w[2]); lazy_mac_v(L.r[2], v.c[1], w[1]); lazy_mac_v(L.r[2], v.c[2], w[5]);
}
// L += s * v for a per-lane base factor s and an extension value v
BFS_HD void lazyx_mac_scale(LazyX& L, const Xfe& v, u64 s) {
    lazy_mac_v(L.r[0], v.c[0], s);
    lazy_mac_v(L.r[1], v.c[1], s);
    lazy_mac_v(L.r[2], v.c[2], s);
}

}  // namespace bfs
