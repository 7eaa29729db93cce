// gl.hpp -- arithmetic in F_p, p = 2^64 - 2^32 + 1, and in F_p[X]/(X^3 - X + 1), for gfx950 device code
// and for the host-side planner.  Replaces the reference's boxed-bigint element classes:
//   BaseField.add/subtract/multiply/negate/inverse      /root/reference/code/algebra.py:89-108
//   BaseFieldElement.__xor__ (square-and-multiply)       algebra.py:39-46
//   ExtensionField.multiply/add/subtract/inverse         extension_field.py:65-86
// All values are canonical residues in [0, p) stored as uint64_t; every function returns canonical values,
// so results are bit-identical to the reference's Python ints.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BFS_HD __host__ __device__ __forceinline__
#else
#define BFS_HD inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define BFS_UNROLL _Pragma("unroll")
#else
#define BFS_UNROLL
#endif

namespace bfs {

typedef uint64_t u64;
typedef uint32_t u32;
typedef unsigned __int128 u128;

constexpr u64 GL_P = 0xFFFFFFFF00000001ULL;
constexpr u64 GL_EPS = 0xFFFFFFFFULL;  // 2^64 mod p = 2^32 - 1

#if defined(__HIP_DEVICE_COMPILE__)
// ---- gfx950 code generation notes (measured with tools/microbench, DESIGN.md "modular arithmetic") ----
// 64-bit compares / selects cost twice a 32-bit VALU op and hipcc recomputes carries with v_cmp_*_u64; the
// __builtin_subc / __builtin_addc forms below compile to v_sub_co_u32 / v_subb_co_u32 carry chains instead
// (5 VALU instructions for a canonical modular subtraction).
// a - b in FOUR instructions (gl_sub4; a translation unit asks for it as its gl_sub with BFS_GL_SUB4 before including this header):
// d = a - b with its borrow B kept in an SGPR pair; a borrow means + p = - EPS = + 1 - 2^32, so the low word becomes d_lo + B (an
// add-with-carry whose carry-IN is B and whose carry-out C says d_lo was 0xFFFFFFFF) and the high word d_hi - (B and not C) (a
// subtract-with-borrow whose borrow-in is that scalar combination).  hipcc keeps every carry in VCC and materialises the borrow as
// a mask first (v_cndmask + v_sub_co + v_subb_co: five instructions, gl_sub5).  In the NTT tile kernels the asm form is 6.9 % fewer
// VALU instructions per launch (profiles/r03/ab_sub4.txt); in the generated constraint code of air.hip its fixed register demands cost
// the allocator more than the instruction saves (air_quotient_kernel<0>: 144 -> 243 VGPRs), so that unit keeps the compiler's form.
BFS_HD u64 gl_sub4(u64 a, u64 b) {
    u32 rlo, rhi;                                      // (the high word's register may be one of the inputs: they are all read by then)
    u64 sb, sc;
    asm("v_sub_co_u32 %0, vcc, %4, %6\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32 %1, %2, %5, %7, vcc\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %0, %3, %0, 0, %2\n\t"
        "s_nop 1\n\t"
        "s_andn2_b64 %2, %2, %3\n\t"
        "v_subb_co_u32 %1, vcc, %1, 0, %2"
        : "=&v"(rlo), "=v"(rhi), "=&s"(sb), "=&s"(sc)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32))
        : "vcc", "scc");                               // s_andn2_b64 sets SCC: undeclared, loops around the asm lost their exit condition
    return ((u64)rhi << 32) | rlo;
}
// lo - hh (a 32-bit subtrahend, any 64-bit lo) the same way: the first step of the 128-bit reduction
BFS_HD u64 gl_sub_word4(u64 lo, u32 hh) {
    u32 rlo, rhi;
    u64 sb, sc;
    asm("v_sub_co_u32 %0, vcc, %4, %6\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32 %1, %2, %5, 0, vcc\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %0, %3, %0, 0, %2\n\t"
        "s_nop 1\n\t"
        "s_andn2_b64 %2, %2, %3\n\t"
        "v_subb_co_u32 %1, vcc, %1, 0, %2"
        : "=&v"(rlo), "=v"(rhi), "=&s"(sb), "=&s"(sc)
        : "v"((u32)lo), "v"((u32)(lo >> 32)), "v"(hh)
        : "vcc", "scc");
    return ((u64)rhi << 32) | rlo;
}
BFS_HD u64 gl_sub5(u64 a, u64 b) {
    u32 bl, bh, b2;
    u32 dlo = __builtin_subc((u32)a, (u32)b, 0u, &bl);
    u32 dhi = __builtin_subc((u32)(a >> 32), (u32)(b >> 32), bl, &bh);
    u32 t = 0u - bh;                                   // EPS when the subtraction borrowed, else 0
    u32 rlo = __builtin_subc(dlo, t, 0u, &b2);
    u32 rhi = dhi - b2;
    // hipcc (ROCm 7.2) folds this plain subtraction of a borrow into a following add-with-carry of the consumer and then
    // reads the MERGED instruction's carry-out, which is wrong: gl_add(gl_sub(a, b), 1) came out as a - b + 1 mod 2^64
    // (the fold needs a consumer whose own high-word addend is a compile-time zero).  The empty asm makes the value opaque
    // to that combine; it emits no instruction.  Checked by csrc/selftest.hip (bfs_selftest_field) on the device.
    asm("" : "+v"(rhi));
    return ((u64)rhi << 32) | rlo;
}
#ifdef BFS_GL_SUB4
constexpr bool GL_SUB4 = true;
BFS_HD u64 gl_sub(u64 a, u64 b) { return gl_sub4(a, b); }
#else
constexpr bool GL_SUB4 = false;
BFS_HD u64 gl_sub(u64 a, u64 b) { return gl_sub5(a, b); }
#endif
// the reductions' scalar-carry forms (gl_sub_word4 for lo - hi_hi, gl_fold_word<true> for the tail): everywhere unless a unit opts out
constexpr bool GL_REDUCE4 = true;
// a + b: s = a + b and u = s + EPS (= s - p mod 2^64) with both carries; a + b >= p  <=>  either addition carried.  Six instructions
// as the compiler has them.  Five-instruction forms with scalar carries were measured twice (profiles/r03/ab_add5.txt; profiles/r04/
// ab_add5_power.txt: 4 % fewer VALU instructions per NTT step, the step 0.2-1.3 % SLOWER: each trades two selects for a 64-bit
// multiply-add, and the step runs at the package power limit) -- not kept.
BFS_HD u64 gl_add(u64 a, u64 b) {
    u32 c1, c2, c3, c4;
    u32 slo = __builtin_addc((u32)a, (u32)b, 0u, &c1);
    u32 shi = __builtin_addc((u32)(a >> 32), (u32)(b >> 32), c1, &c2);
    u32 ulo = __builtin_addc(slo, 0xFFFFFFFFu, 0u, &c3);
    u32 uhi = __builtin_addc(shi, 0u, c3, &c4);
    const bool over = (c2 | c4) != 0;
    return over ? (((u64)uhi << 32) | ulo) : (((u64)shi << 32) | slo);
}

// a + b for ANY 64-bit a and canonical b: congruent to a + b, in [0, 2^64), not necessarily canonical -- FOUR instructions with scalar
// carries (gl_add: six).  s = a + b with carry K; a wrap means + 2^64 = + EPS, which cannot wrap again because b < p: low word s_lo - K
// (borrow B: s_lo was 0), high word s_hi + (K and not B).  Such a value may be a minuend (gl_sub4), the first operand of another lazy
// sum, an operand of mul_pow2 or of a product, and may be stored for a later pass that multiplies what it loads; it must never be a
// subtrahend, the second operand of a sum, or an operand of gl_add (ntt_core.hpp: dif_level says which sums qualify; the host
// emulation asserts the rule on every operand, tests/emu/build_emu.py -DBFS_CHECK_CANONICAL).
BFS_HD u64 gl_add_lazy(u64 a, u64 b) {
    u32 rlo, rhi;
    u64 sk, sb;
    asm("v_add_co_u32 %0, vcc, %4, %6\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %1, %2, %5, %7, vcc\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32 %0, %3, %0, 0, %2\n\t"
        "s_nop 1\n\t"
        "s_andn2_b64 %2, %2, %3\n\t"
        "v_addc_co_u32 %1, vcc, %1, 0, %2"
        : "=&v"(rlo), "=&v"(rhi), "=&s"(sk), "=&s"(sb)
        : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32))
        : "vcc", "scc");
    return ((u64)rhi << 32) | rlo;
}

// a VGPR holding 0 that the optimiser cannot see through: `x - 0 - borrow` written with it compiles to one v_subb_co_u32,
// while a literal 0 makes hipcc materialise the borrow with v_cndmask first (one more instruction per carry step)
BFS_HD u32 gl_opaque_zero() {
    u32 z;
    asm("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

// base + w * 2^64 = base + w * EPS  ->  canonical residue, for ANY 64-bit base and 32-bit w: the tail of both reductions.
// q = base + w * EPS mod 2^64 with carry C (one multiply-add).  C set: the value is q + EPS, which neither wraps nor reaches p
// (q < w * EPS <= 2^64 - 2^33 + 1).  C clear: q may be >= p, i.e. the value is q + EPS mod 2^64 exactly when that addition carries (K).
// MERGED: both cases are "+ EPS when C or K" -- a second multiply-add for K, the scalar OR of the two carries, one mask, one
// multiply-add that adds it: 4 VALU instructions.  The separate form (wrap fix, then canonical form: carry, mask, add twice) is 6.
template <bool MERGED>
BFS_HD u64 gl_fold_word(u32 w, u64 base) {
    if constexpr (MERGED) {
        u64 q, r, sc;
        u32 m;
        asm("v_mad_u64_u32 %0, %3, %4, -1, %5\n\t"
            "v_mad_u64_u32 %1, vcc, -1, 1, %0\n\t"
            "s_or_b64 vcc, vcc, %3\n\t"
            "v_cndmask_b32 %2, 0, -1, vcc\n\t"
            "v_mad_u64_u32 %1, vcc, %2, 1, %0"
            : "=&v"(q), "=&v"(r), "=&v"(m), "=&s"(sc) : "v"(w), "v"(base) : "vcc", "scc");
        return r;
    } else {
        u64 q, q1, r;
        u32 m;
        asm("v_mad_u64_u32 %0, vcc, %4, -1, %5\n\ts_nop 1\n\tv_cndmask_b32 %3, 0, -1, vcc\n\tv_mad_u64_u32 %1, vcc, %3, 1, %0\n\t"
            "v_mad_u64_u32 %0, vcc, -1, 1, %1\n\ts_nop 1\n\tv_cndmask_b32 %3, 0, -1, vcc\n\tv_mad_u64_u32 %2, vcc, %3, 1, %1"
            : "=&v"(q), "=&v"(q1), "=&v"(r), "=&v"(m) : "v"(w), "v"(base) : "vcc");
        return r;
    }
}

// hi*2^64 + lo  ->  residue; CANON = false leaves the value in [0, 2^64) (fine as an operand of further multiplications).
// 8 VALU instructions in the scalar-carry forms (gl_sub_word4 + gl_fold_word<true>: 4 + 4; 11 in the split forms, 13 with add / add-with-carry pairs, 17 as
// plain C): the multiply-add  hi_lo * (2^32 - 1) + t0  is ONE v_mad_u64_u32 whose carry-out is used
// directly -- C has no way to ask for that carry, and the compiler's version is mad + 64-bit add + 64-bit compare.
template <bool CANON, bool SUB4 = GL_REDUCE4>
BFS_HD u64 gl_reduce128_t(u64 hi, u64 lo) {
    u32 hh = (u32)(hi >> 32), hl = (u32)hi;
    u64 t0;                                            // t0 = lo - hi_hi  (2^96 = -1)
    if constexpr (SUB4) {
        t0 = gl_sub_word4(lo, hh);
    } else {
        const u32 z = gl_opaque_zero();
        u32 bl, bh, b2, b3;
        u32 dlo = __builtin_subc((u32)lo, hh, 0u, &bl);
        u32 dhi = __builtin_subc((u32)(lo >> 32), z, bl, &bh);
        u32 t = 0u - bh;
        u32 t0lo = __builtin_subc(dlo, t, 0u, &b2);
        u32 t0hi = __builtin_subc(dhi, z, b2, &b3);
        t0 = ((u64)t0hi << 32) | t0lo;
    }
    // + hi_lo * (2^32 - 1)  (2^64 = 2^32 - 1): one multiply-add; when that addition wrapped, + EPS (cannot wrap twice), again as a
    // multiply-add  r + mask * 1  on the register pair; and the canonical form (r >= p  <=>  r + EPS carries out of 64 bits) the same
    // way: a multiply-add for the carry, the mask, a multiply-add that adds it.  Every step works on whole 64-bit pairs, so the
    // sequence is 3 (+ 3) instructions where add / add-with-carry / select pairs were 4 (+ 4).
    u64 r;
    if constexpr (!CANON) {
        u64 q;
        u32 m;
        asm("v_mad_u64_u32 %0, vcc, %3, -1, %4\n\ts_nop 1\n\tv_cndmask_b32 %2, 0, -1, vcc\n\tv_mad_u64_u32 %1, vcc, %2, 1, %0"
            : "=&v"(q), "=&v"(r), "=&v"(m) : "v"(hl), "v"(t0) : "vcc");
        return r;
    } else {
        return gl_fold_word<SUB4>(hl, t0);
    }
}
BFS_HD u64 gl_reduce128(u64 hi, u64 lo) { return gl_reduce128_t<true>(hi, lo); }

// lo + top * 2^64 for a 32-bit top  ->  canonical residue (x << r for r < 32 is such a 96-bit value): the tail of gl_reduce128_t
BFS_HD u64 gl_reduce96(u32 top, u64 lo) { return gl_fold_word<GL_REDUCE4>(top, lo); }

// 64 x 64 -> 128 in 8 instructions: the two middle products are added by the multiply-add itself (ah*bl + al*bh in ONE
// v_mad_u64_u32, whose carry-out -- the 2^64 of that sum -- is picked up at once; C cannot ask for it), which leaves one add for
// the low half and two add-with-carry steps for the high one.  (Four independent products and a carry tree were 10.)
BFS_HD void gl_mul128(u64 a, u64 b, u64& hi, u64& lo) {
    const u32 al = (u32)a, ah = (u32)(a >> 32), bl = (u32)b, bh = (u32)(b >> 32);
    const u64 A = (u64)al * bl, B = (u64)ah * bh, M1 = (u64)al * bh;
    u64 M;                                             // ah*bl + al*bh mod 2^64
    u32 k;                                             // and its carry
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %4\n\ts_nop 1\n\tv_cndmask_b32 %1, 0, 1, vcc" : "=&v"(M), "=v"(k) : "v"(ah), "v"(bl), "v"(M1) : "vcc");
    u32 k2, c3, c4;
    u32 l1 = __builtin_addc((u32)(A >> 32), (u32)M, 0u, &k2);
    u32 h0 = __builtin_addc((u32)B, (u32)(M >> 32), k2, &c3);
    u32 h1 = __builtin_addc((u32)(B >> 32), k, c3, &c4);
    lo = ((u64)l1 << 32) | (u32)A;
    hi = ((u64)h1 << 32) | h0;
}
// Product and reduction in one piece: the carry K of the middle sum ah*bl + al*bh (weight 2^96 = -1) never becomes a register -- it stays
// in the scalar pair the multiply-add wrote it to and enters the reduction's first subtraction  lo - hi_hi  as its borrow-IN
// (gl_mul128 spends a v_cndmask on it and adds it into hi_hi).  hi_hi + K is the true top word, so nothing overflows.  15 instructions.
BFS_HD u64 gl_mul_fused(u64 a, u64 b) {
    const u32 al = (u32)a, ah = (u32)(a >> 32), bl = (u32)b, bh = (u32)(b >> 32);
    const u64 A = (u64)al * bl, B = (u64)ah * bh, M1 = (u64)al * bh;
    u64 M, sk;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=&v"(M), "=s"(sk) : "v"(ah), "v"(bl), "v"(M1));
    u32 k2, c3, c4;
    const u32 z = gl_opaque_zero();
    u32 l1 = __builtin_addc((u32)(A >> 32), (u32)M, 0u, &k2);
    u32 h0 = __builtin_addc((u32)B, (u32)(M >> 32), k2, &c3);
    u32 h1 = __builtin_addc((u32)(B >> 32), z, c3, &c4);            // hi_hi without K
    u32 rlo, rhi;
    u64 sb, sc;
    // (s_nop 1 first: K was written by a VALU instruction in ANOTHER asm statement, and gfx950 wants two wait states between a VALU
    //  write of an SGPR and a VALU read of it as a carry; the compiler cannot see either side, and for a constant operand it has been
    //  seen to leave a single s_nop 0 between the two -- round-3 advice; tools/isa_hazards.py scans the shipped ISA for this)
    asm("s_nop 1\n\t"
        "v_subb_co_u32 %0, vcc, %4, %6, %7\n\t"                    // lo_lo - hi_hi - K
        "s_nop 1\n\t"
        "v_subb_co_u32 %1, %2, %5, 0, vcc\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 %0, %3, %0, 0, %2\n\t"
        "s_nop 1\n\t"
        "s_andn2_b64 %2, %2, %3\n\t"
        "v_subb_co_u32 %1, vcc, %1, 0, %2"
        : "=&v"(rlo), "=&v"(rhi), "=&s"(sb), "=&s"(sc)
        : "v"((u32)A), "v"(l1), "v"(h1), "s"(sk)
        : "vcc", "scc");
    return gl_fold_word<true>(h0, ((u64)rhi << 32) | rlo);
}
BFS_HD u64 gl_mul(u64 a, u64 b) {
    if constexpr (GL_REDUCE4) return gl_mul_fused(a, b);
    u64 hi, lo;
    gl_mul128(a, b, hi, lo);
    return gl_reduce128_t<true>(hi, lo);
}
// product left in [0, 2^64): only for values whose next use is another multiplication
BFS_HD u64 gl_mul_lazy(u64 a, u64 b) {
    u64 hi, lo;
    gl_mul128(a, b, hi, lo);
    return gl_reduce128_t<false>(hi, lo);
}
// the product through the OTHER form of the reduction's first step (selftest.hip: both forms are compared with the host on the device)
BFS_HD u64 gl_mul_other_form(u64 a, u64 b) {
    u64 hi, lo;
    gl_mul128(a, b, hi, lo);
    return gl_reduce128_t<true, !GL_REDUCE4>(hi, lo);
}
#else
// -DBFS_CHECK_CANONICAL (the host emulation of the kernels, tests/emu): every operand that has to be canonical is checked -- the NTT
// keeps some sums unreduced (gl_add_lazy) and the rule which operands may be such values is enforced here on every emulated transform:
// violations are counted (tests/emu: emu_canonical_violations) and the tests want zero.  An unreduced sum is only actually >= p
// when it lands within 2^32 of p -- once in 2^32 for random operands -- so the tests that mean it feed values next to 0 and p.
#ifdef BFS_CHECK_CANONICAL
inline unsigned long long& gl_canonical_violations() { static unsigned long long count = 0; return count; }      // (the emulation is single-threaded)
#define BFS_CANONICAL(x) do { if (!((x) < GL_P)) ++gl_canonical_violations(); } while (0)
#else
#define BFS_CANONICAL(x) ((void)0)
#endif
BFS_HD u64 gl_add(u64 a, u64 b) {
    BFS_CANONICAL(a); BFS_CANONICAL(b);
    u64 s = a + b;
    // a, b < p: either the 64-bit add wrapped (then s + EPS is the canonical value) or s may be >= p
    if (s < a) return s + GL_EPS;
    return s >= GL_P ? s - GL_P : s;
}

// any 64-bit a, canonical b
BFS_HD u64 gl_sub(u64 a, u64 b) {
    BFS_CANONICAL(b);
    u64 d = a - b;
    return a < b ? d - GL_EPS : d;  // borrow: add p (== subtract EPS in wrapped arithmetic)
}

// the device's lazy sum (any 64-bit a, canonical b; result in [0, 2^64), not necessarily canonical)
BFS_HD u64 gl_add_lazy(u64 a, u64 b) {
    BFS_CANONICAL(b);
    u64 s = a + b;
    return s < a ? s + GL_EPS : s;
}
BFS_HD u64 gl_sub4(u64 a, u64 b) { return gl_sub(a, b); }      // the device's two instruction sequences (selftest.hip compares both with this)
BFS_HD u64 gl_sub5(u64 a, u64 b) { return gl_sub(a, b); }

// reduce a 128-bit value hi*2^64 + lo.  2^64 = 2^32 - 1, 2^96 = -1 (mod p).
BFS_HD u64 gl_reduce128(u64 hi, u64 lo) {
    u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
    u64 t0 = lo - hi_hi;
    if (lo < hi_hi) t0 -= GL_EPS;          // borrow -> +p ; no second borrow: t0 wrapped >= 2^64 - 2^32 + 1
    u64 t1 = hi_lo * GL_EPS;               // (hi_lo << 32) - hi_lo, fits 64 bits
    u64 r = t0 + t1;
    if (r < t1) r += GL_EPS;               // carry -> +2^64 = +EPS ; cannot carry twice
    return r >= GL_P ? r - GL_P : r;
}
#endif

#if !defined(__HIP_DEVICE_COMPILE__)
BFS_HD u64 gl_reduce96(u32 top, u64 lo) { return gl_reduce128((u64)top, lo); }
#endif

BFS_HD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
// any 64-bit value -> its canonical residue
BFS_HD u64 gl_canon(u64 a) { return a >= GL_P ? a - GL_P : a; }

#if !defined(__HIP_DEVICE_COMPILE__)
BFS_HD u64 gl_mul(u64 a, u64 b) {
    u128 z = (u128)a * b;
    return gl_reduce128((u64)(z >> 64), (u64)z);
}
BFS_HD u64 gl_mul_lazy(u64 a, u64 b) { return gl_mul(a, b); }
BFS_HD u64 gl_mul_other_form(u64 a, u64 b) { return gl_mul(a, b); }
#endif

BFS_HD u64 gl_sqr(u64 a) { return gl_mul(a, a); }

BFS_HD u64 gl_pow(u64 a, u64 e) {
    u64 acc = 1;
    while (e) {
        if (e & 1) acc = gl_mul(acc, a);
        a = gl_sqr(a);
        e >>= 1;
    }
    return acc;
}

// a^(p-2); inverse(0) = 0, which is also what the reference's xgcd-based inverse returns (algebra.py:101-103).
// p - 2 = 2^64 - 2^32 - 1 has 63 one bits, so square-and-multiply costs 64 + 62 products; with t_k = a^(2^k - 1)
// (t_2k = t_k^(2^k) t_k) the exponent is (2^31 - 1) 2^33 + (2^32 - 1): 64 squarings and 10 multiplications.  Intermediate
// squares are only ever multiplied again, so they stay in [0, 2^64) (gl_mul_lazy).
BFS_HD u64 gl_sqr_times(u64 a, int times) {
    for (int k = 0; k < times; ++k) a = gl_mul_lazy(a, a);
    return a;
}
BFS_HD u64 gl_inv(u64 a) {
    const u64 t2 = gl_mul_lazy(gl_sqr_times(a, 1), a);
    const u64 t4 = gl_mul_lazy(gl_sqr_times(t2, 2), t2);
    const u64 t8 = gl_mul_lazy(gl_sqr_times(t4, 4), t4);
    const u64 t16 = gl_mul_lazy(gl_sqr_times(t8, 8), t8);
    const u64 t24 = gl_mul_lazy(gl_sqr_times(t16, 8), t8);
    const u64 t28 = gl_mul_lazy(gl_sqr_times(t24, 4), t4);
    const u64 t30 = gl_mul_lazy(gl_sqr_times(t28, 2), t2);
    const u64 t31 = gl_mul_lazy(gl_sqr_times(t30, 1), a);
    const u64 t32 = gl_mul_lazy(gl_sqr_times(t31, 1), a);
    return gl_mul(gl_sqr_times(t31, 33), t32);
}

// ---- cubic extension, X^3 = X - 1.  Limbs low degree first (c0, c1, c2). ----
struct Xfe {
    u64 c[3];
};

BFS_HD Xfe xfe_add(const Xfe& a, const Xfe& b) { return Xfe{{gl_add(a.c[0], b.c[0]), gl_add(a.c[1], b.c[1]), gl_add(a.c[2], b.c[2])}}; }
BFS_HD Xfe xfe_sub(const Xfe& a, const Xfe& b) { return Xfe{{gl_sub(a.c[0], b.c[0]), gl_sub(a.c[1], b.c[1]), gl_sub(a.c[2], b.c[2])}}; }
BFS_HD Xfe xfe_scale(const Xfe& a, u64 s) { return Xfe{{gl_mul(a.c[0], s), gl_mul(a.c[1], s), gl_mul(a.c[2], s)}}; }
BFS_HD Xfe xfe_neg(const Xfe& a) { return Xfe{{gl_neg(a.c[0]), gl_neg(a.c[1]), gl_neg(a.c[2])}}; }
BFS_HD Xfe xfe_lift(u64 b) { return Xfe{{b, 0, 0}}; }                                            // extension_field.py:113-116
BFS_HD Xfe xfe_add_base(const Xfe& a, u64 b) { return Xfe{{gl_add(a.c[0], b), a.c[1], a.c[2]}}; }
BFS_HD Xfe xfe_sub_base(const Xfe& a, u64 b) { return Xfe{{gl_sub(a.c[0], b), a.c[1], a.c[2]}}; }
BFS_HD Xfe xfe_base_sub(u64 b, const Xfe& a) { return Xfe{{gl_sub(b, a.c[0]), gl_neg(a.c[1]), gl_neg(a.c[2])}}; }

// schoolbook 3x3 then fold X^3 -> X - 1, X^4 -> X^2 - X:  r0 = d0 - d3, r1 = d1 + d3 - d4, r2 = d2 + d4
BFS_HD Xfe xfe_mul(const Xfe& a, const Xfe& b) {
    u64 d0 = gl_mul(a.c[0], b.c[0]);
    u64 d1 = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
    u64 d2 = gl_add(gl_add(gl_mul(a.c[0], b.c[2]), gl_mul(a.c[1], b.c[1])), gl_mul(a.c[2], b.c[0]));
    u64 d3 = gl_add(gl_mul(a.c[1], b.c[2]), gl_mul(a.c[2], b.c[1]));
    u64 d4 = gl_mul(a.c[2], b.c[2]);
    return Xfe{{gl_sub(d0, d3), gl_sub(gl_add(d1, d3), d4), gl_add(d2, d4)}};
}

// inverse by solving the 3x3 system (a * b = 1); host-side use only (Fiat-Shamir scalars), not a hot loop
inline Xfe xfe_inv(const Xfe& a) {
    Xfe x{{1, 0, 0}}, X1{{0, 1, 0}};
    u64 m[3][3];  // m[i][j] = coefficient i of a * X^j
    for (int j = 0; j < 3; ++j) {
        Xfe col = xfe_mul(a, x);
        for (int i = 0; i < 3; ++i) m[i][j] = col.c[i];
        x = xfe_mul(x, X1);
    }
    u64 c00 = gl_sub(gl_mul(m[1][1], m[2][2]), gl_mul(m[1][2], m[2][1]));
    u64 c01 = gl_sub(gl_mul(m[1][2], m[2][0]), gl_mul(m[1][0], m[2][2]));
    u64 c02 = gl_sub(gl_mul(m[1][0], m[2][1]), gl_mul(m[1][1], m[2][0]));
    u64 det = gl_add(gl_add(gl_mul(m[0][0], c00), gl_mul(m[0][1], c01)), gl_mul(m[0][2], c02));
    u64 di = gl_inv(det);
    return Xfe{{gl_mul(c00, di), gl_mul(c01, di), gl_mul(c02, di)}};
}

// algebra.py:122-136: the reference's fixed 2^32-th root of unity squared down to order 2^log_n
inline u64 gl_primitive_root(u32 log_n) {
    u64 r = 1753635133440165772ULL;
    for (u32 k = 32; k > log_n; --k) r = gl_sqr(r);
    return r;
}

}  // namespace bfs
