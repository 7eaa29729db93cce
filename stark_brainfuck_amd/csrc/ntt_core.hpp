// ntt_core.hpp -- tiled mixed-radix Goldilocks NTT, the device-side body of bfs_gl_ntt().
//
// Replaces the recursive object-list transform of the reference:
//   ntt()   /root/reference/code/ntt.py:4-23     out[k] = sum_j v[j] w^(jk), natural order in and out
//   intt()  ntt.py:26-42                          same with w^-1, then * n^-1 (folded into a twiddle table here)
//   Polynomial.scale()  univariate.py:168-169     c_j * s^j, fused into the first pass' load (coset evaluation)
//   zero padding of fast_coset_evaluate  ntt.py:164-168  fused: inputs j >= n_in read as 0, never materialised
//
// Algorithm (DESIGN.md "NTT"): n = n_1 * n_2 * ... * n_m (m <= 4 HBM passes, n_t = 2^S_t, S_t <= 12).
// Pass t transforms digit t of the input index (most significant first).  After pass t the slot that held input
// digit j_t holds output digit k_t, so after all passes slot (p_1|p_2|...|p_m) holds X[p_1 + n_1 p_2 + ...]; the
// last pass writes straight to that natural-order index.  Between passes the element is multiplied by
// w_{N_{t+1}}^(j_{t+1} * K_t) (N_t = n_1..n_t, K_t = k_1 + n_1 k_2 + ...), applied while pass t+1 loads.
// Inside a pass the 2^S-point column transform is again split into <= 3 register stages of radix 2^B <= 16:
// every thread holds 16 elements in VGPRs, runs a decimation-in-frequency network whose twiddles are powers
// of two (any primitive 16th root of unity in this field is 2^(12u), u odd), multiplies by the inner twiddle
// and exchanges through LDS once per stage.
//
// A wave64 integer instruction costs 4 cycles and the tile kernels issue ~2000 of them per 16 elements, which is about what one
// pass' HBM time is worth: the first two passes of a 2^24 transform are memory-bound, the last one VALU-bound.
// So the code spends instructions on arithmetic only: tile shape (LOGC) and pass kind (MODE) are template parameters, every
// global / LDS access is "per-thread base + wave-uniform or immediate offset", digit permutations are wave-uniform scalars,
// strides are powers of two applied as shifts, power-of-two twiddles are shifts and folds.
//
// The stage bodies are pure functions of (thread id, block id, LDS pointer) and compile for the host as well,
// which is how tests/test_emulation.py checks the index arithmetic without a GPU (test infrastructure only:
// the product never runs them on the CPU).
#pragma once
#include "gl.hpp"

namespace bfs {

// 2^k mod p for 0 <= k < 192, evaluated at compile time
constexpr u64 cx_mulmod(u64 a, u64 b) { return (u64)(((u128)a * b) % GL_P); }
constexpr u64 cx_pow2(int k) {
    u64 r = 1;
    for (int i = 0; i < k; ++i) r = cx_mulmod(r, 2);
    return r;
}

// x * 2^K mod p for canonical x, 0 <= K < 96; canonical result.  With K = 32 q + r and (y2:y1:y0) = x << r (96 bits, y2 < 2^r),
// 2^64 = 2^32 - 1 and 2^96 = -1 (mod p) give
//   q = 1:  2^32 (y0 + y1) - (y1 + y2)          q = 2:  2^32 y0 - ((y2:y1) + y0)
// where both operands of the subtraction are canonical by construction (2^32 s mod p = (s_lo : -carry) for a 33-bit s), so
// the whole product is three funnel shifts, two or three additions and one modular subtraction: 10-12 VALU instructions
// instead of 20-26 for a multiplication by the constant 2^K mod p.  For q = 0 the 96-bit value x << r is folded with one
// multiply-add by 2^32 - 1 (gl_reduce96): 11 instructions.
template <int K>
BFS_HD u64 mul_pow2(u64 x) {
    static_assert(K >= 0 && K < 96, "rotation amount");
    constexpr int q = K / 32, r = K % 32;
    if constexpr (K == 0) {
        return x;
    } else if constexpr (q == 0) {
        return gl_reduce96((u32)(x >> (64 - r)), x << r);          // (y2 : y1:y0) = x << r,  2^64 = 2^32 - 1
    } else {
        const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
        u32 y0 = x0, y1 = x1, y2 = 0;
        if constexpr (r != 0) {
            y0 = x0 << r;
            y1 = (u32)(x >> (32 - r));
            y2 = x1 >> (32 - r);
        }
        if constexpr (q == 1) {
            const u64 s = (u64)y0 + y1;
            const u64 w = ((u64)(u32)s << 32) | (u32)(0u - (u32)(s >> 32));
            const u64 u = (u64)y1 + y2;
            return gl_sub(w, u);
        } else {
            const u64 t = (((u64)y2 << 32) | y1) + y0;
            return gl_sub((u64)y0 << 32, t);
        }
    }
}

// one level of the radix-Q decimation-in-frequency network, twiddle w_Q = 2^(192/Q).
// LAZY (A/B switch BFS_NTT_LAZY_SUMS, profiles/r03/ab_lazy_minuend.txt): x[0] of a block is only ever the FIRST operand of the levels
// below it (never a subtrahend), and gl_sub / mul_pow2 / a multiplication take any 64-bit value in that place -- so its sum may stay
// unreduced in [0, 2^64) (gl_add_lazy: 5 instructions instead of 6): 15 of the 32 sums of a radix-16 block.  Every output that
// descends from a lazy x[0] only through sums and untwiddled differences is then non-canonical as well, so only a block whose
// outputs are all multiplied next (the first stage of a pass) can do this.
template <int Q, int I, bool LAZY>
BFS_HD void dif_level(u64* x) {
    if constexpr (I < Q / 2) {
        u64 a = x[I], b = x[I + Q / 2];
        if constexpr (LAZY && I == 0) x[I] = gl_add_lazy(a, b);
        else x[I] = gl_add(a, b);
        x[I + Q / 2] = mul_pow2<(192 / Q) * I>(gl_sub(a, b));
        dif_level<Q, I + 1, LAZY>(x);
    }
}

#ifdef BFS_NTT_LAZY_SUMS
constexpr bool NTT_LAZY_SUMS = true;
#else
constexpr bool NTT_LAZY_SUMS = false;
#endif

// Q-point NTT with root 2^(192/Q); result for output index k is left in x[bitrev(k)].  LAZY: see dif_level (outputs in [0, 2^64))
template <int Q, bool LAZY = false>
BFS_HD void dif(u64* x) {
    if constexpr (Q >= 2) {
        dif_level<Q, 0, LAZY>(x);
        dif<Q / 2, LAZY>(x);
        dif<Q / 2, LAZY>(x + Q / 2);
    }
}

// a layer of power-of-two twiddles with compile-time exponents (what a tile-uniform exponent costs after a uniform branch): register d
// times 2^(3 d (V + 1) mod 96); 15 of 16 registers non-trivial.  Only used by the timing-only switch BFS_ABL_R64.
template <int Q, int V, int D = 0>
BFS_HD void pow2_layer(u64* x) {
    if constexpr (D < Q) {
        x[D] = mul_pow2<(3 * D * (V + 1) + (D ? 5 * V : 0)) % 96>(x[D]);
        pow2_layer<Q, V, D + 1>(x);
    }
}

constexpr u32 cx_bitrev(u32 v, int bits) {
    u32 r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1u) << (bits - 1 - i);
    return r;
}

// output digit held by register m of a radix-2^BITS stage: the network leaves y[bitrev(m)] there, and the true root is
// (2^(192/Q))^u, so y'[k] = y[u k]  ->  k = bitrev(m) * u^-1 (mod Q).  m is a compile-time constant, uinv is wave-uniform.
template <int BITS>
BFS_HD u32 perm_digit(int m, u32 uinv) { return (cx_bitrev((u32)m, BITS) * uinv) & ((1u << BITS) - 1); }

// Streaming accesses.  Every element is read once and written once per pass, so when the data set is larger than the caches
// nothing is gained by keeping it there: the NT = true instantiations mark the data loads and stores non-temporal (the twiddle
// tables stay ordinary, cached reads).  8 x 2^24: 1.45 -> 1.41 ms, the memory-bound first pass 463 -> 437 us
// (profiles/r02/ab_nontemporal.txt); a single 2^24 column, which lives in the 256 MiB Infinity Cache, is 2 % SLOWER with the
// hint, so the launcher only uses it above NTT_STREAMING_BYTES per buffer (ntt.hip).
template <bool NT>
BFS_HD u64 ntt_ld(const u64* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (NT) return __builtin_nontemporal_load(p);
#endif
    return *p;
}
template <bool NT>
BFS_HD void ntt_st(u64* p, u64 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (NT) { __builtin_nontemporal_store(v, p); return; }
#endif
    *p = v;
}

struct NttTables {
    const u64* w_lo;       // w^i,              i < 2^lo_bits
    const u64* w_hi;       // w^(i * 2^lo_bits), i < 2^(log_n - lo_bits)
    u32 lo_bits;
    u32 t_in_log;          // inner table has 2^t_in_log entries: Omega^i, Omega = w^(n / 2^t_in_log)
    const u64* t_in;
    const u64* t_in_last;  // Omega^i * post_scale (used for the last inner twiddle of the final pass)
    const u64* s_lo;       // coset shift s: s^i, i < 2^lo_bits (null when no coset)
    const u64* s_hi;       // s^(i * 2^lo_bits)
    const u64* row;        // load-time product table of this pass (null: chain / scalar / nothing): the tile's row K holds the factors of its 2^S rows
    const u64* srow;       // store-time product table of pass 0 of a balanced plan (PassArgs::sched): the tile's row j_2 holds the factors of its 2^S outputs
};

enum { PASS_COLUMN = 0, PASS_FINAL = 1 };

// everything a pass needs, precomputed on the host (ntt_plan.hpp) so that the kernel does no planning arithmetic
struct PassArgs {
    const u64* in;
    u64* out;
    u64 in_batch_stride, out_batch_stride;
    u64 n_in;            // valid input elements per transform (pass 0 only); the rest reads as zero
    u32 partial;         // pass 0 with n_in < n: loads are predicated
    u32 log_n;
    u32 pass_index;      // t, 0-based
    u32 npass;
    u32 pass_bits;       // S_v packed one byte per pass (no array: kernel-argument arrays indexed at run time go to scratch)
    // column pass: element index = h * 2^(S+logL) + row * 2^logL + l ; a tile is all rows x C consecutive l
    u32 logL;
    u32 lognl;           // log2(L / C): tiles per h
    u32 tw_shift;        // K_prev -> exponent of w_n : kstep = (K << tw_shift) mod n
    // final pass of a multi-pass plan: slot = (((p1 << mid_bits) + mid) << S) + row ; a tile is C consecutive p1 x all rows
    u32 n1_bits, mid_bits, logch;
    u32 uinv;            // u^-1 mod 16 where w^(n/16) = 2^(12u)
    u32 has_coset;       // pass 0: multiply input j by s^j
    u64 coset_delta;     // s^(stride of the stage-1 register index)
    u64 post_scale;      // multiplied in at the final store when the final pass has a single stage
    u32 streaming;       // data loads / stores are non-temporal (the launcher picks the NT instantiation; kept here for the record)
    // Balanced twiddle schedule of a three-pass plan (ntt_plan.hpp, DESIGN.md 4.1 "round 3").  The inter-pass factor in front of pass 3
    // splits, w_N^(j3 (k1 + n1 k2)) = w_N^(j3 k1) * w_{n2 n3}^(j3 k2), and each piece goes where it is cheap AND where there is room:
    //   w_{n1 n2}^(j2 k1)  at pass 1's STORE (j2 is tile-uniform there: one row of a product table in LDS; that pass is memory-bound),
    //   w_N^(j3 k1)        in pass 2 at load: k1 is tile-uniform, j3 is the thread's column, so ONE factor per thread (it commutes with
    //                      the pass' transform, which runs over j2),
    //   w_{n2 n3}^(j3 k2)  at pass 3's load from a tile-uniform row (k2) instead of a 31-product chain per thread.
    u32 sched;
    NttTables tb;
};

BFS_HD u64 tw_pow(const u64* lo, const u64* hi, u32 lo_bits, u64 e) {
    u64 a = lo[e & ((1ull << lo_bits) - 1)];
    u64 b = hi[e >> lo_bits];
    return gl_mul(a, b);
}

// 24-bit x 24-bit multiply (full-rate v_mul_u32_u24; operands here are < 2^12)
BFS_HD u32 mul24(u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return a * b;
#endif
}

BFS_HD u32 pass_bits_of(u32 packed, int v) { return (packed >> (8 * v)) & 0xFFu; }

// digit-reverse the slot digits p_first..p_last (p_first most significant in h) into K = p_first + n_first*(...)
BFS_HD u64 digit_reverse(u64 h, u32 packed_bits, int first, int last) {
    u64 K = 0;
    bool any = false;
    for (int v = last; v >= first; --v) {
        const u32 b = pass_bits_of(packed_bits, v);
        u64 p = h & ((1ull << b) - 1);
        h >>= b;
        K = any ? p + (K << b) : p;
        any = true;
    }
    return K;
}

template <int B1, int B2, int B3, int LOGC, int MODE>
struct TileCfg {
    static constexpr int S = B1 + B2 + B3;
    static constexpr int SH1 = B2 + B3, SH2 = B3;
    static constexpr int U = (B2 == 0) ? 1 : (B3 == 0 ? 2 : 3);
    static constexpr int T = (1 << S) << LOGC;           // elements per tile
    static constexpr int W = T / 16 ? T / 16 : 1;        // threads per tile
    // LDS layout (bank-conflict analysis: ntt_plan.hpp / DESIGN.md):
    //   column pass and single-column tiles: [row][col], + 2^PL words every 256
    //   final pass of a multi-pass plan (lanes run along rows when loading): [col][row], + 1 word per column
    static constexpr bool CMAJOR = (MODE == PASS_FINAL) && (LOGC > 0);
    static constexpr int PL = (MODE == PASS_COLUMN) ? 4 : 1;
    static constexpr int LDS_WORDS = CMAJOR ? (T + (1 << LOGC)) : (T + (((T - 1) >> 8) << PL) + (1 << PL));
    static constexpr int TW_WORDS = (U >= 2) ? (1 << (B1 + B2)) : 0;   // stage-1 -> stage-2 twiddles kept in LDS
};

template <typename Cfg, int LOGC>
BFS_HD u32 lds_addr(u32 r, u32 c) {
    if constexpr (Cfg::CMAJOR) {
        return (c << Cfg::S) + c + r;
    } else {
        const u32 lin = (r << LOGC) + c;
        return lin + ((lin >> 8) << Cfg::PL);
    }
}

// per-tile scalars
struct TileGeom {
    const u64* in;    // transform base (+ batch)
    u64* out;
    u64 row0;         // COLUMN: h*2^(S+logL) + c0 ; FINAL multi: ((c0 << mid_bits) + mid) << S ; single: 0
    u64 kbase;        // COLUMN: exponent step of the inter-pass twiddle ; FINAL: c0 + (kmid << n1_bits)
    u64 K;            // COLUMN: digit-reversed previous output digits (row of the twiddle table)
};

template <typename Cfg, int LOGC, int MODE>
BFS_HD TileGeom tile_geom(const PassArgs& a, u32 bid_x, u32 bid_y) {
    TileGeom g;
    g.in = a.in + (u64)bid_y * a.in_batch_stride;
    g.out = a.out + (u64)bid_y * a.out_batch_stride;
    if constexpr (MODE == PASS_COLUMN) {
        const u64 h = bid_x >> a.lognl;
        const u64 c0 = (u64)(bid_x & ((1u << a.lognl) - 1)) << LOGC;
        g.row0 = (h << (Cfg::S + a.logL)) + c0;
        const u64 K = a.pass_index ? digit_reverse(h, a.pass_bits, 0, (int)a.pass_index - 1) : 0;
        g.kbase = (K << a.tw_shift) & ((1ull << a.log_n) - 1);
        g.K = K;
    } else if (a.npass > 1) {
        const u64 c0 = (u64)(bid_x & ((1u << a.logch) - 1)) << LOGC;
        const u64 mid = bid_x >> a.logch;
        g.row0 = ((c0 << a.mid_bits) + mid) << Cfg::S;
        const u64 kmid = a.mid_bits ? digit_reverse(mid, a.pass_bits, 1, (int)a.npass - 2) : 0;
        g.kbase = c0 + (kmid << a.n1_bits);
    } else {
        g.row0 = 0;
        g.kbase = 0;
    }
    if constexpr (MODE != PASS_COLUMN) g.K = 0;
    return g;
}

// the row of the load-time product table that a tile stages in LDS (null: none): column pass -- row K = the digit-reversed previous
// output digits; final pass of a balanced plan -- row k_2 (digit-reversed `mid`)
template <typename Cfg, int MODE>
BFS_HD const u64* tile_load_row(const PassArgs& a, u32 bid_x) {
    if (a.tb.row == nullptr) return nullptr;
    u64 K;
    if constexpr (MODE == PASS_COLUMN) K = a.pass_index ? digit_reverse((u64)(bid_x >> a.lognl), a.pass_bits, 0, (int)a.pass_index - 1) : 0;
    else K = a.mid_bits ? digit_reverse((u64)(bid_x >> a.logch), a.pass_bits, 1, (int)a.npass - 2) : 0;
    return a.tb.row + (K << Cfg::S);
}
// ... and of the store-time table (pass 1 of a balanced plan): row j_2 = the tile's columns' next digit (the same for all of them)
template <typename Cfg, int LOGC, int MODE>
BFS_HD const u64* tile_store_row(const PassArgs& a, u32 bid_x) {
    if constexpr (MODE != PASS_COLUMN) return nullptr;
    if (a.tb.srow == nullptr) return nullptr;
    const u64 c0 = (u64)(bid_x & ((1u << a.lognl) - 1)) << LOGC;
    const u64 j2 = c0 >> (a.logL - pass_bits_of(a.pass_bits, (int)a.pass_index + 1));
    return a.tb.srow + (j2 << Cfg::S);
}

// store the 2^BQ registers of the last stage.  klow = the already-final lower digits of k_pass, c = column;
// srow (LDS, or null): the store-time factors of the tile's output rows
template <typename Cfg, int LOGC, int MODE, int BQ, bool NT = false>
BFS_HD void final_store(const PassArgs& a, const TileGeom& g, const u64* x, u32 klow, int kshift, u32 c, const u64* srow = nullptr) {
    constexpr int Q = 1 << BQ;
    const bool scale = (Cfg::U == 1) && a.post_scale != 1;
    u64* tp;          // per-thread base; the per-register offset below is wave-uniform
    u32 step_log;
    if constexpr (MODE == PASS_COLUMN) {
        tp = g.out + g.row0 + ((u64)klow << a.logL) + c;
        step_log = (u32)kshift + a.logL;
    } else {
        tp = g.out + g.kbase + c + ((u64)klow << (a.log_n - Cfg::S));
        step_log = (u32)kshift + (a.log_n - Cfg::S);
    }
    BFS_UNROLL
    for (int m = 0; m < Q; ++m) {
#ifdef BFS_ABL_NO_MEM
        if (x[m] != 0x123456789ULL) continue;
#endif
        u64 v = scale ? gl_mul(x[m], a.post_scale) : x[m];
        if (srow != nullptr) v = gl_mul(v, srow[klow + (perm_digit<BQ>(m, a.uinv) << kshift)]);      // (wave-uniform question)
        ntt_st<NT>(tp + ((u64)perm_digit<BQ>(m, a.uinv) << step_log), v);
    }
}

// ---- stage 1: global load (+ coset / inter-pass twiddle), first radix, inner twiddle, LDS write (or final store)
// Split into the load (ntt_stage1_load: address arithmetic and the 2^B1 global loads of a thread) and the rest
// (ntt_stage1_compute), so that the kernel can issue the loads FIRST and copy its twiddle tables to LDS while they are in
// flight (ntt.hip).  sub: which of the 16 / 2^B1 sub-groups of a thread (always 0 for B1 = 4).
template <int B1, int B2, int B3, int LOGC, int MODE>
struct Stage1Pos {
    u32 o, c;         // o: the row bits below this stage's digit, c: column
    u32 step_log;     // element d of this thread sits at tp[d << step_log]
    u64 idx0;         // index of element d = 0 inside the transform (also the n_in predicate and the coset exponent)
};

template <int B1, int B2, int B3, int LOGC, int MODE>
BFS_HD Stage1Pos<B1, B2, B3, LOGC, MODE> stage1_pos(const PassArgs& a, const TileGeom& g, u32 tid, int sub) {
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    Stage1Pos<B1, B2, B3, LOGC, MODE> p;
    const u32 G = (u32)sub * Cfg::W + tid;
    if constexpr (MODE == PASS_COLUMN) {
        p.c = G & ((1u << LOGC) - 1); p.o = G >> LOGC;
        p.idx0 = g.row0 + ((u64)p.o << a.logL) + p.c;
        p.step_log = Cfg::SH1 + a.logL;
    } else {
        p.o = G & ((1u << Cfg::SH1) - 1); p.c = G >> Cfg::SH1;
        p.idx0 = g.row0 + ((u64)p.c << (a.mid_bits + Cfg::S)) + p.o;
        p.step_log = Cfg::SH1;
    }
    return p;
}

template <int B1, int B2, int B3, int LOGC, int MODE, bool NT = false>
BFS_HD void ntt_stage1_load(const PassArgs& a, u32 tid, u32 bid_x, u32 bid_y, int sub, u64* x) {
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    constexpr int Q = 1 << B1;
    const TileGeom g = tile_geom<Cfg, LOGC, MODE>(a, bid_x, bid_y);
    const Stage1Pos<B1, B2, B3, LOGC, MODE> p = stage1_pos<B1, B2, B3, LOGC, MODE>(a, g, tid, sub);
    const u64* tp = g.in + p.idx0;
    if (a.partial) {
        BFS_UNROLL
        for (int d = 0; d < Q; ++d) x[d] = (p.idx0 + ((u64)d << p.step_log) < a.n_in) ? ntt_ld<NT>(tp + ((u64)d << p.step_log)) : 0;
    } else {
        BFS_UNROLL
        for (int d = 0; d < Q; ++d) {
#ifdef BFS_ABL_NO_MEM
            x[d] = (p.idx0 + d) * 0x9E3779B97F4A7C15ULL >> 1;
#else
            x[d] = ntt_ld<NT>(tp + ((u64)d << p.step_log));
#endif
        }
    }
}

// (o, c) of stage-1 thread `tid`: o = the row bits below the stage's digit, c = column (stage1_pos without the tile geometry)
template <int B1, int B2, int B3, int LOGC, int MODE>
BFS_HD void stage1_oc(u32 tid, int sub, u32& o, u32& c) {
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    const u32 G = (u32)sub * Cfg::W + tid;
    if constexpr (MODE == PASS_COLUMN) { c = G & ((1u << LOGC) - 1); o = G >> LOGC; }
    else { o = G & ((1u << Cfg::SH1) - 1); c = G >> Cfg::SH1; }
}

// The stage 1 -> 2 exchange: register m of stage-1 thread `tid` goes to tile word stage1_out_index, and register d of sub-group s
// of stage-2 thread `tid` comes from stage2_in_index (word indices of the padded tile layout, lds_addr).
template <int B1, int B2, int B3, int LOGC, int MODE>
BFS_HD u32 stage1_out_index(const PassArgs& a, u32 tid, int sub, int m) {
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    u32 o, c;
    stage1_oc<B1, B2, B3, LOGC, MODE>(tid, sub, o, c);
    return lds_addr<Cfg, LOGC>((perm_digit<B1>(m, a.uinv) << Cfg::SH1) | o, c);
}

template <int B1, int B2, int B3, int LOGC, int MODE>
BFS_HD u32 stage2_in_index(u32 tid, int s, int d) {
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    const u32 G = (u32)s * Cfg::W + tid;
    const u32 c = G & ((1u << LOGC) - 1);
    const u32 rest = G >> LOGC;
    const u32 f1 = rest & ((1u << B1) - 1), f3 = rest >> B1;
    return lds_addr<Cfg, LOGC>((f1 << Cfg::SH1) | ((u32)d << Cfg::SH2) | f3, c);
}

// stage 1 without the exchange: load-time twiddle, first radix and (U >= 2) the inner twiddle; x[m] is left holding the value
// that stage1_out_index(m) receives.  U == 1: the values are final and are stored.
// tw: dense inner-twiddle table for the stage 1 -> 2 exchange (2^(B1+B2) entries, in LDS on the GPU)
template <int B1, int B2, int B3, int LOGC, int MODE, bool NT = false>
BFS_HD void ntt_stage1_values(const PassArgs& a, const u64* tw, const u64* rowtw, u32 tid, u32 bid_x, u32 bid_y, int sub, u64* x,
                              const u64* srow = nullptr) {
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    constexpr int Q = 1 << B1;
    const TileGeom g = tile_geom<Cfg, LOGC, MODE>(a, bid_x, bid_y);
    const u64 nmask = (1ull << a.log_n) - 1;
    const Stage1Pos<B1, B2, B3, LOGC, MODE> p = stage1_pos<B1, B2, B3, LOGC, MODE>(a, g, tid, sub);
    const u32 o = p.o, c = p.c;
#ifndef BFS_ABL_NO_CHAIN
    if (rowtw != nullptr) {
        // the inter-pass twiddles of this tile are one row of a table (K is the same for the whole tile)
#ifdef BFS_ABL_R64          // timing only (wrong results): see the inner twiddle below
        pow2_layer<Q, 0>(x);
#else
        BFS_UNROLL
        for (int d = 0; d < Q; ++d) x[d] = gl_mul(x[d], rowtw[((u32)d << Cfg::SH1) | o]);
#endif
    } else if (a.sched && a.pass_index > 0) {
        // balanced plan, middle pass: w_N^(j3 k1) -- k1 belongs to the tile, j3 is this thread's column: one factor for all 16 rows
        if constexpr (MODE == PASS_COLUMN) {
            const u64 c0 = g.row0 & ((1ull << a.logL) - 1);
            const u64 f = tw_pow(a.tb.w_lo, a.tb.w_hi, a.tb.lo_bits, (g.K * (c0 + c)) & nmask);
            BFS_UNROLL
            for (int d = 0; d < Q; ++d) x[d] = gl_mul(x[d], f);
        }
    } else if (a.pass_index > 0 || a.has_coset) {
        // factor of row r = (d << SH1) | o is beta^r = gamma * delta^d: a geometric chain per thread
        u64 gam, del;
        if (a.pass_index > 0) {
            const u64 ks = (MODE == PASS_COLUMN) ? g.kbase : (g.kbase + c);
            gam = tw_pow(a.tb.w_lo, a.tb.w_hi, a.tb.lo_bits, ((u64)o * ks) & nmask);
            del = tw_pow(a.tb.w_lo, a.tb.w_hi, a.tb.lo_bits, (ks << Cfg::SH1) & nmask);
        } else {
            gam = tw_pow(a.tb.s_lo, a.tb.s_hi, a.tb.lo_bits, p.idx0);
            del = a.coset_delta;
        }
        u64 f = gam;
        BFS_UNROLL
        for (int d = 0; d < Q; ++d) {
            x[d] = gl_mul(x[d], f);
            if (d + 1 < Q) f = gl_mul_lazy(f, del);   // only ever multiplied again: no canonical form needed
        }
    }
#endif
#ifndef BFS_ABL_NO_DIF
    dif<Q, (Cfg::U >= 2) && NTT_LAZY_SUMS>(x);        // U >= 2: every output is multiplied by an inner twiddle below (or reduced there)
#endif
    if constexpr (Cfg::U == 1) {
        final_store<Cfg, LOGC, MODE, B1, NT>(a, g, x, 0, 0, c, srow);
    } else {
        const u32 i2 = o >> B3;
        BFS_UNROLL
        for (int m = 0; m < Q; ++m) {
            const u32 k1 = perm_digit<B1>(m, a.uinv);                      // wave-uniform
            const u32 e = mul24(i2, k1) & ((1u << (B1 + B2)) - 1);         // exponent of w_M, M = 2^(B1+B2)
#ifdef BFS_ABL_NO_INNER
            x[m] = x[m] + e;
#else
            // register 0 holds output digit 0: its twiddle is w^0, which is 1 unless a post-scale (n^-1 of intt) is folded into the
            // table -- a wave-uniform question, so a forward transform skips that product in its VALU-bound last pass as well
            const bool unit = (m == 0) && (!(Cfg::U == 2 && MODE == PASS_FINAL) || a.post_scale == 1);
#ifdef BFS_ABL_R64
            if (MODE == PASS_FINAL) { if (m == Q - 1) pow2_layer<Q, 3>(x); continue; }
#endif
            if (!unit) x[m] = gl_mul(x[m], tw[e]);
            else if constexpr (NTT_LAZY_SUMS) x[m] = gl_canon(x[m]);      // the one output that skips its (unit) product
#endif
        }
#ifdef BFS_ABL_R64
        // TIMING-ONLY emulation of the 6-bit-digit plan (four radix-64 blocks whose internal 16 x 4 split has power-of-two twiddles with
        // a tile-uniform exponent, three general twiddle layers at bits 6 / 12 / 18; profiles/r03/ab_radix64_emulation.txt): per pass the
        // plan costs  pass 0: P + G,  pass 1: P + G,  pass 2: P + G(24-bit exponent) + P  where this code has  G | G(row) + G |
        // G(chain) + G -- so pass 0 gets one more power-of-two layer here, pass 1's row-table product becomes one (above), and the last
        // pass trades its inner product for two of them.  Results are wrong; the instruction stream is what is measured.
        if (MODE == PASS_COLUMN && a.pass_index == 0) pow2_layer<Q, 1>(x);
        if (MODE == PASS_FINAL) pow2_layer<Q, 2>(x);
#endif
    }
}

template <int B1, int B2, int B3, int LOGC, int MODE, bool NT = false>
BFS_HD void ntt_stage1_compute(const PassArgs& a, u64* smem, const u64* tw, const u64* rowtw, u32 tid, u32 bid_x, u32 bid_y, int sub, u64* x,
                               const u64* srow = nullptr) {
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    ntt_stage1_values<B1, B2, B3, LOGC, MODE, NT>(a, tw, rowtw, tid, bid_x, bid_y, sub, x, srow);
    if constexpr (Cfg::U >= 2) {
        BFS_UNROLL
        for (int m = 0; m < (1 << B1); ++m) smem[stage1_out_index<B1, B2, B3, LOGC, MODE>(a, tid, sub, m)] = x[m];
    }
}

template <int B1, int B2, int B3, int LOGC, int MODE>
BFS_HD void ntt_stage1(const PassArgs& a, u64* smem, const u64* tw, const u64* rowtw, u32 tid, u32 bid_x, u32 bid_y, const u64* srow = nullptr) {
    constexpr int Q = 1 << B1, SG = 16 / Q;
    BFS_UNROLL
    for (int s = 0; s < SG; ++s) {
        u64 x[Q];
        ntt_stage1_load<B1, B2, B3, LOGC, MODE>(a, tid, bid_x, bid_y, s, x);
        ntt_stage1_compute<B1, B2, B3, LOGC, MODE>(a, smem, tw, rowtw, tid, bid_x, bid_y, s, x, srow);
    }
}

// ---- stage 2: LDS read, second radix, (inner twiddle + LDS write) or final store
// ntt_stage2_from: sub-group s of thread `tid` once its 2^B2 values are in x[] (however they got there)
template <int B1, int B2, int B3, int LOGC, int MODE, bool NT = false>
BFS_HD void ntt_stage2_from(const PassArgs& a, u64* smem, u32 tid, u32 bid_x, u32 bid_y, int s, u64* x, const u64* srow = nullptr) {
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    if constexpr (B2 > 0) {
        constexpr int Q = 1 << B2;
        const TileGeom g = tile_geom<Cfg, LOGC, MODE>(a, bid_x, bid_y);
        const u32 G = (u32)s * Cfg::W + tid;
        const u32 c = G & ((1u << LOGC) - 1);
        const u32 rest = G >> LOGC;
        const u32 f1 = rest & ((1u << B1) - 1), f3 = rest >> B1;
#ifndef BFS_ABL_NO_DIF
        dif<Q>(x);
#endif
        if constexpr (Cfg::U == 2) {
            final_store<Cfg, LOGC, MODE, B2, NT>(a, g, x, f1, B1, c, srow);
        } else {
            const u64* tab = (MODE == PASS_FINAL) ? a.tb.t_in_last : a.tb.t_in;
            BFS_UNROLL
            for (int m = 0; m < Q; ++m) {
                const u32 k2 = perm_digit<B2>(m, a.uinv);
                const u32 e = mul24(f3, f1 + (k2 << B1)) & ((1u << Cfg::S) - 1);
                const u64 v = gl_mul(x[m], tab[(u64)e << (a.tb.t_in_log - Cfg::S)]);
                smem[lds_addr<Cfg, LOGC>((f1 << Cfg::SH1) | (k2 << Cfg::SH2) | f3, c)] = v;
            }
        }
    }
}

template <int B1, int B2, int B3, int LOGC, int MODE, bool NT = false>
BFS_HD void ntt_stage2(const PassArgs& a, u64* smem, u32 tid, u32 bid_x, u32 bid_y, const u64* srow = nullptr) {
    if constexpr (B2 > 0) {
        constexpr int Q = 1 << B2, SG = 16 / Q;
        BFS_UNROLL
        for (int s = 0; s < SG; ++s) {
            u64 x[Q];
            BFS_UNROLL
            for (int d = 0; d < Q; ++d) x[d] = smem[stage2_in_index<B1, B2, B3, LOGC, MODE>(tid, s, d)];
            ntt_stage2_from<B1, B2, B3, LOGC, MODE, NT>(a, smem, tid, bid_x, bid_y, s, x, srow);
        }
    }
}

// ---- stage 3: LDS read, third radix, final store
template <int B1, int B2, int B3, int LOGC, int MODE, bool NT = false>
BFS_HD void ntt_stage3(const PassArgs& a, u64* smem, u32 tid, u32 bid_x, u32 bid_y, const u64* srow = nullptr) {
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    if constexpr (B3 > 0) {
        constexpr int Q = 1 << B3, SG = 16 / Q;
        const TileGeom g = tile_geom<Cfg, LOGC, MODE>(a, bid_x, bid_y);
        BFS_UNROLL
        for (int s = 0; s < SG; ++s) {
            const u32 G = (u32)s * Cfg::W + tid;
            const u32 c = G & ((1u << LOGC) - 1);
            const u32 rest = G >> LOGC;
            const u32 f1 = rest & ((1u << B1) - 1), f2 = rest >> B1;
            u64 x[Q];
            BFS_UNROLL
            for (int d = 0; d < Q; ++d) x[d] = smem[lds_addr<Cfg, LOGC>((f1 << Cfg::SH1) | (f2 << Cfg::SH2) | (u32)d, c)];
            dif<Q>(x);
            final_store<Cfg, LOGC, MODE, B3, NT>(a, g, x, f1 + (f2 << B1), B1 + B2, c, srow);
        }
    }
}

// direct O(n^2) transform for n <= 8 (ntt.py:4-23 evaluated literally); one thread per output element
struct SmallArgs {
    const u64* in;
    u64* out;
    u64 in_batch_stride, out_batch_stride;
    u64 n_in;
    u32 log_n;
    u64 root, shift, post_scale;
};

BFS_HD void ntt_small_body(const SmallArgs& a, u32 k, u32 bid_y) {
    const u32 n = 1u << a.log_n;
    if (k >= n) return;
    const u64* in = a.in + (u64)bid_y * a.in_batch_stride;
    u64 wk = gl_pow(a.root, k);
    u64 acc = 0, wjk = 1, sj = 1;
    for (u32 j = 0; j < n; ++j) {
        u64 v = j < a.n_in ? in[j] : 0;
        acc = gl_add(acc, gl_mul(gl_mul(v, sj), wjk));
        wjk = gl_mul(wjk, wk);
        sj = gl_mul(sj, a.shift);
    }
    a.out[(u64)bid_y * a.out_batch_stride + k] = gl_mul(acc, a.post_scale);
}

}  // namespace bfs
