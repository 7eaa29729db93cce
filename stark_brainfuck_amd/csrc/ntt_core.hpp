// ntt_core.hpp -- tiled mixed-radix Goldilocks NTT, the device-side body of bfs_gl_ntt().
//
// Replaces the recursive object-list transform of the reference:
//   ntt()   /root/reference/code/ntt.py:4-23     out[k] = sum_j v[j] w^(jk), natural order in and out
//   intt()  ntt.py:26-42                          same with w^-1, then * n^-1 (folded into a twiddle table here)
//   Polynomial.scale()  univariate.py:168-169     c_j * s^j, fused into the first pass' load (coset evaluation)
//   zero padding of fast_coset_evaluate  ntt.py:164-168  fused: inputs j >= n_in read as 0, never materialised
//
// Algorithm (DESIGN.md "NTT"): n = n_0 * n_1 * ... * n_{m-1} (m <= 4 HBM passes, n_t = 2^S_t, S_t <= 12), input index
// j = (j_0 | j_1 | ... | j_{m-1}) with j_0 most significant, output index k = k_0 + n_0 k_1 + n_0 n_1 k_2 + ...
// Pass 0 (PASS_FIRST) reads the column tiles of digit j_0 -- all n_0 rows x C adjacent values of l = (j_1|...|j_{m-1}) -- transforms
// them into k_0 and writes the TRANSPOSED slot (j_{m-1} | ... | j_1 | k_0): C contiguous rows of n_0 elements.  Every later pass t
// (PASS_COLUMN) then finds the slot layout (j_{m-1} | ... | j_{t+1} | j_t | k_{t-1} | ... | k_0), transforms digit j_t in its slot
// (rows at stride n_0..n_{t-1}, C adjacent values of the finished low part K = k_0 + n_0 k_1 + ...) and leaves k_t there: after
// the last pass the slot index IS the natural output index.  Only pass 0 reads one buffer and writes another; passes 1.. run in
// place on the output, so a call with separate input and output needs no intermediate buffer at all (round 4; before, the LAST
// pass did the transposition and both the first and the last pass were out of place through a library-owned buffer, whose
// placement relative to the caller's buffers decided 10 % of the step: profiles/r03/buffer_placement.txt).
// Before pass t transforms j_t the element is multiplied by w_{N_t}^(j_t * K) (N_t = n_0..n_t); K is the tile's COLUMN index.
// Inside a pass the 2^S-point column transform is again split into <= 3 register stages of radix 2^B <= 16:
// every thread holds 16 elements in VGPRs, runs a decimation-in-frequency network whose twiddles are powers
// of two (any primitive 16th root of unity in this field is 2^(12u), u odd), multiplies by the inner twiddle
// and exchanges through LDS once per stage.
//
// A wave64 integer instruction costs 4 cycles and the tile kernels issue ~1600 of them per 16 elements and pass, which is within 5 % of
// what one pass' HBM time is worth at the clock the package power limit allows: the 8 x 2^24 step sits on both roofs.
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
// OUT_LAZY: the block's outputs may be unreduced 64-bit values (every one of them is multiplied next, or stored for a pass that
// multiplies what it loads).  Then the sums that may stay unreduced (gl_add_lazy: four instructions instead of six) are those whose
// every later use accepts one: index 0 of each block at every level -- below, it is the first operand of the next lazy sum and the
// minuend of a difference, never a second operand -- and all sums of the last level: 15 of the 32 sums of a radix-16 block.  Every
// other operand is canonical by construction: sums with I >= 1 (gl_add), twiddled differences (mul_pow2<K>, K > 0); the untwiddled
// difference of a lazy minuend lands on index 0 of its sub-block, where an unreduced value is allowed again.
// (A first version also kept the sums with 1 <= I < Q/4 unreduced, which then met gl_add as first operands one level down; gl_add adds
// EPS once for "wrapped or >= p" and such an operand can need both.  Random data never shows that -- an unreduced sum is >= p once in
// 2^32 -- but trace columns (0, small counters, p - 1) do: tools/soak_stark.py found a proof with a codeword value >= p.  The emulation
// now counts non-canonical operands on inputs next to 0 and p, tests/test_emulation.py, and the GPU tests transform such inputs too.)
// (Round 3 tried lazy sums with the compiler's six-instruction form and lost; the scalar-carry form is what makes it pay:
// profiles/r04/ab_lazy_sums.txt.)
template <int Q, int I, bool OUT_LAZY>
BFS_HD void dif_level(u64* x) {
    if constexpr (I < Q / 2) {
        u64 a = x[I], b = x[I + Q / 2];
        constexpr bool lazy_sum = OUT_LAZY && (Q == 2 || I == 0);
        if constexpr (lazy_sum) x[I] = gl_add_lazy(a, b);
        else x[I] = gl_add(a, b);
        x[I + Q / 2] = mul_pow2<(192 / Q) * I>(gl_sub(a, b));
        dif_level<Q, I + 1, OUT_LAZY>(x);
    }
}

// Q-point NTT with root 2^(192/Q); result for output index k is left in x[bitrev(k)].  Inputs canonical; outputs canonical unless OUT_LAZY
template <int Q, bool OUT_LAZY = false>
BFS_HD void dif(u64* x) {
    if constexpr (Q >= 2) {
        dif_level<Q, 0, OUT_LAZY>(x);
        dif<Q / 2, OUT_LAZY>(x);
        dif<Q / 2, OUT_LAZY>(x + Q / 2);
    }
}

// dif<Q> for inputs of which only registers 0 .. LIVE-1 can be non-zero -- a zero-padded transform whose input fills a fraction of the
// domain (the shape Table.lde makes: table.py:138-149 evaluates an interpolant of degree ~ n/4 .. n/64 on the FRI domain, ntt.py:164-168).
// While the upper half of a level's block is zero (LIVE <= Q/2) the level neither adds nor subtracts: x[I] stays and x[I + Q/2] is x[I]
// times its shift twiddle; once the block is full the network is the ordinary dif<Q> -- the very calls dif<16> ends in, so the rule which
// sums may stay unreduced is the same.  LIVE = 1 (n_in <= n/16): every output is a copy of x[0]; LIVE = 4: nine shifts and four dif<4>
// instead of 32 sums and 18 shifts.  Registers >= LIVE hold zeros on entry (they were loaded as such) and are overwritten.
template <int Q, int LIVE, bool OUT_LAZY, int I = 0>
BFS_HD void dif_sparse_level(u64* x) {
    if constexpr (I < LIVE && I < Q / 2) {
        x[I + Q / 2] = mul_pow2<(192 / Q) * I>(x[I]);
        dif_sparse_level<Q, LIVE, OUT_LAZY, I + 1>(x);
    }
}
template <int Q, int LIVE, bool OUT_LAZY>
BFS_HD void dif_sparse(u64* x) {
    if constexpr (Q >= 2) {
        if constexpr (LIVE > Q / 2) {
            dif<Q, OUT_LAZY>(x);
        } else {
            dif_sparse_level<Q, LIVE, OUT_LAZY>(x);
            dif_sparse<Q / 2, LIVE, OUT_LAZY>(x);
            dif_sparse<Q / 2, LIVE, OUT_LAZY>(x + Q / 2);
        }
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
    const u64* t_in_last;  // Omega^i * post_scale (the last inner twiddle of the last pass: n^-1 of intt costs nothing)
    const u64* s_lo;       // coset shift s: s^i, i < 2^lo_bits (null when no coset)
    const u64* s_hi;       // s^(i * 2^lo_bits)
    const u64* row;        // load-time product table of the last pass of a balanced plan: the tile's row k_1 holds the factors of its 2^S rows
    const u64* srow;       // store-time product table of pass 0 of a balanced plan: the tile's row j_1 holds the factors of its 2^S outputs
};

// PASS_SINGLE: the whole transform in one tile (n <= 4096, one column, natural order in and out)
// PASS_FIRST:  pass 0 of a multi-pass plan: column tile in, transposed rows out
// PASS_COLUMN: passes 1.. of a multi-pass plan: column tile in, the same slots out
// PASS_EXPAND: the first REAL pass of an expansion plan (ntt_plan.hpp: ntt_make_expand_plan) -- a zero-padded transform whose
//              coefficients fill at most 2^-e of the domain, e >= 4.  The leading digit of the input index is then zero (but for a few
//              "extra" coefficients), its transform is a copy, and the pass that would follow it reads the coefficients straight from
//              the input instead of from the slots a first pass would have written: a column pass whose loads come from `in` (index
//              (row << tw_shift) + h, the same for all C columns of the tile), whose factor carries the coset shift, and which adds the
//              extras' rank-one terms to row 0.  One HBM pass and one launch less than the plain plan.
enum { PASS_COLUMN = 0, PASS_SINGLE = 1, PASS_FIRST = 2, PASS_EXPAND = 3 };

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
    // column tiles (PASS_FIRST, PASS_COLUMN): element index = h * 2^(S+logL) + row * 2^logL + l ; a tile is all rows x C consecutive l.
    // pass 0: logL = log n - S_0 (h = 0), l = the remaining input digits; pass t >= 1: logL = S_0 + .. + S_{t-1}, l = K = the finished
    // output digits, h = the input digits still to come
    u32 logL;
    u32 lognl;           // log2(L / C): tiles per h
    u32 tw_shift;        // pass t >= 1: w_{N_t} = w^(2^tw_shift), tw_shift = log n - logL - S_t
    u32 uinv;            // u^-1 mod 16 where w^(n/16) = 2^(12u)
    u32 has_coset;       // pass 0: multiply input j by s^j
    u64 coset_delta;     // s^(stride of the stage-1 register index)
    u64 post_scale;      // last pass: n^-1 of intt (folded into tw1 / tw2 below, or multiplied in at the store of a single-stage pass); else 1
    u32 streaming;       // data loads / stores are non-temporal (the launcher picks the NT instantiation; kept here for the record)
    // Balanced twiddle schedule of a three-pass plan (ntt_plan.hpp).  The factor in front of pass 2 splits,
    // w_N^(j2 (k0 + n0 k1)) = w_N^(j2 k0) * w_{n1 n2}^(j2 k1), and each piece goes where one of its indices is tile-uniform AND where
    // there is room:
    //   w_{n0 n1}^(j1 k0)  at pass 0's STORE (j1 is tile-uniform there: one row of a product table in LDS),
    //   w_N^(j2 k0)        in pass 1 at load: j2 is tile-uniform (the tile's h), k0 is the thread's column, so ONE factor per thread (it
    //                      commutes with the pass' transform, which runs over j1),
    //   w_{n1 n2}^(j2 k1)  at pass 2's load from a tile-uniform row (k1) of a product table.
    // Without it (two- and four-pass plans) every pass t >= 1 multiplies row r of column K by w_{N_t}^(r K): a chain gamma * delta^d per thread.
    u32 sched;
    // expansion plan, PASS_EXPAND only: coefficients [0, n_main) are the main part (n_main <= 2^main_bits = n / n_0), coefficients
    // 2^main_bits + t, t < extras, the few beyond it (a trace column's randomizers: table.py:112-136 interpolates over height + 1 points)
    u64 n_main;
    u32 main_bits, extras;
    u64 extra_scale;     // s^(2^main_bits): the coset factor of extra t is extra_scale * s^t
    u32 unit0;           // register 0 of stage 1 carries output digit 0, whose inner twiddle is w^0: skip that product unless tw1 has post_scale folded in
    const u64* tw1;      // inner twiddle table after stage 1 (t_in, or t_in_last when that is the pass' last inner twiddle)
    const u64* tw2;      // ... after stage 2 (three-stage tiles)
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

// reverse the digits p_first..p_last of h = (p_first | ... | p_last) (p_first most significant) into (p_last | ... | p_first)
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
    static_assert(MODE != PASS_FIRST || B3 == 0, "multi-pass tiles have at most two register stages");
    static constexpr int S = B1 + B2 + B3;
    static constexpr int SH1 = B2 + B3, SH2 = B3;
    static constexpr int U = (B2 == 0) ? 1 : (B3 == 0 ? 2 : 3);
    static constexpr int T = (1 << S) << LOGC;           // elements per tile
    static constexpr int W = T / 16 ? T / 16 : 1;        // threads per tile
    // LDS layout of the exchange (bank-conflict analysis: ntt_plan.hpp / DESIGN.md):
    //   PASS_COLUMN and PASS_SINGLE: [row][col], + 2^PL words every 256 (stage-1 and stage-2 lanes both run along the columns)
    //   PASS_FIRST: stage-2 lanes run along the ROW digit f1 (so that the transposed store writes 128 contiguous bytes per 16
    //               lanes) while stage-1 lanes run along the columns: a swizzled layout, lds_addr below; no padding
    static constexpr bool SWZ = (MODE == PASS_FIRST);
    static constexpr int PL = (MODE == PASS_SINGLE) ? 1 : 4;
    static constexpr int LDS_WORDS = SWZ ? T : (T + (((T - 1) >> 8) << PL) + (1 << PL));
    static constexpr int TW_WORDS = (U >= 2) ? (1 << (B1 + B2)) : 0;   // stage-1 -> stage-2 twiddles kept in LDS
};

// word index of tile element (row r, column c).  PASS_FIRST (two stages, B1 = 4): r = (k1 << B2) | o where k1 is the digit stage 1
// produced and o the digit stage 2 consumes; t1 = (o << LOGC) | c is the stage-1 thread that owns the element, and the word is
//   16 t1 + (k1 xor ((t1 + (t1 >> 4)) & 15))        (one v_xad_u32 per access: k1 is a wave-uniform scalar in stage 1).
// Stage-1 writes (lanes = consecutive t1, k1 fixed): 16 t1 steps a quarter of the 64 four-byte banks per lane and the permutation by
// t1 + (t1 >> 4) spreads the 16 lanes of a quarter over its 16 banks -- conflict-free for 64 lanes x 4 bytes (split exchange) and
// for 32 lanes x 8 bytes.  Stage-2 reads (lanes = f1 = k1 fastest, then c): 16 consecutive words per column, and the four (two)
// columns of a wave (half-wave) differ in t1 mod 4 (mod 2), i.e. sit in different quarters (halves) of the banks -- conflict-free too.
template <typename Cfg, int LOGC>
BFS_HD u32 lds_addr(u32 r, u32 c) {
    if constexpr (Cfg::SWZ) {
        const u32 k1 = r >> Cfg::SH1, o = r & ((1u << Cfg::SH1) - 1);
        const u32 t1 = (o << LOGC) | c;
        return (t1 << 4) + (k1 ^ ((t1 + (t1 >> 4)) & 15u));
    } else {
        const u32 lin = (r << LOGC) + c;
        return lin + ((lin >> 8) << Cfg::PL);
    }
}

// per-tile scalars
struct TileGeom {
    const u64* in;    // transform base (+ batch)
    u64* out;
    u64 row0;         // index of the tile's (row 0, column 0) element: h * 2^(S+logL) + c0 ; PASS_SINGLE: 0
    u64 c0;           // the tile's first column (pass t >= 1: the first of its C values of K)
    u64 h;            // the digits above the pass' own (pass 1 of a balanced plan: j_2)
    u64 kbase;        // pass t >= 1: exponent of w that column c0's chain steps by per row: c0 << tw_shift
};

template <typename Cfg, int LOGC, int MODE>
BFS_HD TileGeom tile_geom(const PassArgs& a, u32 bid_x, u32 bid_y) {
    TileGeom g;
    g.in = a.in + (u64)bid_y * a.in_batch_stride;
    g.out = a.out + (u64)bid_y * a.out_batch_stride;
    if constexpr (MODE != PASS_SINGLE) {
        g.h = bid_x >> a.lognl;
        g.c0 = (u64)(bid_x & ((1u << a.lognl) - 1)) << LOGC;
        g.row0 = (g.h << (Cfg::S + a.logL)) + g.c0;
        g.kbase = g.c0 << a.tw_shift;
    } else {
        g.h = 0; g.c0 = 0; g.row0 = 0; g.kbase = 0;
    }
    return g;
}

// the row of the load-time product table that a tile stages in LDS (null: none): last pass of a balanced plan -- row k_1, the digit of
// the tile's columns K = k_0 + n_0 k_1 above k_0 (the same for all C <= n_0 of them)
template <typename Cfg, int LOGC, int MODE>
BFS_HD const u64* tile_load_row(const PassArgs& a, u32 bid_x) {
    if constexpr (MODE != PASS_COLUMN) return nullptr;
    if (a.tb.row == nullptr) return nullptr;
    const u64 c0 = (u64)(bid_x & ((1u << a.lognl) - 1)) << LOGC;
    return a.tb.row + ((c0 >> pass_bits_of(a.pass_bits, 0)) << Cfg::S);
}
// ... and of the store-time table (pass 0 of a balanced plan): row j_1 = the leading digit of the tile's columns l = (j_1 | j_2)
template <typename Cfg, int LOGC, int MODE>
BFS_HD const u64* tile_store_row(const PassArgs& a, u32 bid_x) {
    if constexpr (MODE != PASS_FIRST) return nullptr;
    if (a.tb.srow == nullptr) return nullptr;
    const u64 c0 = (u64)(bid_x & ((1u << a.lognl) - 1)) << LOGC;
    const u64 j1 = c0 >> (a.logL - pass_bits_of(a.pass_bits, 1));
    return a.tb.srow + (j1 << Cfg::S);
}

// which tile element a thread of the LAST stage-2 (or stage-3) step holds: column c, f1 = the digit stage 1 produced, f3 = the digit
// stage 3 will consume.  Lanes run along the columns, except in PASS_FIRST where they run along f1 (TileCfg)
template <int B1, int B2, int B3, int LOGC, int MODE>
BFS_HD void stage2_pos(u32 G, u32& c, u32& f1, u32& f3) {
    if constexpr (MODE == PASS_FIRST) {
        f1 = G & ((1u << B1) - 1);
        c = (G >> B1) & ((1u << LOGC) - 1);
        f3 = G >> (B1 + LOGC);
    } else {
        c = G & ((1u << LOGC) - 1);
        const u32 rest = G >> LOGC;
        f1 = rest & ((1u << B1) - 1);
        f3 = rest >> B1;
    }
}

// store the 2^BQ registers of the last stage.  klow = the already-final lower digits of k_pass, c = column;
// srow (LDS, or null): the store-time factors of the tile's output rows
template <typename Cfg, int LOGC, int MODE, int BQ, bool NT = false>
BFS_HD void final_store(const PassArgs& a, const TileGeom& g, const u64* x, u32 klow, int kshift, u32 c, const u64* srow = nullptr) {
    constexpr int Q = 1 << BQ;
    const bool scale = (Cfg::U == 1) && a.post_scale != 1;
    u64* tp;          // per-thread base; the per-register offset below is wave-uniform
    u32 step_log;
    if constexpr (MODE == PASS_COLUMN || MODE == PASS_EXPAND) {
        tp = g.out + g.row0 + ((u64)klow << a.logL) + c;
        step_log = (u32)kshift + a.logL;
    } else if constexpr (MODE == PASS_FIRST) {
        // slot (j_{m-1} | ... | j_1 | k_0): the row of column l = c0 + c starts at reverse(l) * n_0
        tp = g.out + (digit_reverse(g.c0 + c, a.pass_bits, 1, (int)a.npass - 1) << Cfg::S) + klow;
        step_log = (u32)kshift;
    } else {
        tp = g.out + klow;
        step_log = (u32)kshift;
    }
    BFS_UNROLL
    for (int m = 0; m < Q; ++m) {
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

// (o, c) of stage-1 thread `tid`: o = the row bits below the stage's digit, c = column
template <int B1, int B2, int B3, int LOGC, int MODE>
BFS_HD void stage1_oc(u32 tid, int sub, u32& o, u32& c) {
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    const u32 G = (u32)sub * Cfg::W + tid;
    c = G & ((1u << LOGC) - 1);
    o = G >> LOGC;
}

template <int B1, int B2, int B3, int LOGC, int MODE>
BFS_HD Stage1Pos<B1, B2, B3, LOGC, MODE> stage1_pos(const PassArgs& a, const TileGeom& g, u32 tid, int sub) {
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    Stage1Pos<B1, B2, B3, LOGC, MODE> p;
    stage1_oc<B1, B2, B3, LOGC, MODE>(tid, sub, p.o, p.c);
    if constexpr (MODE != PASS_SINGLE) {
        p.idx0 = g.row0 + ((u64)p.o << a.logL) + p.c;
        p.step_log = Cfg::SH1 + a.logL;
    } else {
        p.idx0 = p.o;
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
    if constexpr (MODE == PASS_EXPAND) {
        // coefficient (row << tw_shift) + h for row = (d << SH1) | o: one address for all C columns of the tile
        const u64 i0 = ((u64)p.o << a.tw_shift) + g.h;
        const u32 sl = Cfg::SH1 + a.tw_shift;
        const u64* ip = a.in + (u64)bid_y * a.in_batch_stride + i0;
        BFS_UNROLL
        for (int d = 0; d < Q; ++d) x[d] = (i0 + ((u64)d << sl) < a.n_main) ? ntt_ld<NT>(ip + ((u64)d << sl)) : 0;
        return;
    }
    const u64* tp = g.in + p.idx0;
    if (a.partial) {
        BFS_UNROLL
        for (int d = 0; d < Q; ++d) x[d] = (p.idx0 + ((u64)d << p.step_log) < a.n_in) ? ntt_ld<NT>(tp + ((u64)d << p.step_log)) : 0;
    } else {
        BFS_UNROLL
        for (int d = 0; d < Q; ++d) x[d] = ntt_ld<NT>(tp + ((u64)d << p.step_log));
    }
}

// The stage 1 -> 2 exchange: register m of stage-1 thread `tid` goes to tile word stage1_out_index, and register d of sub-group s
// of stage-2 thread `tid` comes from stage2_in_index (word indices of the tile layout, lds_addr).
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
    u32 c, f1, f3;
    stage2_pos<B1, B2, B3, LOGC, MODE>((u32)s * Cfg::W + tid, c, f1, f3);
    return lds_addr<Cfg, LOGC>((f1 << Cfg::SH1) | ((u32)d << Cfg::SH2) | f3, c);
}

// stage 1 without the exchange: load-time twiddle, first radix and (U >= 2) the inner twiddle; x[m] is left holding the value
// that stage1_out_index(m) receives.  U == 1: the values are final and are stored.
// tw: dense inner-twiddle table for the stage 1 -> 2 exchange (2^(B1+B2) entries of a.tw1, in LDS on the GPU)
template <int B1, int B2, int B3, int LOGC, int MODE, bool NT = false>
BFS_HD void ntt_stage1_values(const PassArgs& a, const u64* tw, const u64* rowtw, u32 tid, u32 bid_x, u32 bid_y, int sub, u64* x,
                              const u64* srow = nullptr) {
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    constexpr int Q = 1 << B1;
    const TileGeom g = tile_geom<Cfg, LOGC, MODE>(a, bid_x, bid_y);
    const u64 nmask = (1ull << a.log_n) - 1;
    const Stage1Pos<B1, B2, B3, LOGC, MODE> p = stage1_pos<B1, B2, B3, LOGC, MODE>(a, g, tid, sub);
    const u32 o = p.o, c = p.c;
    // zero-padded input: element d of a thread is input index idx0 + (d << step_log), which lies beyond n_in for every thread once
    // d << step_log does -- registers d >= dmax hold zeros in the whole launch (a wave-uniform, in fact launch-uniform, number)
    u32 dmax = Q;
    if constexpr (MODE == PASS_EXPAND) {
        const u32 sl = Cfg::SH1 + a.tw_shift;
        const u64 live = (a.n_main + ((1ull << sl) - 1)) >> sl;
        if (live < (u64)Q) dmax = (u32)live;
    } else if (a.partial) {
        const u64 live = (a.n_in + ((1ull << p.step_log) - 1)) >> p.step_log;
        if (live < (u64)Q) dmax = (u32)live;
    }
    if constexpr (MODE == PASS_EXPAND) {
        // The extras.  Coefficient 2^M + t (M = main_bits, t < extras) has leading digit 1 and the rest t: the copy that stands for the
        // skipped first pass leaves  x_t s^t + x_(2^M + t) s^(2^M + t) w_{n_0}^(k_0)  in the slot of rest t and column k_0, i.e. the main
        // coefficient plus  s^(2^M) w^(k_0 2^M)  times the extra one, before this pass' own factor s^t w_{N_1}^(row K) multiplies both.  The
        // planner keeps extras below the stride of register 1, so only register 0 of the threads whose own coefficient index
        // (o << tw_shift) + h is below `extras` is touched: a handful of threads per transform.
        const u64 i0 = ((u64)o << a.tw_shift) + g.h;
        if (a.extras != 0 && i0 < (u64)a.extras) {
            const u64 e = g.in[(1ull << a.main_bits) + i0];
            u64 f = tw_pow(a.tb.w_lo, a.tb.w_hi, a.tb.lo_bits, ((g.c0 + c) << a.main_bits) & nmask);
            if (a.has_coset) f = gl_mul(f, a.extra_scale);
            x[0] = gl_add(x[0], gl_mul(e, f));
        }
    }
    if (rowtw != nullptr) {
        // last pass of a balanced plan: the factors w_{n1 n2}^(j2 k1) of the tile's rows j2 are one row (k1) of a table
        BFS_UNROLL
        for (int d = 0; d < Q; ++d) x[d] = gl_mul(x[d], rowtw[((u32)d << Cfg::SH1) | o]);
    } else if (a.sched && a.pass_index == 1) {
        // balanced plan, middle pass: w_N^(j2 k0) -- j2 belongs to the tile, k0 is this thread's column: one factor for all 16 rows
        const u64 f = tw_pow(a.tb.w_lo, a.tb.w_hi, a.tb.lo_bits, (g.h * (g.c0 + c)) & nmask);
        BFS_UNROLL
        for (int d = 0; d < Q; ++d) x[d] = gl_mul(x[d], f);
    } else if (a.pass_index > 0 || a.has_coset) {
        // factor of row r = (d << SH1) | o is beta^r = gamma * delta^d: a geometric chain per thread
        u64 gam, del;
        if (a.pass_index > 0) {
            const u64 ks = g.kbase + ((u64)c << a.tw_shift);                 // w^ks = w_{N_t}^K for this thread's column K = c0 + c
            gam = tw_pow(a.tb.w_lo, a.tb.w_hi, a.tb.lo_bits, ((u64)o * ks) & nmask);
            del = tw_pow(a.tb.w_lo, a.tb.w_hi, a.tb.lo_bits, (ks << Cfg::SH1) & nmask);
            if constexpr (MODE == PASS_EXPAND) {
                if (a.has_coset) {       // ... times s^j for the coefficient's own index j = (row << tw_shift) + h: a geometric chain in d as well
                    gam = gl_mul(gam, tw_pow(a.tb.s_lo, a.tb.s_hi, a.tb.lo_bits, ((u64)o << a.tw_shift) + g.h));
                    del = gl_mul(del, a.coset_delta);
                }
            }
        } else {
            gam = tw_pow(a.tb.s_lo, a.tb.s_hi, a.tb.lo_bits, p.idx0);
            del = a.coset_delta;
        }
        u64 f = gam;
        BFS_UNROLL
        for (int d = 0; d < Q; ++d) {
            if ((u32)d < dmax) {                      // (zeros need neither their factor nor the chain's next step)
                x[d] = gl_mul(x[d], f);
                if (d + 1 < Q) f = gl_mul_lazy(f, del);   // only ever multiplied again: no canonical form needed
            }
        }
    }
    // U >= 2: every output meets an inner twiddle below (the one with a unit twiddle is reduced there)
    if (Q == 16 && dmax <= 8) {                    // (wave-uniform) zero-padded input: the network for the registers that can be non-zero
        if constexpr (Q == 16) {
            if (dmax <= 1) dif_sparse<16, 1, (Cfg::U >= 2)>(x);
            else if (dmax <= 2) dif_sparse<16, 2, (Cfg::U >= 2)>(x);
            else if (dmax <= 4) dif_sparse<16, 4, (Cfg::U >= 2)>(x);
            else dif_sparse<16, 8, (Cfg::U >= 2)>(x);
        }
    } else {
        dif<Q, (Cfg::U >= 2)>(x);
    }
    if constexpr (Cfg::U == 1) {
        final_store<Cfg, LOGC, MODE, B1, NT>(a, g, x, 0, 0, c, srow);
    } else {
        const u32 i2 = o >> B3;
        BFS_UNROLL
        for (int m = 0; m < Q; ++m) {
            const u32 k1 = perm_digit<B1>(m, a.uinv);                      // wave-uniform
            const u32 e = mul24(i2, k1) & ((1u << (B1 + B2)) - 1);         // exponent of w_M, M = 2^(B1+B2)
            // register 0 holds output digit 0: its twiddle is w^0, which is 1 unless a post-scale (n^-1 of intt) is folded into the
            // table -- a wave-uniform question, so a forward transform skips that product
            const bool unit = (m == 0) && a.unit0;
            if (!unit) x[m] = gl_mul(x[m], tw[e]);
            else x[m] = gl_canon(x[m]);
        }
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
        u32 c, f1, f3;
        stage2_pos<B1, B2, B3, LOGC, MODE>((u32)s * Cfg::W + tid, c, f1, f3);
        // unreduced outputs where all of them are multiplied next: by the inner twiddles of a third stage, or -- first pass of a multi-pass
        // plan -- by the store-time row here or by the next pass, which multiplies everything it loads
        dif<Q, (Cfg::U == 3 || MODE == PASS_FIRST)>(x);
        if constexpr (Cfg::U == 2) {
            final_store<Cfg, LOGC, MODE, B2, NT>(a, g, x, f1, B1, c, srow);
        } else {
            BFS_UNROLL
            for (int m = 0; m < Q; ++m) {
                const u32 k2 = perm_digit<B2>(m, a.uinv);
                const u32 e = mul24(f3, f1 + (k2 << B1)) & ((1u << Cfg::S) - 1);
                const u64 v = gl_mul(x[m], a.tw2[(u64)e << (a.tb.t_in_log - Cfg::S)]);
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
