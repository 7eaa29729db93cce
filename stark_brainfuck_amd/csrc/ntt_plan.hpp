// ntt_plan.hpp -- host-side planning for bfs_gl_ntt(): pass split, tile shapes, twiddle tables.
// Pure C++ (no HIP calls) so that the planner is also exercised by the host emulation test.
// Validation mirrors the reference's asserts: /root/reference/code/ntt.py:5-6 (power of two),
// :13-14 (w^n == 1), :15-16 (w^(n/2) != 1).
#pragma once
#include <vector>

#include "../../include/bfstark.h"
#include "ntt_core.hpp"

namespace bfs {


constexpr u32 NTT_TILE_LOG = 12;      // 4096 elements (32 KiB) per tile in the multi-pass regime
constexpr u32 NTT_MAX_PASS_BITS = 8;  // digits of a multi-pass plan are <= 2^8 so tiles keep >= 16 columns (128 B segments)
constexpr u32 NTT_SMALL_LOG = 3;
      // n <= 8 goes through the direct small kernel

struct NttPlan {
    u32 log_n = 0;
    u32 npass = 0;
    u32 pass_bits[4] = {0, 0, 0, 0};
    u32 logC[4] = {0, 0, 0, 0};
    u32 uinv = 1;
    u32 lo_bits = 0;
    u32 t_in_log = 0;
    u32 sched = 0;                    // 1: the balanced twiddle schedule of a three-pass plan (PassArgs::sched)
    // expansion plan (ntt_make_expand_plan): pass 0 is VIRTUAL -- pass_bits[0] = e leading bits of the input index that are zero for
    // all but `extras` coefficients -- and passes 1.. are real; main_bits = log n - e
    u32 expand = 0, main_bits = 0, extras = 0;
};
constexpr u32 NTT_EXPAND_MAX_EXTRAS = 16;

inline int ntt_check_root(u64 root, u32 log_n) {
    if (log_n == 0) return BFS_OK;  // ntt.py:8-9: length <= 1 returns its input unchecked
    if (gl_pow(root, 1ull << log_n) != 1) return BFS_ERR_NOT_ROOT;
    if (gl_pow(root, 1ull << (log_n - 1)) == 1) return BFS_ERR_NOT_PRIMITIVE;
    return BFS_OK;
}

// u with root^(n/16) = 2^(12u); returns u^-1 mod 16
inline u32 ntt_uinv(u64 root, u32 log_n) {
    if (log_n < 4) return 1;
    u64 w16 = gl_pow(root, 1ull << (log_n - 4));
    u64 c = cx_pow2(12), acc = 1;
    u32 u = 0;
    for (u32 i = 1; i < 16; ++i) {
        acc = gl_mul(acc, c);
        if (acc == w16) { u = i; break; }
    }
    for (u32 v = 1; v < 16; v += 2)
        if (((u * v) & 15) == 1) return v;
    return 1;
}

// tests: -1 = the planner's choice, 0 / 1 = force the chain / the balanced schedule where a plan allows it (tests/test_emulation.py
// runs three-pass plans through both twiddle paths; the product never sets it)
inline int& ntt_schedule_override() {
    static int v = -1;
    return v;
}

inline bool ntt_make_plan(u32 log_n, u64 root, NttPlan& p) {
    p = NttPlan();
    p.log_n = log_n;
    p.uinv = ntt_uinv(root, log_n);
    p.lo_bits = (log_n + 1) / 2;
    p.t_in_log = log_n < 12 ? log_n : 12;
    if (log_n <= NTT_SMALL_LOG) { p.npass = 0; return true; }
    if (log_n <= NTT_TILE_LOG) {
        p.npass = 1; p.pass_bits[0] = log_n; p.logC[0] = 0;
        return true;
    }
    u32 m = (log_n + NTT_MAX_PASS_BITS - 1) / NTT_MAX_PASS_BITS;
    if (m > 4) return false;
    // enumerate splits S_0..S_{m-1} in [5,8] (two register stages per tile; every log n in 13..32 has such a split); tiles are 4096
    // elements (C_t = 2^(12 - S_t) columns); pick the most balanced feasible one
    u32 best[4] = {0, 0, 0, 0};
    u32 best_score = ~0u;
    u32 s[4];
    u32 total = 1;
    for (u32 i = 0; i < m; ++i) total *= 4;
    for (u32 code = 0; code < total; ++code) {
        u32 c = code, sum = 0, mx = 0, mn = 99;
        for (u32 i = 0; i < m; ++i) { s[i] = 5 + c % 4; c /= 4; sum += s[i]; mx = s[i] > mx ? s[i] : mx; mn = s[i] < mn ? s[i] : mn; }
        if (sum != log_n) continue;
        // pass 0: its C_0 columns are values of l = (j_1|..|j_{m-1}) < n / n_0 -- always enough for log n > 12.
        // pass t >= 1: its C_t columns are values of K < n_0 .. n_{t-1}
        bool ok = true;
        u32 done = s[0];
        for (u32 t = 1; t < m && ok; ++t) {
            if (NTT_TILE_LOG - s[t] > done) ok = false;
            done += s[t];
        }
        if (!ok) continue;
        // balanced first; then (three passes) a split the balanced twiddle schedule accepts; then a long first digit (long rows in
        // the transposed store)
        const bool sched_ok = m == 3 && s[0] + s[2] >= NTT_TILE_LOG && s[0] >= 5 && s[1] >= 5 && s[2] >= 5;
        u32 score = (mx - mn) * 32 + (m == 3 && !sched_ok ? 16 : 0) + (8 - s[0]);
        if (score < best_score) { best_score = score; for (u32 i = 0; i < m; ++i) best[i] = s[i]; }
    }
    if (best_score == ~0u) return false;
    p.npass = m;
    for (u32 i = 0; i < m; ++i) { p.pass_bits[i] = best[i]; p.logC[i] = NTT_TILE_LOG - best[i]; }
    // Balanced schedule (three passes): pass 0's tile must sit inside one value of j_1 (C_0 <= n_2: its columns l = j_1 n_2 + j_2
    // then share j_1) and the last pass' tile inside one value of k_1 (C_2 <= n_0) -- both say S_0 + S_2 >= 12 --, two-stage tiles
    // everywhere, and product tables of at most 2^16 entries.
    const bool can = m == 3 && best[0] + best[2] >= NTT_TILE_LOG && best[0] >= 5 && best[1] >= 5 && best[2] >= 5 &&
                     best[0] + best[1] <= 16 && best[1] + best[2] <= 16;
    p.sched = (can && ntt_schedule_override() != 0) ? 1 : 0;
    return true;
}

// Expansion plan for a zero-padded transform of n_in coefficients on n = 2^log_n points (fast_coset_evaluate, ntt.py:164-168; the shape
// Table.lde makes, table.py:138-149: height + 1 coefficients on ~64 height points).  Write the input index as (j_0 | j_1 | ..) with e
// leading bits j_0: when n_in <= 2^(log n - e) (+ a few extras) every coefficient has j_0 = 0, the first pass' 2^e-point transform of
// (x, 0, .., 0) is 2^e copies of x, and the plan can start at the second digit: passes over the remaining M = log n - e bits, the first of
// them reading the coefficients instead of slots (PASS_EXPAND).  Taken when that saves a pass: M <= 8 -> one real pass, M <= 16 -> two.
// Extras: up to NTT_EXPAND_MAX_EXTRAS coefficients 2^M + t (a trace interpolant has height + num_randomizers coefficients, one or two
// past a power of two) enter as rank-one terms of the first real pass instead of costing a bit of e.
inline bool ntt_try_expand_plan(u32 log_n, u32 M, u64 r, const NttPlan& plain, NttPlan& p) {
    if (M < 5 || M == 9 || M > 16 || log_n < M + 4) return false;   // digits are 5..8 bits (9 = 5 + 4 does not split), at most two real passes; e >= 4 keeps 16 adjacent columns per tile
    const u32 e = log_n - M;
    const u32 real = M <= 8 ? 1 : 2;
    if (real >= plain.npass) return false;                    // taken only when it saves a pass
    p = plain;
    p.sched = 0;
    p.expand = 1;
    p.main_bits = M;
    p.extras = (u32)r;
    p.npass = 1 + real;
    for (int i = 0; i < 4; ++i) { p.pass_bits[i] = 0; p.logC[i] = 0; }
    p.pass_bits[0] = e;
    // the real digits: one of M bits, or the most balanced pair S_1 + S_2 = M (5..8 bits each) whose tiles find their columns among the
    // finished digits (C_1 = 2^(12 - S_1) <= 2^e values of k_0, C_2 <= 2^(e + S_1)) and whose first pass reaches the extras: an extra's
    // rest index t must belong to register 0 of a thread of the first real pass, t < 2^(SH1 + tw_shift) with SH1 = S_1 - 4
    u32 best1 = 0, best_gap = 99;
    for (u32 s1 = 5; s1 <= 8; ++s1) {
        const u32 s2 = real == 1 ? 0 : M - s1;
        if (real == 1 ? s1 != M : (s2 < 5 || s2 > 8)) continue;
        if (NTT_TILE_LOG - s1 > e) continue;
        if (real == 2 && NTT_TILE_LOG - s2 > e + s1) continue;
        if (r > (1ull << ((s1 - 4) + s2))) continue;
        const u32 gap = real == 1 ? 0 : (s1 > s2 ? s1 - s2 : s2 - s1);
        if (gap < best_gap || (gap == best_gap && s1 > best1)) { best_gap = gap; best1 = s1; }
    }
    if (best1 == 0) return false;
    p.pass_bits[1] = best1;
    p.logC[1] = NTT_TILE_LOG - best1;
    if (real == 2) {
        p.pass_bits[2] = M - best1;
        p.logC[2] = NTT_TILE_LOG - p.pass_bits[2];
    }
    return true;
}
inline bool ntt_make_expand_plan(u32 log_n, u64 n_in, u64 root, const NttPlan& plain, NttPlan& p) {
    (void)root;
    if (plain.npass < 2 || n_in == 0 || n_in >= (1ull << log_n)) return false;
    u32 m0 = 0;
    while ((2ull << m0) <= n_in) ++m0;                       // 2^m0 <= n_in < 2^(m0 + 1)
    const u64 r = n_in - (1ull << m0);
    // the main part 2^m0 with the r coefficients beyond it as extras, else the next sizes up with zero padding inside the main part
    if (r <= NTT_EXPAND_MAX_EXTRAS && ntt_try_expand_plan(log_n, m0, r, plain, p)) return true;
    for (u32 M = (r == 0 ? m0 : m0 + 1); M <= 16; ++M)
        if (M >= 5 && ntt_try_expand_plan(log_n, M, 0, plain, p)) return true;
    return false;
}

// product table: out[(a << b_bits) + b] = omega^(a b)
inline void ntt_product_table(u64 omega, u32 a_bits, u32 b_bits, std::vector<u64>& out) {
    out.resize((size_t)1 << (a_bits + b_bits));
    u64 wa = 1;                                  // omega^a
    for (u64 a = 0; a < (1ull << a_bits); ++a) {
        u64 v = 1;
        for (u64 b = 0; b < (1ull << b_bits); ++b) { out[(a << b_bits) + b] = v; v = gl_mul(v, wa); }
        wa = gl_mul(wa, omega);
    }
}

// The product tables pass t of a balanced plan reads: `store` (NttTables::srow, pass 0: a tile multiplies its 2^S_0 OUTPUT rows k_0 by
// row j_1 of it) and `load` (NttTables::row, pass 2: ... its 2^S_2 INPUT rows j_2 by row k_1).  omega = 0: no such table.
struct NttRowSpec {
    u64 omega = 0;
    u32 a_bits = 0, b_bits = 0;
};
inline void ntt_row_specs(const NttPlan& p, u32 t, u64 root, NttRowSpec& load, NttRowSpec& store) {
    load = NttRowSpec(); store = NttRowSpec();
    if (!p.sched) return;
    const u32 s0 = p.pass_bits[0], s1 = p.pass_bits[1], s2 = p.pass_bits[2];
    if (t == 0) store = NttRowSpec{gl_pow(root, 1ull << s2), s1, s0};          // rows j1, entries k0:  w_{n0 n1}^(j1 k0)
    if (t == 2) load = NttRowSpec{gl_pow(root, 1ull << s0), s1, s2};           // rows k1, entries j2:  w_{n1 n2}^(k1 j2)
}

// powers table helper: out[i] = base^i * scale
inline void fill_powers(std::vector<u64>& out, size_t count, u64 base, u64 scale) {
    out.resize(count);
    u64 v = scale;
    for (size_t i = 0; i < count; ++i) { out[i] = v; v = gl_mul(v, base); }
}

struct NttHostTables {
    std::vector<u64> w_lo, w_hi, t_in, t_in_last;
};

inline void ntt_build_tables(const NttPlan& p, u64 root, u64 post_scale, NttHostTables& t) {
    u32 hi_bits = p.log_n - p.lo_bits;
    fill_powers(t.w_lo, 1ull << p.lo_bits, root, 1);
    fill_powers(t.w_hi, 1ull << hi_bits, gl_pow(root, 1ull << p.lo_bits), 1);
    u64 omega = gl_pow(root, 1ull << (p.log_n - p.t_in_log));
    fill_powers(t.t_in, 1ull << p.t_in_log, omega, 1);
    fill_powers(t.t_in_last, 1ull << p.t_in_log, omega, post_scale);
}

struct CosetHostTables {
    std::vector<u64> s_lo, s_hi;
};

inline void ntt_build_coset_tables(const NttPlan& p, u64 shift, CosetHostTables& t) {
    u32 hi_bits = p.log_n - p.lo_bits;
    fill_powers(t.s_lo, 1ull << p.lo_bits, shift, 1);
    fill_powers(t.s_hi, 1ull << hi_bits, gl_pow(shift, 1ull << p.lo_bits), 1);
}

// LDS layout per pass (TileCfg / lds_addr in ntt_core.hpp), chosen so that both the stage-1 writes and the stage-2 reads of the tile
// are free of bank conflicts (ds_read_b64: 64 banks x 4 B per 32-lane group; ds_write_b64: 32 banks per 16-lane group):
//   column pass  [row][col], +16 words every 256: stage-2 lanes (c = tid & 15, f1 = tid >> 4) of one 32-lane group then fall on
//                disjoint bank halves (first version: +2 words -> 2-way conflicts on every read)
//   first pass of a multi-pass plan: stage-2 lanes run along the output ROW digit (contiguous transposed stores) while stage-1 lanes
//                run along the columns (coalesced loads): the swizzled layout of lds_addr
//   single-pass plans (one column): +2 words every 256 rows

// fill the per-pass kernel arguments (pointers are whatever address space the caller runs in)
inline PassArgs ntt_pass_args(const NttPlan& p, u32 t, const u64* in, u64* out, u64 in_stride, u64 out_stride,
                              u64 n_in, const NttTables& tb, bool has_coset, u64 shift, u64 post_scale) {
    PassArgs a{};
    const bool last = (t + 1 == p.npass);
    const u32 S = p.pass_bits[t];
    a.in = in; a.out = out;
    a.in_batch_stride = in_stride; a.out_batch_stride = out_stride;
    a.n_in = n_in;
    a.partial = (t == 0 && n_in < (1ull << p.log_n)) ? 1 : 0;
    a.log_n = p.log_n;
    a.pass_index = t;
    a.npass = p.npass;
    a.pass_bits = p.pass_bits[0] | (p.pass_bits[1] << 8) | (p.pass_bits[2] << 16) | (p.pass_bits[3] << 24);
    if (p.npass > 1) {
        if (t == 0) {
            a.logL = p.log_n - S;
        } else {
            for (u32 v = 0; v < t; ++v) a.logL += p.pass_bits[v];
            a.tw_shift = p.log_n - a.logL - S;
        }
        a.lognl = a.logL - p.logC[t];
    }
    a.uinv = p.uinv;
    a.sched = p.sched;
    if (p.expand && t == 1) {
        // the first real pass of an expansion plan reads the coefficients: n_in of them, of which the first n_main are the main part
        a.main_bits = p.main_bits;
        a.extras = p.extras;
        a.n_main = n_in < (1ull << p.main_bits) ? n_in : (1ull << p.main_bits);
        a.has_coset = has_coset ? 1 : 0;
        if (has_coset) {
            const u32 sh1 = S - (S < 4 ? S : 4);
            a.coset_delta = gl_pow(shift, 1ull << (sh1 + a.tw_shift));
            a.extra_scale = gl_pow(shift, 1ull << p.main_bits);
        }
        a.post_scale = last ? post_scale : 1;
        a.tb = tb;
        a.tw1 = last ? tb.t_in_last : tb.t_in;
        a.tw2 = tb.t_in;
        a.unit0 = (a.tw1 == tb.t_in || post_scale == 1) ? 1 : 0;
        return a;
    }
    a.has_coset = (t == 0 && has_coset) ? 1 : 0;
    a.post_scale = last ? post_scale : 1;
    a.tb = tb;
    // the inner twiddle tables: n^-1 of intt rides on the LAST inner twiddle of the last pass (a single-stage last pass multiplies at its store)
    const u32 stages = S <= 4 ? 1 : (S <= 8 ? 2 : 3);
    a.tw1 = (last && stages == 2) ? tb.t_in_last : tb.t_in;
    a.tw2 = (last && stages == 3) ? tb.t_in_last : tb.t_in;
    a.unit0 = (a.tw1 == tb.t_in || post_scale == 1) ? 1 : 0;
    if (a.has_coset) {
        u32 b1 = S < 4 ? S : 4;
        u32 sh1 = S - b1;
        u64 stride = p.npass > 1 ? ((1ull << (p.log_n - S)) << sh1) : (1ull << sh1);
        a.coset_delta = gl_pow(shift, stride);
    }
    return a;
}

}  // namespace bfs
