// ntt_plan.hpp -- host-side planning for bfs_gl_ntt(): pass split, tile shapes, twiddle tables.
// Pure C++ (no HIP calls) so that the planner is also exercised by the host emulation test.
// Validation mirrors the reference's asserts: /root/reference/code/ntt.py:5-6 (power of two),
// :13-14 (w^n == 1), :15-16 (w^(n/2) != 1).
#pragma once
#include <cstdlib>
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
};

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

// tests: -1 = follow the environment (BFS_NTT_SCHEDULE), 0 / 1 = force the load-time / the balanced schedule
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
    // enumerate splits S_0..S_{m-1} in [4,8]; keep tiles at 4096 elements; pick the most balanced feasible one
    u32 best[4] = {0, 0, 0, 0};
    u32 best_score = ~0u;
    u32 s[4];
    u32 total = 1;
    for (u32 i = 0; i < m; ++i) total *= 5;
    for (u32 code = 0; code < total; ++code) {
        u32 c = code, sum = 0, mx = 0, mn = 99;
        for (u32 i = 0; i < m; ++i) { s[i] = 4 + c % 5; c /= 5; sum += s[i]; mx = s[i] > mx ? s[i] : mx; mn = s[i] < mn ? s[i] : mn; }
        if (sum != log_n) continue;
        bool ok = true;
        u32 done = 0;
        for (u32 t = 0; t + 1 < m && ok; ++t) {
            done += s[t];
            if (NTT_TILE_LOG - s[t] > log_n - done) ok = false;  // C_t <= L_t
        }
        if (NTT_TILE_LOG - s[m - 1] > s[0]) ok = false;          // final pass: C <= n_1
        if (!ok) continue;
        u32 score = (mx - mn) * 16 + (8 - s[m - 1]);             // balanced first, then a long last digit
        if (score < best_score) { best_score = score; for (u32 i = 0; i < m; ++i) best[i] = s[i]; }
    }
    if (best_score == ~0u) return false;
    p.npass = m;
    for (u32 i = 0; i < m; ++i) { p.pass_bits[i] = best[i]; p.logC[i] = NTT_TILE_LOG - best[i]; }
    // Balanced schedule (three passes): needs the first pass' tiles to sit inside one value of the next digit (C_1 <= n_3 ... the
    // tile's columns l = j2 n3 + j3 then share j2), two-stage tiles everywhere, and product tables of at most 2^16 entries.
    // BFS_NTT_SCHEDULE=0 keeps the load-time schedule (A/B, tools/ab_ntt.sh).
    static const bool env_balanced = [] { const char* e = getenv("BFS_NTT_SCHEDULE"); return !(e && e[0] == '0'); }();
    const bool balanced = ntt_schedule_override() < 0 ? env_balanced : ntt_schedule_override() != 0;
    p.sched = (balanced && m == 3 && p.logC[0] <= best[2] && best[0] >= 5 && best[1] >= 5 && best[2] >= 5 && best[0] + best[1] <= 16 &&
               best[1] + best[2] <= 16) ? 1 : 0;
    return true;
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

// The product tables pass t of a plan reads: `load` (NttTables::row: a tile multiplies its 2^S input rows by one row of it) and `store`
// (NttTables::srow: ... its 2^S output rows).  omega = 0: no such table.
struct NttRowSpec {
    u64 omega = 0;
    u32 a_bits = 0, b_bits = 0;
};
inline void ntt_row_specs(const NttPlan& p, u32 t, u64 root, NttRowSpec& load, NttRowSpec& store) {
    load = NttRowSpec(); store = NttRowSpec();
    if (p.npass < 2) return;
    const bool last = t + 1 == p.npass;
    if (p.sched) {
        const u32 s0 = p.pass_bits[0], s1 = p.pass_bits[1], s2 = p.pass_bits[2];
        if (t == 0) store = NttRowSpec{gl_pow(root, 1ull << s2), s1, s0};          // rows j2, entries k1:  w_{n1 n2}^(j2 k1)
        if (t == 2) load = NttRowSpec{gl_pow(root, 1ull << s0), s1, s2};           // rows k2, entries j3:  w_{n2 n3}^(k2 j3)
        return;
    }
    u32 done = 0;
    for (u32 v = 0; v <= t; ++v) done += p.pass_bits[v];
    if (t == 0 || last || done > 16) return;
    load = NttRowSpec{gl_pow(root, 1ull << (p.log_n - done)), done - p.pass_bits[t], p.pass_bits[t]};     // rows K, entries r:  w_{N_t}^(K r)
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

// LDS layout per pass (TileCfg in ntt_core.hpp), chosen so that both the stage-1 writes and the stage-2 reads of the tile
// are free of bank conflicts (ds_read_b64: 64 banks x 4 B per 32-lane group; ds_write_b64: 32 banks per 16-lane group):
//   column pass  [row][col], +16 words every 256: stage-2 lanes (c = tid & 15, f1 = tid >> 4) of one 32-lane group then fall on
//                disjoint bank halves (first version: +2 words -> 2-way conflicts on every read)
//   final pass of a multi-pass plan: lanes run along the ROW index when loading (contiguous HBM rows), so the tile is kept
//                [col][row] with +1 word per column: writes are consecutive, stage-2 reads (lanes along c) step 2 banks per lane
//                (first version: [row][col] -> 16-way conflicts on every write, 73 % of LDS cycles, profiles/r01)
//   single-pass plans (one column): +2 words every 256 rows

// fill the per-pass kernel arguments (pointers are whatever address space the caller runs in)
inline PassArgs ntt_pass_args(const NttPlan& p, u32 t, const u64* in, u64* out, u64 in_stride, u64 out_stride,
                              u64 n_in, const NttTables& tb, bool has_coset, u64 shift, u64 post_scale) {
    PassArgs a{};
    const bool final_pass = (t + 1 == p.npass);
    const u32 S = p.pass_bits[t];
    a.in = in; a.out = out;
    a.in_batch_stride = in_stride; a.out_batch_stride = out_stride;
    a.n_in = n_in;
    a.partial = (t == 0 && n_in < (1ull << p.log_n)) ? 1 : 0;
    a.log_n = p.log_n;
    a.pass_index = t;
    a.npass = p.npass;
    a.pass_bits = p.pass_bits[0] | (p.pass_bits[1] << 8) | (p.pass_bits[2] << 16) | (p.pass_bits[3] << 24);
    u32 done = 0;
    for (u32 v = 0; v <= t; ++v) done += p.pass_bits[v];
    if (!final_pass) {
        a.logL = p.log_n - done;
        a.lognl = a.logL - p.logC[t];
        a.tw_shift = p.log_n - done;
    } else if (p.npass > 1) {
        a.n1_bits = p.pass_bits[0];
        for (u32 v = 1; v + 1 < p.npass; ++v) a.mid_bits += p.pass_bits[v];
        a.logch = a.n1_bits - p.logC[t];
    }
    a.uinv = p.uinv;
    a.sched = p.sched;
    a.has_coset = (t == 0 && has_coset) ? 1 : 0;
    a.post_scale = final_pass ? post_scale : 1;
    a.tb = tb;
    if (a.has_coset) {
        u32 b1 = S < 4 ? S : 4;
        u32 sh1 = S - b1;
        u64 stride = !final_pass ? ((1ull << (p.log_n - S)) << sh1) : (1ull << sh1);
        a.coset_delta = gl_pow(shift, stride);
    }
    return a;
}

}  // namespace bfs
