// Host emulation of the NTT tile kernels (TEST INFRASTRUCTURE, never part of the product path).
// Compiles stark_brainfuck_amd/csrc/ntt_core.hpp for the host and runs every (block, thread) of every pass
// sequentially, stage by stage -- __syncthreads() becomes "finish the stage for all threads of the block".
// Lets tests/test_emulation.py check the planner and all index/twiddle arithmetic against the oracle on CPU.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../stark_brainfuck_amd/csrc/ntt_plan.hpp"

using namespace bfs;

template <int B1, int B2, int B3, int LOGC, int MODE>
static void run_pass(const PassArgs& a, u32 grid_x, u32 batch) {
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    std::vector<u64> smem(Cfg::LDS_WORDS + 2);
    // dense stage-1 -> stage-2 twiddle table, as the kernel builds it in LDS
    std::vector<u64> tw(1u << (B1 + B2));
    const u64* tab = a.tw1;
    if (B2 > 0) for (u32 i = 0; i < tw.size(); ++i) tw[i] = tab[(u64)i << (a.tb.t_in_log - (B1 + B2))];
    for (u32 by = 0; by < batch; ++by)
        for (u32 bx = 0; bx < grid_x; ++bx) {
            // the tile's row of the load-time / store-time product table, copied as the kernel copies it to LDS
            std::vector<u64> row, srow_copy;
            const u64* lrow = tile_load_row<Cfg, LOGC, MODE>(a, bx);
            const u64* srow_g = tile_store_row<Cfg, LOGC, MODE>(a, bx);
            if (lrow) row.assign(lrow, lrow + (1u << Cfg::S));
            if (srow_g) srow_copy.assign(srow_g, srow_g + (1u << Cfg::S));
            const u64* rowtw = lrow ? row.data() : nullptr;
            const u64* srow = srow_g && !lrow ? srow_copy.data() : nullptr;
            for (u32 t = 0; t < (u32)Cfg::W; ++t) ntt_stage1<B1, B2, B3, LOGC, MODE>(a, smem.data(), tw.data(), rowtw, t, bx, by, srow);
            if (B2 > 0) for (u32 t = 0; t < (u32)Cfg::W; ++t) ntt_stage2<B1, B2, B3, LOGC, MODE>(a, smem.data(), t, bx, by, srow);
            if (B3 > 0) for (u32 t = 0; t < (u32)Cfg::W; ++t) ntt_stage3<B1, B2, B3, LOGC, MODE>(a, smem.data(), t, bx, by, srow);
        }
}

template <int MODE>
static void dispatch_multi(const PassArgs& a, u32 S, u32 grid_x, u32 batch) {
    switch (S) {
        case 5: run_pass<4, 1, 0, 7, MODE>(a, grid_x, batch); break;
        case 6: run_pass<4, 2, 0, 6, MODE>(a, grid_x, batch); break;
        case 7: run_pass<4, 3, 0, 5, MODE>(a, grid_x, batch); break;
        case 8: run_pass<4, 4, 0, 4, MODE>(a, grid_x, batch); break;
        default: abort();
    }
}

static void dispatch_single(const PassArgs& a, u32 S, u32 batch) {
    switch (S) {
        case 4: run_pass<4, 0, 0, 0, PASS_SINGLE>(a, 1, batch); break;
        case 5: run_pass<4, 1, 0, 0, PASS_SINGLE>(a, 1, batch); break;
        case 6: run_pass<4, 2, 0, 0, PASS_SINGLE>(a, 1, batch); break;
        case 7: run_pass<4, 3, 0, 0, PASS_SINGLE>(a, 1, batch); break;
        case 8: run_pass<4, 4, 0, 0, PASS_SINGLE>(a, 1, batch); break;
        case 9: run_pass<4, 4, 1, 0, PASS_SINGLE>(a, 1, batch); break;
        case 10: run_pass<4, 4, 2, 0, PASS_SINGLE>(a, 1, batch); break;
        case 11: run_pass<4, 4, 3, 0, PASS_SINGLE>(a, 1, batch); break;
        case 12: run_pass<4, 4, 4, 0, PASS_SINGLE>(a, 1, batch); break;
        default: abort();
    }
}

static int emu_allow_expand = 1;
static unsigned long long emu_expand_plans = 0;
// 0: zero-padded transforms take the plain plan too; returns how many calls have taken the expansion plan so far
extern "C" unsigned long long emu_set_expand(int allow) { emu_allow_expand = allow; return emu_expand_plans; }
static int emu_force_ws = 0;
extern "C" void emu_set_force_ws(int v) { emu_force_ws = v; }
// operands that had to be canonical and were not, since the last reset (gl.hpp, BFS_CHECK_CANONICAL)
extern "C" unsigned long long emu_canonical_violations(int reset) {
    const unsigned long long c = gl_canonical_violations();
    if (reset) gl_canonical_violations() = 0;
    return c;
}

extern "C" int emu_gl_ntt(const u64* in, u64 n_in, u64 in_stride, u64* out, u64 out_stride, u32 log_n, u32 batch,
                          u64 root, u64 shift, u64 post_scale) {
    int rc = ntt_check_root(root, log_n);
    if (rc) return rc;
    const u64 n = 1ull << log_n;
    if (n_in > n) return BFS_ERR_TOO_MANY_COEFFS;
    NttPlan p;
    if (!ntt_make_plan(log_n, root, p)) return BFS_ERR_BAD_ARG;
    if (p.npass == 0) {
        SmallArgs a{in, out, in_stride, out_stride, n_in, log_n, root, shift, post_scale};
        for (u32 b = 0; b < batch; ++b)
            for (u32 k = 0; k < n; ++k) ntt_small_body(a, k, b);
        return 0;
    }
    NttHostTables ht;
    ntt_build_tables(p, root, post_scale, ht);
    CosetHostTables ct;
    const bool coset = shift != 1;
    if (coset) ntt_build_coset_tables(p, shift, ct);
    NttTables tb{ht.w_lo.data(), ht.w_hi.data(), p.lo_bits, p.t_in_log, ht.t_in.data(), ht.t_in_last.data(),
                 coset ? ct.s_lo.data() : nullptr, coset ? ct.s_hi.data() : nullptr, nullptr, nullptr};
    // the product's buffer flow (ntt.hip: ntt_launch): pass 0 in -> out, later passes in place on out; overlapping in / out go through
    // an intermediate buffer in passes 0 and 1.  `force_ws` lets a test take that route with separate buffers too.
    const u64* in_end = in + (u64)(batch - 1) * in_stride + n_in;
    const u64* out_end = out + (u64)(batch - 1) * out_stride + n;
    const bool overlap = p.npass > 1 && n_in != 0 && ((in < out_end && out < in_end) || emu_force_ws);
    // the expansion plan of a zero-padded transform (ntt.hip: ntt_launch takes it under the same condition)
    NttPlan xp;
    if (emu_allow_expand && !overlap && ntt_make_expand_plan(log_n, n_in, root, p, xp)) {
        ++emu_expand_plans;
        for (u32 t = 1; t < xp.npass; ++t) {
            const u32 S = xp.pass_bits[t];
            const u32 grid_x = (u32)((n >> S) >> xp.logC[t]);
            tb.row = nullptr; tb.srow = nullptr;
            if (t == 1) {
                PassArgs a = ntt_pass_args(xp, t, in, out, in_stride, out_stride, n_in, tb, coset, shift, post_scale);
                dispatch_multi<PASS_EXPAND>(a, S, grid_x, batch);
            } else {
                PassArgs a = ntt_pass_args(xp, t, out, out, out_stride, out_stride, n, tb, coset, shift, post_scale);
                dispatch_multi<PASS_COLUMN>(a, S, grid_x, batch);
            }
        }
        return 0;
    }
    std::vector<u64> ws;
    if (overlap) ws.resize((size_t)n * batch);
    std::vector<std::vector<u64>> rows(p.npass), srows(p.npass);
    for (u32 t = 0; t < p.npass; ++t) {
        const u64* src = out;
        u64* dst = out;
        u64 src_stride = out_stride, dst_stride = out_stride;
        if (t == 0) {
            src = in; src_stride = in_stride;
            if (overlap) { dst = ws.data(); dst_stride = n; }
        } else if (t == 1 && overlap) {
            src = ws.data(); src_stride = n;
        }
        tb.row = nullptr;
        tb.srow = nullptr;
        {
            NttRowSpec load, store;
            ntt_row_specs(p, t, root, load, store);
            if (load.omega) { ntt_product_table(load.omega, load.a_bits, load.b_bits, rows[t]); tb.row = rows[t].data(); }
            if (store.omega) { ntt_product_table(store.omega, store.a_bits, store.b_bits, srows[t]); tb.srow = srows[t].data(); }
        }
        PassArgs a = ntt_pass_args(p, t, src, dst, src_stride, dst_stride, t == 0 ? n_in : n, tb, coset, shift, post_scale);
        u32 grid_x = (u32)((n >> p.pass_bits[t]) >> p.logC[t]);
        if (p.npass == 1) dispatch_single(a, p.pass_bits[0], batch);
        else if (t == 0) dispatch_multi<PASS_FIRST>(a, p.pass_bits[t], grid_x, batch);
        else dispatch_multi<PASS_COLUMN>(a, p.pass_bits[t], grid_x, batch);
    }
    return 0;
}

extern "C" int emu_plan(u32 log_n, u64 root, u32* npass, u32* bits, u32* logc, u32* uinv) {
    NttPlan p;
    if (!ntt_make_plan(log_n, root, p)) return 1;
    *npass = p.npass; *uinv = p.uinv;
    for (int i = 0; i < 4; ++i) { bits[i] = p.pass_bits[i]; logc[i] = p.logC[i]; }
    return 0;
}

// what ntt_launch decides for a zero-padded transform (no arithmetic): 0 the plain plan, 1 an expansion plan; main_bits / extras as planned
extern "C" int emu_expand_plan(u32 log_n, u64 n_in, u64 root, u32* main_bits, u32* extras) {
    NttPlan p, xp;
    if (!ntt_make_plan(log_n, root, p) || p.npass == 0) return 0;
    if (!ntt_make_expand_plan(log_n, n_in, root, p, xp)) return 0;
    *main_bits = xp.main_bits; *extras = xp.extras;
    return 1;
}

// the twiddle schedule of three-pass plans: -1 environment, 0 load-time, 1 balanced; returns what a plan for log_n then uses
extern "C" int emu_set_schedule(int mode, u32 log_n, u64 root) {
    ntt_schedule_override() = mode;
    NttPlan p;
    if (!ntt_make_plan(log_n, root, p)) return -1;
    return (int)p.sched;
}
