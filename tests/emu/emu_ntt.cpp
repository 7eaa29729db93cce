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
    const u64* tab = (Cfg::U == 2 && MODE == PASS_FINAL) ? a.tb.t_in_last : a.tb.t_in;
    if (B2 > 0) for (u32 i = 0; i < tw.size(); ++i) tw[i] = tab[(u64)i << (a.tb.t_in_log - (B1 + B2))];
    for (u32 by = 0; by < batch; ++by)
        for (u32 bx = 0; bx < grid_x; ++bx) {
            const u64* rowtw = nullptr;
            std::vector<u64> row;
            if (MODE == PASS_COLUMN && a.tb.row != nullptr) {
                const u64 K = a.pass_index ? digit_reverse((u64)(bx >> a.lognl), a.pass_bits, 0, (int)a.pass_index - 1) : 0;
                row.assign(a.tb.row + (K << Cfg::S), a.tb.row + (K << Cfg::S) + (1u << Cfg::S));
                rowtw = row.data();
            }
            for (u32 t = 0; t < (u32)Cfg::W; ++t) ntt_stage1<B1, B2, B3, LOGC, MODE>(a, smem.data(), tw.data(), rowtw, t, bx, by);
            if (B2 > 0) for (u32 t = 0; t < (u32)Cfg::W; ++t) ntt_stage2<B1, B2, B3, LOGC, MODE>(a, smem.data(), t, bx, by);
            if (B3 > 0) for (u32 t = 0; t < (u32)Cfg::W; ++t) ntt_stage3<B1, B2, B3, LOGC, MODE>(a, smem.data(), t, bx, by);
        }
}

template <int MODE>
static void dispatch_multi(const PassArgs& a, u32 S, u32 grid_x, u32 batch) {
    switch (S) {
        case 4: run_pass<4, 0, 0, 8, MODE>(a, grid_x, batch); break;
        case 5: run_pass<4, 1, 0, 7, MODE>(a, grid_x, batch); break;
        case 6: run_pass<4, 2, 0, 6, MODE>(a, grid_x, batch); break;
        case 7: run_pass<4, 3, 0, 5, MODE>(a, grid_x, batch); break;
        case 8: run_pass<4, 4, 0, 4, MODE>(a, grid_x, batch); break;
        default: abort();
    }
}

static void dispatch_single(const PassArgs& a, u32 S, u32 batch) {
    switch (S) {
        case 4: run_pass<4, 0, 0, 0, PASS_FINAL>(a, 1, batch); break;
        case 5: run_pass<4, 1, 0, 0, PASS_FINAL>(a, 1, batch); break;
        case 6: run_pass<4, 2, 0, 0, PASS_FINAL>(a, 1, batch); break;
        case 7: run_pass<4, 3, 0, 0, PASS_FINAL>(a, 1, batch); break;
        case 8: run_pass<4, 4, 0, 0, PASS_FINAL>(a, 1, batch); break;
        case 9: run_pass<4, 4, 1, 0, PASS_FINAL>(a, 1, batch); break;
        case 10: run_pass<4, 4, 2, 0, PASS_FINAL>(a, 1, batch); break;
        case 11: run_pass<4, 4, 3, 0, PASS_FINAL>(a, 1, batch); break;
        case 12: run_pass<4, 4, 4, 0, PASS_FINAL>(a, 1, batch); break;
        default: abort();
    }
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
                 coset ? ct.s_lo.data() : nullptr, coset ? ct.s_hi.data() : nullptr, nullptr};
    std::vector<u64> ws;
    if (p.npass > 1) ws.resize((size_t)n * batch);
    std::vector<std::vector<u64>> rows(p.npass);
    for (u32 t = 0; t < p.npass; ++t) {
        const bool first = t == 0, last = t + 1 == p.npass;
        const u64* src = first ? in : ws.data();
        u64* dst = last ? out : ws.data();
        tb.row = nullptr;
        {
            u32 done = 0;
            for (u32 v = 0; v <= t; ++v) done += p.pass_bits[v];
            if (t > 0 && !last && done <= 16) {
                const u32 S = p.pass_bits[t];
                const u64 wN = gl_pow(root, 1ull << (log_n - done));
                rows[t].resize((size_t)1 << done);
                u64 wK = 1;
                for (u64 K = 0; K < (1ull << (done - S)); ++K) {
                    u64 v = 1;
                    for (u64 r = 0; r < (1ull << S); ++r) { rows[t][(K << S) + r] = v; v = gl_mul(v, wK); }
                    wK = gl_mul(wK, wN);
                }
                tb.row = rows[t].data();
            }
        }
        PassArgs a = ntt_pass_args(p, t, src, dst, first ? in_stride : n, last ? out_stride : n, first ? n_in : n, tb,
                                   coset, shift, post_scale);
        u32 grid_x = (u32)((n >> p.pass_bits[t]) >> p.logC[t]);
        if (p.npass == 1) dispatch_single(a, p.pass_bits[0], batch);
        else if (last) dispatch_multi<PASS_FINAL>(a, p.pass_bits[t], grid_x, batch);
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
