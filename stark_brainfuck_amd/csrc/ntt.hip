// ntt.hip -- gfx950 kernels and launcher for bfs_gl_ntt() (algorithm and reference citations: ntt_core.hpp)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <algorithm>
#include <atomic>
#include <vector>
#include <mutex>
#include <set>
#include <string>
#include <tuple>

// the four-instruction subtraction (gl.hpp: gl_sub4, explicit SGPR carries): 6.9 % fewer VALU instructions per launch of the tile kernels,
// 8 x 2^24 1.345 -> 1.31 ms on one box (profiles/r03/ab_sub4.txt)
#define BFS_GL_SUB4
#include "runtime.hpp"

namespace bfs {

// One workgroup per tile: T/16 threads hold 16 elements each.  LDS = tile (padded) + the stage-1 -> stage-2 twiddles.
// Used by single-pass plans (n <= 4096: up to three register stages).
// (A persistent variant with next-tile prefetch and 16-byte paired-lane accesses was measured slower: hardware workgroup turnover
//  already overlaps HBM latency)
template <int B1, int B2, int B3, int LOGC, int MODE, bool NT>
__global__ void __launch_bounds__(256) ntt_tile_kernel(const PassArgs a) {
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    u64* tw = smem + (B2 > 0 ? ((Cfg::LDS_WORDS + 1) & ~1) : 0);
    // Everything a workgroup needs from memory is requested up front: its entries of the two twiddle tables first (so that
    // waiting for them -- vmcnt counts in order -- does not wait for the data), then the 16 elements of every thread; the
    // tables go to LDS behind ONE barrier while the data is still in flight.  (With the table copies and their two barriers in
    // front of the loads, a workgroup spent an L2 round trip and a half before its HBM loads were even issued.)
    static_assert(B1 == 4, "one sub-group of 16 elements per thread");
    constexpr u32 TW_N = Cfg::U >= 2 ? (1u << (B1 + B2)) : 0;
    const u64* tab = a.tw1;
    const u32 tw_shift = a.tb.t_in_log - (B1 + B2);
    const u32 tid = threadIdx.x;
    // this tile's row of a product table (one coalesced 2^S-entry read per workgroup): factors of its input rows (last pass of a
    // balanced plan) or of its output rows (first pass); never both
    const u64* lrow = tile_load_row<Cfg, LOGC, MODE>(a, blockIdx.x);
    const u64* srow_g = tile_store_row<Cfg, LOGC, MODE>(a, blockIdx.x);
    const u64* row = lrow ? lrow : srow_g;
    const bool has_row = row != nullptr;
    u64 tw0 = 0, rw0 = 0;
    if (TW_N && tid < TW_N) tw0 = tab[(u64)tid << tw_shift];
    if (has_row && tid < (1u << Cfg::S)) rw0 = row[tid];
    u64 x[16];
    ntt_stage1_load<B1, B2, B3, LOGC, MODE, NT>(a, tid, blockIdx.x, blockIdx.y, 0, x);
    u64* rw = tw + Cfg::TW_WORDS;
    if (TW_N && tid < TW_N) tw[tid] = tw0;
    if (has_row && tid < (1u << Cfg::S)) rw[tid] = rw0;
    for (u32 i = tid + blockDim.x; i < TW_N; i += blockDim.x) tw[i] = tab[(u64)i << tw_shift];
    if (has_row)
        for (u32 i = tid + blockDim.x; i < (1u << Cfg::S); i += blockDim.x) rw[i] = row[i];
    const u64* rowtw = lrow ? rw : nullptr;
    const u64* srow = srow_g && !lrow ? rw : nullptr;
    if (Cfg::U >= 2 || has_row) __syncthreads();
    ntt_stage1_compute<B1, B2, B3, LOGC, MODE, NT>(a, smem, tw, rowtw, threadIdx.x, blockIdx.x, blockIdx.y, 0, x, srow);
    if constexpr (B2 > 0) {
        __syncthreads();
        ntt_stage2<B1, B2, B3, LOGC, MODE, NT>(a, smem, threadIdx.x, blockIdx.x, blockIdx.y, srow);
    }
    if constexpr (B3 > 0) {
        __syncthreads();
        ntt_stage3<B1, B2, B3, LOGC, MODE, NT>(a, smem, threadIdx.x, blockIdx.x, blockIdx.y, srow);
    }
}

// Two-stage tiles of multi-pass plans: the stage 1 -> 2 exchange goes through LDS in two halves -- the low
// 32-bit words of all 4096 values, then the high words -- so that the tile buffer is 17 KiB instead of 34 and a workgroup needs
// 21.5 KiB of LDS instead of 39.  With 39 KiB four workgroups (4 waves per SIMD) fit a CU, and a SIMD whose four waves are all
// waiting -- for their loads, at a barrier -- idles: VALU issue was 82 % of the time.  A timing-only experiment (the same kernel
// launched with less LDS than it indexes) put five or more workgroups per CU at 1.34 ms against 1.51 (profiles/r02).  The
// price: 32 + 32 four-byte LDS accesses per thread instead of 16 + 16 eight-byte ones and two more barriers per tile.
// six waves per SIMD asked of the register allocator (80 VGPRs): the store-time row of the balanced schedule took the column
// instantiation from 78 to 82 registers, i.e. from six waves to five (the 5- and 6-bit digits would spill 20 bytes at 80: left as the compiler has them)
#ifndef BFS_NTT_SPLIT_WAVES
#define BFS_NTT_SPLIT_WAVES 6
#endif
template <int B1, int B2, int B3, int LOGC, int MODE, bool NT>
__global__ void __launch_bounds__(256, B2 >= 3 ? BFS_NTT_SPLIT_WAVES : 0) ntt_tile_kernel_split(const PassArgs a) {
    static_assert(B1 == 4 && B2 > 0 && B3 == 0, "two register stages, 16 elements per thread");
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    constexpr u32 TILE_BYTES = (Cfg::LDS_WORDS * 4 + 15) & ~15u;
    u32* tile = (u32*)smem;
    u64* tw = (u64*)((char*)smem + TILE_BYTES);
    constexpr u32 TW_N = 1u << (B1 + B2);
    const u64* tab = a.tw1;                                                  // n^-1 folded in when this is the last pass
    const u32 tw_shift = a.tb.t_in_log - (B1 + B2);
    const u32 tid = threadIdx.x;
    const u32 bx = blockIdx.x, by = blockIdx.y;
    const u64* lrow = tile_load_row<Cfg, LOGC, MODE>(a, bx);
    const u64* srow_g = tile_store_row<Cfg, LOGC, MODE>(a, bx);
    const u64* row = lrow ? lrow : srow_g;
    const bool has_row = row != nullptr;
    // table entries first, then the data (see ntt_tile_kernel); W = 256 >= both table sizes
    u64 tw0 = 0, rw0 = 0;
    if (tid < TW_N) tw0 = tab[(u64)tid << tw_shift];
    if (has_row && tid < (1u << Cfg::S)) rw0 = row[tid];
    u64 x[16];
    ntt_stage1_load<B1, B2, B3, LOGC, MODE, NT>(a, tid, bx, by, 0, x);
    u64* rw = tw + Cfg::TW_WORDS;
    if (tid < TW_N) tw[tid] = tw0;
    if (has_row && tid < (1u << Cfg::S)) rw[tid] = rw0;
    __syncthreads();
    const u64* srow = srow_g && !lrow ? rw : nullptr;
    ntt_stage1_values<B1, B2, B3, LOGC, MODE>(a, tw, lrow ? rw : nullptr, tid, bx, by, 0, x);
    constexpr int Q2 = 1 << B2, SG2 = 16 / Q2;
    BFS_UNROLL
    for (int m = 0; m < 16; ++m) tile[stage1_out_index<B1, B2, B3, LOGC, MODE>(a, tid, 0, m)] = (u32)x[m];
    __syncthreads();
    u32 lo[16];
    BFS_UNROLL
    for (int s = 0; s < SG2; ++s)
        BFS_UNROLL
        for (int d = 0; d < Q2; ++d) lo[s * Q2 + d] = tile[stage2_in_index<B1, B2, B3, LOGC, MODE>(tid, s, d)];
    __syncthreads();
    BFS_UNROLL
    for (int m = 0; m < 16; ++m) tile[stage1_out_index<B1, B2, B3, LOGC, MODE>(a, tid, 0, m)] = (u32)(x[m] >> 32);
    __syncthreads();
    BFS_UNROLL
    for (int s = 0; s < SG2; ++s) {
        u64 y[Q2];
        BFS_UNROLL
        for (int d = 0; d < Q2; ++d) y[d] = ((u64)tile[stage2_in_index<B1, B2, B3, LOGC, MODE>(tid, s, d)] << 32) | lo[s * Q2 + d];
        ntt_stage2_from<B1, B2, B3, LOGC, MODE, NT>(a, smem, tid, bx, by, s, y, srow);
    }
}

template <int B1, int B2, int B3, int LOGC, int MODE>
static int launch_tile_split(const PassArgs& a, u32 grid_x, u32 batch, hipStream_t stream) {
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    const size_t row_words = (a.tb.row != nullptr || a.tb.srow != nullptr) ? (1u << Cfg::S) : 0;
    const size_t lds = ((Cfg::LDS_WORDS * 4 + 15) & ~15u) + (Cfg::TW_WORDS + row_words) * sizeof(u64);
    if (a.streaming)
        hipLaunchKernelGGL((ntt_tile_kernel_split<B1, B2, B3, LOGC, MODE, true>), dim3(grid_x, batch), dim3(Cfg::W), lds, stream, a);
    else
        hipLaunchKernelGGL((ntt_tile_kernel_split<B1, B2, B3, LOGC, MODE, false>), dim3(grid_x, batch), dim3(Cfg::W), lds, stream, a);
    BFS_HIP(hipGetLastError());
    return BFS_OK;
}

__global__ void ntt_small_kernel(const SmallArgs a) { ntt_small_body(a, threadIdx.x, blockIdx.y); }

template <int B1, int B2, int B3, int LOGC, int MODE>
static int launch_tile(const PassArgs& a, u32 grid_x, u32 batch, hipStream_t stream) {
    typedef TileCfg<B1, B2, B3, LOGC, MODE> Cfg;
    const size_t row_words = (a.tb.row != nullptr || a.tb.srow != nullptr) ? (1u << Cfg::S) : 0;
    const size_t lds = (size_t)((B2 > 0 ? ((Cfg::LDS_WORDS + 1) & ~1) + Cfg::TW_WORDS : 0) + row_words) * sizeof(u64);
    if (a.streaming)
        hipLaunchKernelGGL((ntt_tile_kernel<B1, B2, B3, LOGC, MODE, true>), dim3(grid_x, batch), dim3(Cfg::W), lds, stream, a);
    else
        hipLaunchKernelGGL((ntt_tile_kernel<B1, B2, B3, LOGC, MODE, false>), dim3(grid_x, batch), dim3(Cfg::W), lds, stream, a);
    BFS_HIP(hipGetLastError());
    return BFS_OK;
}

// multi-pass plans use 4096-element tiles (logC = 12 - S, S = 5..8: two register stages, split exchange); single-pass plans one column of 2^S rows (S = 4..12)
template <int MODE>
static int dispatch_multi(const PassArgs& a, u32 S, u32 grid_x, u32 batch, hipStream_t stream) {
    switch (S) {
        case 5: return launch_tile_split<4, 1, 0, 7, MODE>(a, grid_x, batch, stream);
        case 6: return launch_tile_split<4, 2, 0, 6, MODE>(a, grid_x, batch, stream);
        case 7: return launch_tile_split<4, 3, 0, 5, MODE>(a, grid_x, batch, stream);
        case 8: return launch_tile_split<4, 4, 0, 4, MODE>(a, grid_x, batch, stream);
    }
    set_error("internal: no tile kernel for a %u-bit digit of a multi-pass plan", S);
    return BFS_ERR_BAD_ARG;
}

static int dispatch_single(const PassArgs& a, u32 S, u32 batch, hipStream_t stream) {
    switch (S) {
        case 4: return launch_tile<4, 0, 0, 0, PASS_SINGLE>(a, 1, batch, stream);
        case 5: return launch_tile<4, 1, 0, 0, PASS_SINGLE>(a, 1, batch, stream);
        case 6: return launch_tile<4, 2, 0, 0, PASS_SINGLE>(a, 1, batch, stream);
        case 7: return launch_tile<4, 3, 0, 0, PASS_SINGLE>(a, 1, batch, stream);
        case 8: return launch_tile<4, 4, 0, 0, PASS_SINGLE>(a, 1, batch, stream);
        case 9: return launch_tile<4, 4, 1, 0, PASS_SINGLE>(a, 1, batch, stream);
        case 10: return launch_tile<4, 4, 2, 0, PASS_SINGLE>(a, 1, batch, stream);
        case 11: return launch_tile<4, 4, 3, 0, PASS_SINGLE>(a, 1, batch, stream);
        case 12: return launch_tile<4, 4, 4, 0, PASS_SINGLE>(a, 1, batch, stream);
    }
    set_error("internal: no tile kernel for a %u-bit single-pass transform", S);
    return BFS_ERR_BAD_ARG;
}

static int dispatch_tile(const NttPlan& p, u32 t, const PassArgs& a, u32 grid_x, u32 batch, hipStream_t stream) {
    if (p.npass == 1) return dispatch_single(a, p.pass_bits[0], batch, stream);
    if (t == 0) return dispatch_multi<PASS_FIRST>(a, p.pass_bits[t], grid_x, batch, stream);
    return dispatch_multi<PASS_COLUMN>(a, p.pass_bits[t], grid_x, batch, stream);
}

constexpr u64 NTT_STREAMING_BYTES = 128ull << 20;     // one 2^24-point column (128 MiB) still runs out of the Infinity Cache, two do not

enum { TBL_W_LO = 1, TBL_W_HI, TBL_T_IN, TBL_T_IN_LAST, TBL_S_LO, TBL_S_HI, TBL_ROW };

// the product tables of pass t (ntt_row_specs): <= 2^16 entries each (512 KiB, L2 resident), cached per (omega, shape)
static int get_row_tables(const NttPlan& p, u32 t, u64 root, const u64** d_row, const u64** d_srow) {
    *d_row = nullptr;
    *d_srow = nullptr;
    NttRowSpec load, store;
    ntt_row_specs(p, t, root, load, store);
    for (int which = 0; which < 2; ++which) {
        const NttRowSpec& sp = which ? store : load;
        if (sp.omega == 0) continue;
        const u64** out = which ? d_srow : d_row;
        const u64 key = ((u64)sp.a_bits << 24) | ((u64)sp.b_bits << 16) | TBL_ROW;
        if (cached_table_lookup(sp.omega, key, 0, out)) continue;
        std::vector<u64> host;
        ntt_product_table(sp.omega, sp.a_bits, sp.b_bits, host);
        BFS_TRY(cached_table(sp.omega, key, 0, host.data(), host.size(), out));
    }
    return BFS_OK;
}

static int get_tables(const NttPlan& p, u64 root, u64 shift, u64 post_scale, NttTables& tb) {
    tb = NttTables{};
    tb.lo_bits = p.lo_bits;
    tb.t_in_log = p.t_in_log;
    const u64 kb = ((u64)p.log_n << 8);
    bool have = cached_table_lookup(root, kb | TBL_W_LO, 0, &tb.w_lo) && cached_table_lookup(root, kb | TBL_W_HI, 0, &tb.w_hi) &&
                cached_table_lookup(root, kb | TBL_T_IN, 0, &tb.t_in) &&
                cached_table_lookup(root, kb | TBL_T_IN_LAST, post_scale, &tb.t_in_last);
    if (!have) {
        NttHostTables ht;
        ntt_build_tables(p, root, post_scale, ht);
        BFS_TRY(cached_table(root, kb | TBL_W_LO, 0, ht.w_lo.data(), ht.w_lo.size(), &tb.w_lo));
        BFS_TRY(cached_table(root, kb | TBL_W_HI, 0, ht.w_hi.data(), ht.w_hi.size(), &tb.w_hi));
        BFS_TRY(cached_table(root, kb | TBL_T_IN, 0, ht.t_in.data(), ht.t_in.size(), &tb.t_in));
        BFS_TRY(cached_table(root, kb | TBL_T_IN_LAST, post_scale, ht.t_in_last.data(), ht.t_in_last.size(), &tb.t_in_last));
    }
    if (shift != 1) {
        if (!(cached_table_lookup(shift, kb | TBL_S_LO, 0, &tb.s_lo) && cached_table_lookup(shift, kb | TBL_S_HI, 0, &tb.s_hi))) {
            CosetHostTables ct;
            ntt_build_coset_tables(p, shift, ct);
            BFS_TRY(cached_table(shift, kb | TBL_S_LO, 0, ct.s_lo.data(), ct.s_lo.size(), &tb.s_lo));
            BFS_TRY(cached_table(shift, kb | TBL_S_HI, 0, ct.s_hi.data(), ct.s_hi.size(), &tb.s_hi));
        }
    }
    return BFS_OK;
}

// two-level power tables of `root` (order 2^log_n): root^e = lo[e & mask] * hi[e >> lo_bits]; shared with the fold kernel
int ntt_power_tables(u64 root, u32 log_n, const u64** lo, const u64** hi, u32* lo_bits) {
    NttPlan p;
    if (!ntt_make_plan(log_n, root, p)) { set_error("no table plan for log_n = %u", log_n); return BFS_ERR_BAD_ARG; }
    NttTables tb;
    BFS_TRY(get_tables(p, root, 1, 1, tb));
    *lo = tb.w_lo; *hi = tb.w_hi; *lo_bits = tb.lo_bits;
    return BFS_OK;
}

// pass t of a plan.  ws == nullptr: pass 0 writes the output and every later pass runs in place there; ws given: pass 0 writes ws
// and pass 1 reads it (its tiles touch one 2^(S_0+S_1)-element block each: reading one buffer and writing another costs it
// nothing -- measured, profiles/r04/ab_ws_probe.txt), passes 2.. in place on the output
static int ntt_run_pass(const NttPlan& p, u32 t, NttTables tb, const u64* d_in, u64 n_in, u64 in_stride, u64* d_out, u64 out_stride, u64* ws,
                        u32 batch, u64 root, u64 shift, u64 post_scale, u32 streaming, hipStream_t stream) {
    const u64 n = 1ull << p.log_n;
    BFS_TRY(get_row_tables(p, t, root, &tb.row, &tb.srow));
    const u64* src = d_out;
    u64* dst = d_out;
    u64 src_stride = out_stride, dst_stride = out_stride;
    if (t == 0) {
        src = d_in; src_stride = in_stride;
        if (ws) { dst = ws; dst_stride = n; }
    } else if (t == 1 && ws) {
        src = ws; src_stride = n;
    }
    PassArgs a = ntt_pass_args(p, t, src, dst, src_stride, dst_stride, t == 0 ? n_in : n, tb, shift != 1, shift, post_scale);
    a.streaming = streaming;
    u32 grid_x = (u32)((n >> p.pass_bits[t]) >> p.logC[t]);
    return dispatch_tile(p, t, a, grid_x, batch, stream);
}

// Where pass 0 of a LARGE out-of-place transform writes.  Pass 0 is the one pass that streams one buffer in and another out with
// long strides, and how fast that goes depends on the PAIR of buffers -- some property of their physical placement that user space
// cannot see: 405-510 us for the same launch between different pairs of 1 GiB buffers of one process, while the in-place passes do
// not move (profiles/r03/buffer_placement.txt, profiles/r04/ab_ws_probe.txt; address-translation counters are flat, so it is not the
// TLB).  The library cannot move the caller's buffers, but it can put one of its own in between at no cost in traffic: pass 0 ->
// intermediate, pass 1 intermediate -> output (pass 1 is indifferent to being out of place).
//
// Since round 5 this is OPT-IN (round-4 advice: a measurement hidden inside the third call of a stream-ordered entry point allocated
// three buffers of the transform's size, synchronised the stream and broke under stream capture): a caller that keeps coming back with
// the same (input, output) pair -- the bench step, a prover's pooled buffers -- calls bfs_ntt_tune() ONCE, outside anything it times or
// captures; bfs_gl_ntt itself only looks the pair up and never measures, allocates candidates or synchronises.  ntt_tune times passes
// 0 + 1 on the direct route and through each of NTT_ROUTE_CANDIDATES library buffers in the state the transform will run in, the
// power-limited clock: NTT_ROUTE_WARM untimed rounds over all routes first, then NTT_ROUTE_REPS timed ones, every other one
// backwards, median per route (18 rounds x 8 launches = ~63 ms at 8 x 2^24, one stream synchronisation).  A short probe straight after
// idle (one warm-up round, minimum of four) read 0.89-0.94 ms for routes that run at 0.85 and did not tell fast from slow: 5 of 12
// processes ended on a slow pair against 0 of 12 with the long one (profiles/r04/ab_ws_probe.txt).  Only transforms of >=
// NTT_ROUTE_MIN_BYTES.  A remembered route dies with either buffer: bfs_free / bfs_free_async of a block drops every pair that
// touches it (ntt_route_forget_range, called by the pool), and bfs_ntt_route_forget() is there for memory the library does not own.
// BFS_NTT_WS_PROBE: "0" never route (tune becomes a no-op), "direct" / "buffer0..2" that route for every large transform without
// measuring (the GPU tests run a large transform over every route), "auto" the round-4 behaviour (bfs_gl_ntt tunes a pair by itself
// the third time it sees it).  BFS_NTT_WS_PROBE_LOG=1: the measurements go to stderr.
constexpr int NTT_ROUTE_CANDIDATES = 3;
constexpr int NTT_ROUTE_SIGHTINGS = 3;
#ifndef NTT_ROUTE_WARM
#define NTT_ROUTE_WARM 10
#define NTT_ROUTE_REPS 8
#endif
constexpr int NTT_ROUTE_SLOT0 = 16;                      // workspace slots 16.. hold the candidates
constexpr u64 NTT_ROUTE_MIN_BYTES = 256ull << 20;
constexpr int ROUTE_UNSEEN = -100;
namespace {
std::mutex g_route_mu;
struct RouteKey {
    int dev; hipStream_t stream; const void* in; const void* out; u64 in_stride, out_stride, shape; u64 in_bytes, out_bytes;
    bool operator<(const RouteKey& o) const {
        return std::tie(dev, stream, in, out, in_stride, out_stride, shape) < std::tie(o.dev, o.stream, o.in, o.out, o.in_stride, o.out_stride, o.shape);
    }
};
std::map<RouteKey, int> g_routes;                       // a route (>= -1), or ROUTE_UNSEEN - sightings so far ("auto" mode)
std::set<std::pair<int, hipStream_t>> g_candidate_owners;   // (device, stream) pairs that may hold candidate buffers
struct { float us[NTT_ROUTE_CANDIDATES + 1] = {0}; int route = -1; unsigned long long probes = 0; } g_last_probe;      // (under g_route_mu)
int route_mode() {
    static const int mode = [] {
        const char* e = getenv("BFS_NTT_WS_PROBE");
        if (!e) return -2;                                                   // default: remembered routes only (bfs_ntt_tune)
        if (e[0] == '0' || !strcmp(e, "direct")) return -1;
        if (!strncmp(e, "buffer", 6) && e[6] >= '0' && e[6] < '0' + NTT_ROUTE_CANDIDATES && !e[7]) return e[6] - '0';
        if (!strcmp(e, "auto") || !strcmp(e, "1")) return -3;
        return -2;
    }();
    return mode;
}
// the candidate buffers no remembered pair of (dev, stream) is routed through go back to the driver; the stream must be idle
void release_unused_candidates_locked(int dev, hipStream_t stream) {
    bool used[NTT_ROUTE_CANDIDATES] = {false};
    for (const auto& kv : g_routes)
        if (kv.first.dev == dev && kv.first.stream == stream && kv.second >= 0) used[kv.second] = true;
    for (int k = 0; k < NTT_ROUTE_CANDIDATES; ++k)
        if (!used[k]) (void)workspace_release(NTT_ROUTE_SLOT0 + k, stream);
}
// hipEvents of one measurement: destroyed on every way out of ntt_tune (the round-4 version leaked all of them when a launch failed)
struct EventGrid {
    std::vector<hipEvent_t> ev;
    int make(size_t count) {
        ev.reserve(count);
        for (size_t i = 0; i < count; ++i) {
            hipEvent_t e = nullptr;
            BFS_HIP(hipEventCreate(&e));
            ev.push_back(e);
        }
        return BFS_OK;
    }
    ~EventGrid() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
};
}
// what the last route measurement of this process read (bfs_ntt_route_probe_info; bench.py prints it next to the step it explains)
int ntt_route_probe_info(float* us, int* route, unsigned long long* probes) {
    std::lock_guard<std::mutex> lock(g_route_mu);
    if (us) for (int k = 0; k <= NTT_ROUTE_CANDIDATES; ++k) us[k] = g_last_probe.us[k];
    if (route) *route = g_last_probe.route;
    if (probes) *probes = g_last_probe.probes;
    return BFS_OK;
}

// forget every remembered pair with a buffer inside [lo, lo + bytes) (bytes == 0: the pair whose buffer STARTS at lo; lo == nullptr:
// everything); candidate buffers that no pair needs any more are freed when `may_free` (the device must then be idle on those streams:
// the callers below synchronise first).  Returns the number of pairs forgotten.
size_t ntt_route_forget_range(const void* lo, size_t bytes, bool may_free) {
    std::lock_guard<std::mutex> lock(g_route_mu);
    size_t gone = 0;
    std::vector<std::pair<int, hipStream_t>> touched;
    for (auto it = g_routes.begin(); it != g_routes.end();) {
        const RouteKey& k = it->first;
        auto hits = [&](const void* p, u64 span) {
            if (lo == nullptr) return true;
            const char *a = (const char*)p, *b = (const char*)lo;
            if (bytes == 0) return a == b;
            return a < b + bytes && b < a + span;
        };
        if (hits(k.in, k.in_bytes) || hits(k.out, k.out_bytes)) {
            touched.emplace_back(k.dev, k.stream);
            it = g_routes.erase(it);
            ++gone;
        } else {
            ++it;
        }
    }
    if (may_free) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        for (const auto& ds : touched) {
            if (ds.first != cur && hipSetDevice(ds.first) != hipSuccess) continue;
            if (hipStreamSynchronize(ds.second) == hipSuccess) release_unused_candidates_locked(ds.first, ds.second);
            else (void)hipGetLastError();
        }
        (void)hipSetDevice(cur);
    }
    return gone;
}

// bfs_pool_trim (the device is idle): candidate buffers that no remembered pair is routed through any more go back to the driver
void ntt_route_trim() {
    std::lock_guard<std::mutex> lock(g_route_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (const auto& ds : g_candidate_owners)
        if (ds.first == cur) release_unused_candidates_locked(ds.first, ds.second);
}

static RouteKey route_key(int dev, hipStream_t stream, const NttPlan& p, const u64* d_in, u64 n_in, u64 in_stride, const u64* d_out, u64 out_stride, u32 batch) {
    const u64 n = 1ull << p.log_n;
    return RouteKey{dev, stream, d_in, d_out, in_stride, out_stride, ((u64)p.log_n << 32) | batch,
                    ((u64)(batch - 1) * in_stride + n_in) * sizeof(u64), ((u64)(batch - 1) * out_stride + n) * sizeof(u64)};
}

// the measurement itself (bfs_ntt_tune, or bfs_gl_ntt in "auto" mode).  Overwrites the output with passes 0 + 1 of the transform,
// synchronises the stream.  *route: -1 direct, k >= 0 through candidate buffer k.  Not being able to measure (no memory for the
// candidates) is not an error: the pair stays direct.
static int ntt_measure_route(const NttPlan& p, const NttTables& tb, const RouteKey& key, const u64* d_in, u64 n_in, u64 in_stride, u64* d_out,
                             u64 out_stride, u32 batch, u64 root, u64 shift, u64 post_scale, u32 streaming, hipStream_t stream, int* route) {
    *route = -1;
    static const bool log = [] { const char* e = getenv("BFS_NTT_WS_PROBE_LOG"); return e && e[0] == '1'; }();
    const u64 n = 1ull << p.log_n;
    const size_t bytes = (size_t)n * batch * sizeof(u64);
    constexpr int R = NTT_ROUTE_CANDIDATES + 1, REPS = NTT_ROUTE_REPS, WARM = NTT_ROUTE_WARM;
    auto stay_direct = [&](const char* why) {
        if (log) fprintf(stderr, "bfs ntt route: in %p out %p 2^%u x %u: not measured (%s) -> direct\n", (const void*)d_in, (void*)d_out, p.log_n, batch, why);
        std::lock_guard<std::mutex> lock(g_route_mu);
        g_routes[key] = -1;
        return BFS_OK;
    };
    // room for the candidates AND for whatever the caller allocates next: four transform sizes free, or the pair stays direct
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return stay_direct("hipMemGetInfo failed"); }
    size_t held = 0;                                     // candidates of this stream that are already allocated count as free
    {
        void* w = nullptr;
        for (int k = 0; k < NTT_ROUTE_CANDIDATES; ++k)
            if (workspace_peek(NTT_ROUTE_SLOT0 + k, stream, &w, nullptr) && w) held += bytes;
    }
    if (free_b + held < 4 * bytes) return stay_direct("less than four transform sizes of free memory");
    u64* cand[R] = {nullptr};                            // [0]: direct
    { std::lock_guard<std::mutex> lock(g_route_mu); g_candidate_owners.emplace(key.dev, stream); }
    for (int k = 0; k < NTT_ROUTE_CANDIDATES; ++k) {
        void* w = nullptr;
        if (workspace(NTT_ROUTE_SLOT0 + k, bytes, stream, &w) != BFS_OK) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(stream);
            { std::lock_guard<std::mutex> lock(g_route_mu); g_routes[key] = -1; release_unused_candidates_locked(key.dev, stream); }
            return stay_direct("no memory for the candidate buffers");
        }
        cand[k + 1] = (u64*)w;
    }
    int rc = BFS_OK;
    float ms[R] = {0};
    {
        EventGrid grid;                                  // REPS x R x {start, stop}; gone when this block ends, however it ends
        rc = grid.make((size_t)REPS * R * 2);
        auto ev = [&](int rep, int r, int which) { return grid.ev[((size_t)rep * R + r) * 2 + which]; };
        // untimed rounds first (a freshly allocated buffer is slow the first time it is written, 1.2 ms against 0.85, and the clock takes
        // tens of ms of load to settle at the power limit), then REPS timed rounds over all routes; the median per route counts
        for (int rep = -WARM; rc == BFS_OK && rep < REPS; ++rep)
            for (int k = 0; rc == BFS_OK && k < R; ++k) {
                // (every other round backwards: while the clock is still ramping after idle, whatever is measured later in a round looks
                //  faster -- the first version always found direct > buffer 0 > buffer 1 > buffer 2, the order it measured them in)
                const int r = (rep & 1) ? R - 1 - k : k;
                if (rep >= 0 && hipEventRecord(ev(rep, r, 0), stream) != hipSuccess) { set_error("hipEventRecord failed in the route measurement"); rc = BFS_ERR_HIP; break; }
                for (u32 t = 0; rc == BFS_OK && t < 2; ++t)
                    rc = ntt_run_pass(p, t, tb, d_in, n_in, in_stride, d_out, out_stride, cand[r], batch, root, shift, post_scale, streaming, stream);
                if (rc == BFS_OK && rep >= 0 && hipEventRecord(ev(rep, r, 1), stream) != hipSuccess) { set_error("hipEventRecord failed in the route measurement"); rc = BFS_ERR_HIP; }
            }
        if (hipStreamSynchronize(stream) != hipSuccess && rc == BFS_OK) { set_error("hipStreamSynchronize failed in the route measurement"); rc = BFS_ERR_HIP; }
        for (int r = 0; rc == BFS_OK && r < R; ++r) {
            float t[REPS];
            for (int rep = 0; rep < REPS; ++rep)
                if (hipEventElapsedTime(&t[rep], ev(rep, r, 0), ev(rep, r, 1)) != hipSuccess) { set_error("hipEventElapsedTime failed in the route measurement"); rc = BFS_ERR_HIP; break; }
            std::sort(t, t + REPS);
            ms[r] = 0.5f * (t[(REPS - 1) / 2] + t[REPS / 2]);
        }
    }
    if (rc != BFS_OK) {                                   // a failed measurement leaves nothing behind: no events (above), no candidates, no route
        (void)hipGetLastError();
        (void)hipStreamSynchronize(stream);
        std::lock_guard<std::mutex> lock(g_route_mu);
        g_routes.erase(key);
        release_unused_candidates_locked(key.dev, stream);
        return rc;
    }
    int best = 0;
    for (int r = 1; r < R; ++r) if (ms[r] < ms[best]) best = r;
    if (ms[0] <= ms[best] * 1.01f) best = 0;             // the direct route unless an intermediate buffer is clearly faster
    if (log) {
        fprintf(stderr, "bfs ntt route: in %p out %p 2^%u x %u: passes 0+1 direct %.1f us", (const void*)d_in, (void*)d_out, p.log_n, batch, ms[0] * 1e3);
        for (int k = 1; k <= NTT_ROUTE_CANDIDATES; ++k) fprintf(stderr, ", via buffer %d (%p) %.1f us", k - 1, (void*)cand[k], ms[k] * 1e3);
        fprintf(stderr, " -> %s\n", best == 0 ? "direct" : (std::string("buffer ") + std::to_string(best - 1)).c_str());
    }
    *route = best - 1;
    // the candidates that lost go back to the driver (the stream is idle: synchronised above).  Slot NTT_ROUTE_SLOT0 + k is buffer k for
    // every pair of this stream, so a buffer another pair was routed through must stay
    {
        std::lock_guard<std::mutex> lock(g_route_mu);
        if (g_routes.size() >= 256 && !g_routes.count(key)) g_routes.clear();
        g_routes[key] = *route;
        for (int r = 0; r < R; ++r) g_last_probe.us[r] = ms[r] * 1e3f;
        g_last_probe.route = *route;
        ++g_last_probe.probes;
        release_unused_candidates_locked(key.dev, stream);
    }
    return BFS_OK;
}

// bfs_gl_ntt's side: the remembered route of the pair, nothing else (unless BFS_NTT_WS_PROBE forces a route or asks for "auto")
static int ntt_route(const NttPlan& p, const NttTables& tb, const u64* d_in, u64 n_in, u64 in_stride, u64* d_out, u64 out_stride, u32 batch,
                     u64 root, u64 shift, u64 post_scale, u32 streaming, hipStream_t stream, int* route) {
    *route = -1;
    const u64 n = 1ull << p.log_n;
    const int mode = route_mode();
    if (mode == -1 || n_in != n || (u64)n * batch * sizeof(u64) < NTT_ROUTE_MIN_BYTES) return BFS_OK;
    if (mode >= 0) { *route = mode; return BFS_OK; }
    int dev = 0;
    BFS_HIP(hipGetDevice(&dev));
    const RouteKey key = route_key(dev, stream, p, d_in, n_in, in_stride, d_out, out_stride, batch);
    {
        std::lock_guard<std::mutex> lock(g_route_mu);
        auto it = g_routes.find(key);
        if (it != g_routes.end() && it->second >= -1) { *route = it->second; return BFS_OK; }
        if (mode != -3) return BFS_OK;                                           // default: pairs nobody tuned run direct
        if (g_routes.size() >= 256 && it == g_routes.end()) { g_routes.clear(); it = g_routes.end(); }
        int& state = it != g_routes.end() ? it->second : g_routes.emplace(key, ROUTE_UNSEEN).first->second;
        if (ROUTE_UNSEEN - --state < NTT_ROUTE_SIGHTINGS) return BFS_OK;         // "auto": direct until the pair has come back often enough
    }
    return ntt_measure_route(p, tb, key, d_in, n_in, in_stride, d_out, out_stride, batch, root, shift, post_scale, streaming, stream, route);
}

// bfs_ntt_tune (include/bfstark.h)
int ntt_tune(const u64* d_in, u64 in_stride, u64* d_out, u64 out_stride, u32 log_n, u32 batch, u64 root, hipStream_t stream, int* route_out) {
    if (route_out) *route_out = -1;
    if (log_n > 32 || batch == 0 || batch > 65535 || d_in == nullptr || d_out == nullptr) { set_error("bfs_ntt_tune: bad argument"); return BFS_ERR_BAD_ARG; }
    const u64 n = 1ull << log_n;
    if (batch > 1 && (out_stride < n || in_stride < n)) { set_error("bfs_ntt_tune: transforms of a batch overlap"); return BFS_ERR_BAD_ARG; }
    int rc = ntt_check_root(root, log_n);
    if (rc != BFS_OK) { set_error("bfs_ntt_tune: the root is not a primitive 2^%u-th root of unity", log_n); return rc; }
    NttPlan p;
    if (!ntt_make_plan(log_n, root, p)) { set_error("no NTT plan for log_n = %u", log_n); return BFS_ERR_BAD_ARG; }
    const u64* in_end = d_in + (u64)(batch - 1) * in_stride + n;
    const u64* out_end = d_out + (u64)(batch - 1) * out_stride + n;
    const bool overlap = d_in < out_end && d_out < in_end;
    const int mode = route_mode();
    if (p.npass < 2 || overlap || (mode != -2 && mode != -3) || (u64)n * batch * sizeof(u64) < NTT_ROUTE_MIN_BYTES) return BFS_OK;   // nothing to choose
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
        set_error("bfs_ntt_tune: the stream is being captured (the measurement synchronises it)");
        return BFS_ERR_BAD_ARG;
    }
    (void)hipGetLastError();
    NttTables tb;
    BFS_TRY(get_tables(p, root, 1, 1, tb));
    const u32 streaming = (u32)((u64)n * batch * sizeof(u64) > NTT_STREAMING_BYTES);
    int dev = 0;
    BFS_HIP(hipGetDevice(&dev));
    const RouteKey key = route_key(dev, stream, p, d_in, n, in_stride, d_out, out_stride, batch);
    int route = -1;
    BFS_TRY(ntt_measure_route(p, tb, key, d_in, n, in_stride, d_out, out_stride, batch, root, 1, 1, streaming, stream, &route));
    if (route_out) *route_out = route;
    return BFS_OK;
}

int ntt_launch(const u64* d_in, u64 n_in, u64 in_stride, u64* d_out, u64 out_stride, u32 log_n, u32 batch, u64 root,
               u64 shift, u64 post_scale, hipStream_t stream) {
    if (log_n > 32) { set_error("field has no 2^%u-th root of unity (algebra.py:124-125)", log_n); return BFS_ERR_BAD_ARG; }
    const u64 n = 1ull << log_n;
    if (n_in > n) { set_error("more coefficients (%llu) than the evaluation order (%llu)", (unsigned long long)n_in, (unsigned long long)n); return BFS_ERR_TOO_MANY_COEFFS; }
    if (batch == 0) return BFS_OK;
    if (d_out == nullptr || (d_in == nullptr && n_in != 0)) { set_error("bfs_gl_ntt: null device pointer"); return BFS_ERR_BAD_ARG; }
    if (batch > 1 && (out_stride < n || in_stride < n_in)) {
        set_error("bfs_gl_ntt: transforms of a batch overlap (in_stride %llu < %llu coefficients or out_stride %llu < n = %llu)",
                  (unsigned long long)in_stride, (unsigned long long)n_in, (unsigned long long)out_stride, (unsigned long long)n);
        return BFS_ERR_BAD_ARG;
    }
    if (batch > 65535) {
        // grid.y carries the batch index and is limited to 65535: larger batches go in slices (transforms are independent)
        for (u32 done = 0; done < batch;) {
            const u32 part = batch - done < 65535 ? batch - done : 65535;
            BFS_TRY(ntt_launch(d_in + (u64)done * in_stride, n_in, in_stride, d_out + (u64)done * out_stride, out_stride, log_n, part, root,
                               shift, post_scale, stream));
            done += part;
        }
        return BFS_OK;
    }
    int rc = ntt_check_root(root, log_n);
    if (rc == BFS_ERR_NOT_ROOT) { set_error("primitive root must be nth root of unity, where n is %llu", (unsigned long long)n); return rc; }
    if (rc == BFS_ERR_NOT_PRIMITIVE) { set_error("primitive root %llu is not primitive nth root of unity, where n is %llu", (unsigned long long)root, (unsigned long long)n); return rc; }
    NttPlan p;
    if (!ntt_make_plan(log_n, root, p)) { set_error("no NTT plan for log_n = %u", log_n); return BFS_ERR_BAD_ARG; }
    if (p.npass == 0) {
        SmallArgs a{d_in, d_out, in_stride, out_stride, n_in, log_n, root, shift, post_scale};
        hipLaunchKernelGGL(ntt_small_kernel, dim3(1, batch), dim3(64), 0, stream, a);
        BFS_HIP(hipGetLastError());
        return BFS_OK;
    }
    NttTables tb;
    BFS_TRY(get_tables(p, root, shift, post_scale, tb));
    // non-temporal data accesses once a buffer of the call no longer fits the Infinity Cache next to its neighbours (ntt_core.hpp);
    // BFS_NTT_STREAMING=0 / 1 forces the choice (A/B, tools/ab_ntt.sh)
    static const int force_streaming = [] { const char* e = getenv("BFS_NTT_STREAMING"); return e ? atoi(e) : -1; }();
    const u32 streaming = force_streaming >= 0 ? (u32)(force_streaming != 0) : (u32)((u64)n * batch * sizeof(u64) > NTT_STREAMING_BYTES);
    // Where the passes run.  Pass 0 transposes (it cannot run in place); every later pass rewrites the slots it read.  Separate
    // input and output: in -> out, then in place on out -- no intermediate buffer.  Input and output overlapping (a transform "in
    // place" for the caller): in -> intermediate, intermediate -> out in pass 1 (whose tiles touch one 2^(S_0+S_1)-element block
    // each, so reading one buffer and writing another costs it nothing), then in place on out.
    const u64* in_end = d_in + (u64)(batch - 1) * in_stride + n_in;
    const u64* out_end = d_out + (u64)(batch - 1) * out_stride + n;
    const bool overlap = p.npass > 1 && n_in != 0 && d_in < out_end && d_out < in_end;
    // Zero-padded transforms whose coefficients fill at most 1/16 of the domain (every trace column's low-degree extension): the
    // expansion plan starts at the second digit and saves a pass (ntt_plan.hpp: ntt_make_expand_plan; BFS_NTT_EXPAND=0: never)
    static const bool allow_expand = [] { const char* e = getenv("BFS_NTT_EXPAND"); return !(e && e[0] == '0'); }();
    NttPlan xp;
    static const bool log_plans = getenv("BFS_NTT_PLAN_LOG") != nullptr;
    if (log_plans) fprintf(stderr, "ntt plan: log_n %u n_in %llu batch %u overlap %d allow %d in %p out %p\n", log_n, (unsigned long long)n_in, batch, (int)overlap, (int)allow_expand, (const void*)d_in, (void*)d_out);
    if (allow_expand && !overlap && ntt_make_expand_plan(log_n, n_in, root, p, xp)) {
        for (u32 t = 1; t < xp.npass; ++t) {
            const u32 S = xp.pass_bits[t];
            const u32 grid_x = (u32)((n >> S) >> xp.logC[t]);
            if (t == 1) {
                PassArgs a = ntt_pass_args(xp, t, d_in, d_out, in_stride, out_stride, n_in, tb, shift != 1, shift, post_scale);
                a.streaming = streaming;
                BFS_TRY(dispatch_multi<PASS_EXPAND>(a, S, grid_x, batch, stream));
            } else {
                PassArgs a = ntt_pass_args(xp, t, d_out, d_out, out_stride, out_stride, n, tb, shift != 1, shift, post_scale);
                a.streaming = streaming;
                BFS_TRY(dispatch_multi<PASS_COLUMN>(a, S, grid_x, batch, stream));
            }
        }
        return BFS_OK;
    }
    // Large out-of-place transforms: which buffer pass 0 writes to is chosen by measurement (ntt_route above)
    int route = -1;
    if (!overlap && p.npass > 1) BFS_TRY(ntt_route(p, tb, d_in, n_in, in_stride, d_out, out_stride, batch, root, shift, post_scale, streaming, stream, &route));
    u64* ws = nullptr;
    if (overlap || route >= 0) {
        void* w = nullptr;
        BFS_TRY(workspace(route >= 0 ? NTT_ROUTE_SLOT0 + route : 0, (size_t)n * batch * sizeof(u64), stream, &w));
        ws = (u64*)w;
    }
    for (u32 t = 0; t < p.npass; ++t)
        BFS_TRY(ntt_run_pass(p, t, tb, d_in, n_in, in_stride, d_out, out_stride, ws, batch, root, shift, post_scale, streaming, stream));
    return BFS_OK;
}

// ---- element-wise kernels (ntt.py:76, ntt.py:177-188, univariate.py:168-169) ----
__global__ void gl_mul_pointwise_kernel(const u64* a, const u64* b, u64* out, u64 n) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) out[i] = gl_mul(a[i], b[i]);
}

// batch_inverse (ntt.py:177-188) by Montgomery's trick at workgroup scope: 2048 elements share ONE field inversion.
// Thread t holds elements base + k * 256 + t (k < 8, coalesced), multiplies them up, the 256 thread products are scanned from
// both ends in LDS (Kogge-Stone, 8 steps each), thread 0 inverts the workgroup's product (the only a^(p-2): 64 squarings), and
// every thread unwinds: 1 / (its product) = 1 / total * (product of the threads before) * (product of the threads after), then
// element by element.  ~5 multiplications per element instead of ~96.  Zeros (the reference asserts there are none, ntt.py:180
// "batch inverse does not work when input contains a zero") are taken out of the products, reported through `zero_flag` and
// get inverse(0) = 0 (algebra.py:101-103) in the output.
constexpr int BINV_T = 256, BINV_E = 8;
__global__ void __launch_bounds__(BINV_T) gl_batch_inverse_kernel(const u64* in, u64* out, u64 n, unsigned int* zero_flag) {
    __shared__ u64 pre[BINV_T], suf[BINV_T];
    __shared__ u64 inv_total;
    const u32 tid = threadIdx.x;
    const u64 base = (u64)blockIdx.x * (BINV_T * BINV_E);
    u64 v[BINV_E], before[BINV_E];
    bool zero[BINV_E];
    u64 prod = 1;
    bool any_zero = false;
    BFS_UNROLL
    for (int k = 0; k < BINV_E; ++k) {
        const u64 i = base + (u64)k * BINV_T + tid;
        const u64 x = i < n ? in[i] : 1;
        zero[k] = x == 0;
        any_zero |= zero[k];
        v[k] = zero[k] ? 1 : x;
        before[k] = prod;                        // product of this thread's elements 0..k-1
        prod = gl_mul(prod, v[k]);
    }
    if (any_zero) *(volatile unsigned int*)zero_flag = 1u;    // pinned host memory; every writer stores the same value
    pre[tid] = prod;
    suf[tid] = prod;
    __syncthreads();
    // inclusive scans: pre[t] = prod of threads 0..t, suf[t] = prod of threads t..255
    for (u32 d = 1; d < BINV_T; d <<= 1) {
        const u64 a = tid >= d ? pre[tid - d] : 1, b = tid + d < BINV_T ? suf[tid + d] : 1;
        const u64 p0 = pre[tid], s0 = suf[tid];
        __syncthreads();
        pre[tid] = gl_mul(p0, a);
        suf[tid] = gl_mul(s0, b);
        __syncthreads();
    }
    if (tid == 0) inv_total = gl_inv(pre[BINV_T - 1]);
    __syncthreads();
    u64 run = inv_total;                         // -> 1 / (product of this thread's elements)
    if (tid > 0) run = gl_mul(run, pre[tid - 1]);
    if (tid + 1 < BINV_T) run = gl_mul(run, suf[tid + 1]);
    BFS_UNROLL
    for (int k = BINV_E - 1; k >= 0; --k) {
        const u64 i = base + (u64)k * BINV_T + tid;
        const u64 r = gl_mul(run, before[k]);    // 1 / v[k]
        if (i < n) out[i] = zero[k] ? 0 : r;
        run = gl_mul(run, v[k]);
    }
}

// ---- the same two over the cubic extension (limb planes): the Hadamard product of fast_multiply (ntt.py:76) and the batch_inverse of
// fast_coset_divide (ntt.py:226) when Table.ldex interpolates extension columns (table.py:133-134 -> ntt.py:126-161 -> 82-98 -> 45-79)
__global__ void xfe_mul_pointwise_kernel(const u64* a, u64 a_stride, const u64* b, u64 b_stride, u64* out, u64 out_stride, u64 n) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const Xfe x{{a[i], a[a_stride + i], a[2 * a_stride + i]}}, y{{b[i], b[b_stride + i], b[2 * b_stride + i]}};
        const Xfe r = xfe_mul(x, y);
        out[i] = r.c[0]; out[out_stride + i] = r.c[1]; out[2 * out_stride + i] = r.c[2];
    }
}

// 1 / a = adj(M_a) e_0 / det(M_a), M_a the matrix of multiplication by a = a0 + a1 X + a2 X^2 modulo X^3 - X + 1:
//     M_a = [ a0  -a2      -a1     ]
//           [ a1   a0+a2    a1-a2  ]
//           [ a2   a1       a0+a2  ]
// det(M_a) is the norm of a, an element of F_p that is zero only for a = 0: the norms go through the base field's batch inversion
// (one field inversion per 2048 elements) and the cofactors are scaled by the result.
__global__ void xfe_cofactors_kernel(const u64* in, u64 in_stride, u64* out, u64 out_stride, u64* norm, u64 n) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 a0 = in[i], a1 = in[in_stride + i], a2 = in[2 * in_stride + i];
        const u64 s = gl_add(a0, a2), d = gl_sub(a1, a2);
        const u64 c0 = gl_sub(gl_mul(s, s), gl_mul(d, a1));                      // (a0+a2)^2 - (a1-a2) a1
        const u64 c1 = gl_sub(gl_mul(d, a2), gl_mul(a1, s));                      // (a1-a2) a2 - a1 (a0+a2)
        const u64 c2 = gl_sub(gl_mul(a1, a1), gl_mul(s, a2));                     // a1^2 - (a0+a2) a2
        norm[i] = gl_sub(gl_mul(a0, c0), gl_add(gl_mul(a2, c1), gl_mul(a1, c2))); // first row of M_a times the cofactors
        out[i] = c0; out[out_stride + i] = c1; out[2 * out_stride + i] = c2;
    }
}

__global__ void xfe_scale_by_kernel(u64* x, u64 stride, const u64* f, u64 n) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 s = f[i];
        x[i] = gl_mul(x[i], s); x[stride + i] = gl_mul(x[stride + i], s); x[2 * stride + i] = gl_mul(x[2 * stride + i], s);
    }
}

// power tables of an arbitrary factor, built on the device (bfs_gl_scale: no host tables, no copies, no synchronisation)
__global__ void gl_power_tables_kernel(u64* lo, u64* hi, u32 lo_bits, u32 hi_bits, u64 factor) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (1u << lo_bits)) lo[i] = gl_pow(factor, i);
    if (i < (1u << hi_bits)) hi[i] = gl_pow(factor, (u64)i << lo_bits);
}

__global__ void gl_scale_kernel(const u64* in, u64* out, u64 n, u64 stride, const u64* s_lo, const u64* s_hi, u32 lo_bits) {
    const u64 b = blockIdx.y;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        out[b * stride + i] = gl_mul(in[b * stride + i], tw_pow(s_lo, s_hi, lo_bits, i));
}

static u32 grid_for(u64 n, u32 block) {
    u64 g = (n + block - 1) / block;
    return (u32)(g > 2048 ? 2048 : (g ? g : 1));
}

int mul_pointwise_launch(const u64* a, const u64* b, u64* out, u64 n, hipStream_t stream) {
    if (!n) return BFS_OK;
    hipLaunchKernelGGL(gl_mul_pointwise_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, a, b, out, n);
    BFS_HIP(hipGetLastError());
    return BFS_OK;
}

int batch_inverse_launch(const u64* in, u64* out, u64 n, hipStream_t stream) {
    if (!n) return BFS_OK;
    // the zero flag lives in pooled pinned host memory that the kernel writes directly: the reference's assert needs the answer
    // now, which costs one stream synchronisation but no copy command and no pinning of pageable memory
    void* h_flag = nullptr;
    void* d_flag = nullptr;
    BFS_TRY(host_alloc(64, &h_flag));
    *(volatile unsigned int*)h_flag = 0;
    if (hipHostGetDevicePointer(&d_flag, h_flag, 0) != hipSuccess) { (void)host_release(h_flag); set_error("hipHostGetDevicePointer failed"); return BFS_ERR_HIP; }
    const u64 per_block = (u64)BINV_T * BINV_E;
    hipLaunchKernelGGL(gl_batch_inverse_kernel, dim3((u32)((n + per_block - 1) / per_block)), dim3(BINV_T), 0, stream, in, out, n, (unsigned int*)d_flag);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    const unsigned int flag = *(volatile unsigned int*)h_flag;
    (void)host_release(h_flag);
    if (e != hipSuccess) { set_error("batch inverse: %s", hipGetErrorString(e)); return BFS_ERR_HIP; }
    if (flag) { set_error("batch inverse does not work when input contains a zero"); return BFS_ERR_ZERO_IN_BATCH_INVERSE; }
    return BFS_OK;
}

int xfe_mul_pointwise_launch(const u64* a, u64 a_stride, const u64* b, u64 b_stride, u64* out, u64 out_stride, u64 n, hipStream_t stream) {
    if (!n) return BFS_OK;
    hipLaunchKernelGGL(xfe_mul_pointwise_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, a, a_stride, b, b_stride, out, out_stride, n);
    BFS_HIP(hipGetLastError());
    return BFS_OK;
}

int xfe_batch_inverse_launch(const u64* in, u64 in_stride, u64* out, u64 out_stride, u64 n, hipStream_t stream) {
    if (!n) return BFS_OK;
    void* w = nullptr;
    BFS_TRY(workspace(8, n * sizeof(u64), stream, &w));
    u64* norm = (u64*)w;
    hipLaunchKernelGGL(xfe_cofactors_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, in, in_stride, out, out_stride, norm, n);
    BFS_HIP(hipGetLastError());
    // a zero element has norm zero: the base field's launch reports it (BFS_ERR_ZERO_IN_BATCH_INVERSE, ntt.py:178-179) and leaves
    // inverse(0) = 0 in its place, so the output of a zero is zero as in extension_field.py:80-83 (xgcd of the zero polynomial)
    const int rc = batch_inverse_launch(norm, norm, n, stream);
    if (rc != BFS_OK && rc != BFS_ERR_ZERO_IN_BATCH_INVERSE) return rc;
    hipLaunchKernelGGL(xfe_scale_by_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, out, out_stride, norm, n);
    BFS_HIP(hipGetLastError());
    return rc;
}

int scale_launch(const u64* in, u64* out, u64 n, u64 stride, u32 batch, u64 factor, hipStream_t stream) {
    if (!n || !batch) return BFS_OK;
    u32 log_n = 0;
    while ((1ull << log_n) < n) ++log_n;
    // two-level power tables of `factor` split at lo_bits (factor^i = lo[i & mask] * hi[i >> lo_bits]), not cached (arbitrary
    // factors would pile up) and built by a small kernel in stream-ordered workspace: nothing here touches the host
    const u32 lo_bits = (log_n + 1) / 2, hi_bits = log_n - lo_bits;
    void* w = nullptr;
    BFS_TRY(workspace(2, ((1ull << lo_bits) + (1ull << hi_bits)) * sizeof(u64), stream, &w));
    u64* d_lo = (u64*)w;
    u64* d_hi = d_lo + (1ull << lo_bits);
    const u32 entries = 1u << lo_bits;           // lo_bits >= hi_bits
    hipLaunchKernelGGL(gl_power_tables_kernel, dim3((entries + 255) / 256), dim3(256), 0, stream, d_lo, d_hi, lo_bits, hi_bits, factor);
    BFS_HIP(hipGetLastError());
    hipLaunchKernelGGL(gl_scale_kernel, dim3(grid_for(n, 256), batch), dim3(256), 0, stream, in, out, n, stride, d_lo, d_hi, lo_bits);
    BFS_HIP(hipGetLastError());
    return BFS_OK;
}

}  // namespace bfs
