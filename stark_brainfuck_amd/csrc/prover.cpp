// prover.cpp -- BrainfuckStark.prove's orchestration between its Fiat-Shamir points, natively (bfs_stark_commit / bfs_stark_finish).
//
// The reference's prove() (/root/reference/code/brainfuck_stark.py:134-341) is a straight line of stages; stark_brainfuck_amd/
// brainfuck_stark.py mirrors it in Python on top of the kernels of this library, and for a small proof ("Hello World!", FRI domain
// 2^17) the interpreter between the ~110 launches was 2/3 of the 3 ms (round-4 verdict: 1 / 2 / 4 prover THREADS gave 291 / 281 / 307
// proofs/s while 8 prover processes gave 1 056).  Here the same stages are driven from C++ -- padding (processor_table.py:24-35,
// instruction_table.py:19-25, memory_table.py:40-44, io_table.py:17-21), interpolation + low-degree extension (table.py:112-148), the
// zipped commitments (brainfuck_stark.py:178-179, 197-198), the table extensions (processor_table.py:329-427, instruction_table.py:
// 167-231, memory_table.py:172-206, io_table.py:77-110), the non-linear combination (brainfuck_stark.py:236-300), openings (:315-333) and
// FRI (:336) -- through the library's own entry points (include/bfstark.h), in two calls:
//
//   bfs_stark_commit   randomizer codeword, padding, base LDE, base commitment, challenges, table extension, terminals; the extension
//                      columns' LDE is queued and the call returns while the GPU runs it
//   (host language)    what depends on OBJECT identity in the reference and on its symbolic degree bookkeeping stays with the caller:
//                      the five terminal objects (processor_table.py:390-404: which BaseFieldElement object an evaluation terminal is
//                      made of) and the quotient degree bounds (multivariate.py:144-170) -- computed while the GPU works
//   bfs_stark_finish   extension commitment, terminals into the transcript, weights, combination, its tree, indices, openings, FRI
//
// Nothing here computes a field element that the stages above do not already compute; the proof bytes are those of the Python
// prover (tests: both paths on the reference's ten golden proofs and on random programs).
#include "../../include/bfstark.h"

#include "blake2b.hpp"
#include "runtime.hpp"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace bfs;

namespace {

constexpr int NT = 5;                                       // processor, instruction, memory, input, output (brainfuck_stark.py:56-60)
constexpr u32 BASE_W[NT] = {7, 3, 4, 1, 1};
constexpr u32 FULL_W[NT] = {11, 5, 5, 2, 2};
constexpr u32 NUM_RAND[NT] = {1, 1, 1, 0, 0};               // brainfuck_stark.py:48: one randomizer per column; the IO tables have none

// The coset transform of all tables' coefficient columns (table.py:138-149).  Small domains: ONE call over every column with the largest
// table's coefficient count (a small proof is a chain of launches; merging them was round 5's gain).  Large domains (>= 2^20 points): one
// call per run of tables with the same count -- a column's zero padding is what the transform's cost depends on (ntt_plan.hpp: a table
// with 2^16 + 1 coefficients on 2^22 points takes the two-pass expansion plan, one with 2^17 + 1 the three-pass plan), and a table must not
// pay for its neighbour's height.  Columns beyond a table's own count are zero in `coeffs` either way: same values.
constexpr u32 LDE_GROUP_MIN_LOG_N = 20;
constexpr int NUM_SCANS = 9;

inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct DeviceBlock {
    void* ptr = nullptr;
    hipStream_t stream = nullptr;
    int get(size_t bytes, hipStream_t s) {
        release();
        stream = s;
        return device_alloc(bytes ? bytes : 8, s, &ptr);
    }
    void release() { if (ptr) { (void)device_release(ptr, stream); ptr = nullptr; } }
    u64* words() const { return (u64*)ptr; }
    ~DeviceBlock() { release(); }
    DeviceBlock() = default;
    DeviceBlock(const DeviceBlock&) = delete;
    DeviceBlock& operator=(const DeviceBlock&) = delete;
};

struct PinnedBlock {
    void* ptr = nullptr;
    int get(size_t bytes) { release(); return host_alloc(bytes ? bytes : 8, &ptr); }
    void release() { if (ptr) { (void)host_release(ptr); ptr = nullptr; } }
    ~PinnedBlock() { release(); }
    PinnedBlock() = default;
    PinnedBlock(const PinnedBlock&) = delete;
    PinnedBlock& operator=(const PinnedBlock&) = delete;
};

struct StarkSession {
    bfs_stark_params P{};
    u64 n = 0;
    u64 height[NT] = {0}, length[NT] = {0}, omicron[NT] = {0};
    u64 base_at[NT] = {0}, ext_at[NT] = {0};                // first column of the table inside the shared codeword buffers
    u32 total_base = 0, total_ext = 0;                      // base columns; extension columns (elements, three planes each)
    DeviceBlock randomizer_cw, trace[NT], masks, ext_trace[NT], terminals_dev, coeffs, base_cw, ext_cw, base_nodes, ext_nodes, base_salts_dev,
        ext_salts_dev, combination, comb_nodes, zerofiers;
    std::vector<uint8_t> base_salts_host, ext_salts_host;   // explicit salts (a test replaced urandom): opened salts are read from here
    bool base_salts_on_device = false, ext_salts_on_device = false;
    u64 challenges[33] = {0};
    u64 ext_moduli[16] = {0};
    bfs_stark_randomness R{};                               // (pointers are only valid during bfs_stark_commit; values are copied below)
    std::vector<u64> ext_randomizers;
    std::vector<uint8_t> ext_salts_in;
    uint8_t ext_salt_seed[32];
    bool have_ext_salt_seed = false;
    bool committed = false;
    hipStream_t stream = nullptr;
    // The commitment to the zipped extension rows is the first thing bfs_stark_finish needs and depends on nothing the caller does
    // between the two calls, so bfs_stark_commit hands it to a thread of its own: the GPU hashes rows while the caller's interpreter
    // makes terminal objects and degree bounds.  Joined by bfs_stark_finish (or by whatever ends the session first).
    // The thread is the session's own and lives as long as the session (a thread per proof cost its creation on the critical path of
    // every small proof): it sleeps on a condition variable between proofs.
    std::thread ext_tree_thread;
    std::mutex ext_mu;
    std::condition_variable ext_cv;
    std::function<void()> ext_job;          // posted by bfs_stark_commit, taken by the thread
    bool ext_busy = false, ext_quit = false;
    int ext_tree_rc = BFS_OK;
    std::string ext_tree_error;
    uint8_t ext_root[64];
    void post_ext_tree(std::function<void()> job) {
        std::unique_lock<std::mutex> lock(ext_mu);
        ext_cv.wait(lock, [this] { return !ext_busy; });
        ext_job = std::move(job);
        ext_busy = true;
        if (!ext_tree_thread.joinable())
            ext_tree_thread = std::thread([this] {
                std::unique_lock<std::mutex> l(ext_mu);
                for (;;) {
                    ext_cv.wait(l, [this] { return ext_quit || ext_job; });
                    if (ext_quit) return;
                    std::function<void()> job = std::move(ext_job);
                    ext_job = nullptr;
                    l.unlock();
                    job();
                    l.lock();
                    ext_busy = false;
                    ext_cv.notify_all();
                }
            });
        ext_cv.notify_all();
    }
    void join_ext_tree() {                  // (waits for the posted commitment; the thread itself stays)
        std::unique_lock<std::mutex> lock(ext_mu);
        ext_cv.wait(lock, [this] { return !ext_busy; });
    }
    void stop_ext_thread() {
        {
            std::unique_lock<std::mutex> lock(ext_mu);
            ext_cv.wait(lock, [this] { return !ext_busy; });
            ext_quit = true;
        }
        ext_cv.notify_all();
        if (ext_tree_thread.joinable()) ext_tree_thread.join();
    }
    // Side streams.  At the sizes where this driver matters the kernels of a proof are small (10-20 us, a handful of workgroups) and
    // queue faster than they run, so independent chains -- the randomizer's sampling + transform, each table's interpolation +
    // randomizer correction -- run side by side instead of one after the other: fork / join with events around them.
    static constexpr int AUX = 3;
    hipStream_t aux[AUX] = {nullptr, nullptr, nullptr};
    hipEvent_t fork_ev = nullptr, join_ev[AUX] = {nullptr, nullptr, nullptr}, rand_ev = nullptr;
    int aux_device = -1;
    int ensure_streams() {
        int dev = 0;
        BFS_HIP(hipGetDevice(&dev));
        if (aux[0] && dev == aux_device) return BFS_OK;
        drop_streams();
        aux_device = dev;
        for (int k = 0; k < AUX; ++k) {
            BFS_HIP(hipStreamCreateWithFlags(&aux[k], hipStreamNonBlocking));
            BFS_HIP(hipEventCreateWithFlags(&join_ev[k], hipEventDisableTiming));
        }
        BFS_HIP(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
        BFS_HIP(hipEventCreateWithFlags(&rand_ev, hipEventDisableTiming));
        return BFS_OK;
    }
    void drop_streams() {
        for (int k = 0; k < AUX; ++k) {
            if (aux[k]) { (void)stream_retire(aux[k]); (void)hipStreamDestroy(aux[k]); aux[k] = nullptr; }
            if (join_ev[k]) { (void)hipEventDestroy(join_ev[k]); join_ev[k] = nullptr; }
        }
        if (fork_ev) { (void)hipEventDestroy(fork_ev); fork_ev = nullptr; }
        if (rand_ev) { (void)hipEventDestroy(rand_ev); rand_ev = nullptr; }
    }
    ~StarkSession() { stop_ext_thread(); drop_streams(); }
};

u64 padding_length(u64 rows) {                              // table.py: rows to add so that the count becomes a power of two (0 and 2^k stay)
    if ((rows & (rows - 1)) == 0) return 0;
    u64 p = 1;
    while (p < rows) p <<= 1;
    return p - rows;
}

u32 log2_exact(u64 v) { u32 l = 0; while ((1ull << l) < v) ++l; return l; }

Xfe xfe_pow(Xfe a, u64 e) {
    Xfe acc{{1, 0, 0}};
    while (e) {
        if (e & 1) acc = xfe_mul(acc, a);
        a = xfe_mul(a, a);
        e >>= 1;
    }
    return acc;
}

int check_session(void* s, const char* who) {
    if (!s) { set_error("%s: null session", who); return BFS_ERR_BAD_ARG; }
    return BFS_OK;
}

}  // namespace

extern "C" {

void* bfs_stark_session_new(void) { return new StarkSession(); }
void bfs_stark_session_free(void* s) { delete (StarkSession*)s; }

static int lde_all_tables(const StarkSession& S, const u64* coeffs, u64 stride, u64* codewords, u64 n, u32 log_n, u64 omega, u64 offset, u32 planes,
                          bool extension, hipStream_t stream) {
    const u32 total = planes * (extension ? S.total_ext : S.total_base);
    if (total == 0) return BFS_OK;
    if (log_n < LDE_GROUP_MIN_LOG_N) return bfs_gl_ntt(coeffs, stride, stride, codewords, n, log_n, total, omega, offset, 1, stream);
    for (int t = 0; t < NT;) {
        const u64 count = S.height[t] + NUM_RAND[t];                       // a table's interpolant: height + randomizers coefficients (table.py:112-136)
        int u = t;
        u32 columns = 0;
        while (u < NT && S.height[u] + NUM_RAND[u] == count) {
            columns += planes * (extension ? FULL_W[u] - BASE_W[u] : BASE_W[u]);
            ++u;
        }
        if (columns != 0) {
            const u64 first = planes * (extension ? S.ext_at[t] : S.base_at[t]);
            const u64 n_in = count == 0 ? 1 : (count < stride ? count : stride);   // (an empty table's columns are zero: one zero coefficient each)
            BFS_TRY(bfs_gl_ntt(coeffs + first * stride, n_in, stride, codewords + first * n, n, log_n, columns, omega, offset, 1, stream));
        }
        t = u;
    }
    return BFS_OK;
}

int bfs_stark_commit(void* session, void* ps, const bfs_stark_params* params, const bfs_stark_table_in* tables, const bfs_stark_randomness* rnd,
                     uint64_t* out_challenges, uint64_t* out_scan_terminals, uint64_t* out_io_terminals, double* out_ms, void* stream_) {
    BFS_TRY(check_session(session, "bfs_stark_commit"));
    StarkSession& S = *(StarkSession*)session;
    hipStream_t stream = (hipStream_t)stream_;
    const double t0 = now_ms();
    static const bool trace_marks = [] { const char* e = getenv("BFS_STARK_MARKS"); return e && e[0] == '1'; }();
    double t_last = t0;
    auto mark = [&](const char* what) { if (trace_marks) { const double t = now_ms(); fprintf(stderr, "[bfs mark] %-28s %.3f ms\n", what, t - t_last); t_last = t; } };
    S.join_ext_tree();                                     // (a commit that was never finished)
    S.P = *params;
    S.stream = stream;
    S.committed = false;
    const bfs_stark_params& P = S.P;
    if (P.log_n < 2 || P.log_n > 32) { set_error("bfs_stark_commit: log_n"); return BFS_ERR_BAD_ARG; }
    const u64 n = S.n = 1ull << P.log_n;
    const u64 offset = P.offset, omega = P.omega;
    if (P.max_degree + 1 > n) { set_error("bfs_stark_commit: max_degree does not fit the domain"); return BFS_ERR_BAD_ARG; }
    // ---- shapes.  Tables 0..2 are as tall as their constructor said (table.py:25: roundup_npo2(length)); the IO tables take their
    // height from the symbols they hold (io_table.py:17-21)
    u64 hmax = 0;
    S.total_base = S.total_ext = 0;
    for (int t = 0; t < NT; ++t) {
        const u64 rows = tables[t].rows;
        const u64 h = rows + padding_length(rows);
        if (t < 3 && h != P.heights[t]) {
            set_error("bfs_stark_commit: table %d has %llu rows (padded %llu), its constructor said height %llu", t, (unsigned long long)rows,
                      (unsigned long long)h, (unsigned long long)P.heights[t]);
            return BFS_ERR_BAD_ARG;
        }
        if (rows && (tables[t].values == nullptr || tables[t].row_stride < BASE_W[t])) { set_error("bfs_stark_commit: table %d matrix", t); return BFS_ERR_BAD_ARG; }
        S.height[t] = h;
        S.length[t] = rows;
        S.omicron[t] = h >= 2 ? bfs_gl_primitive_root(log2_exact(h)) : 1;          // table.py:41-46 derive_omicron
        S.base_at[t] = S.total_base; S.total_base += BASE_W[t];
        S.ext_at[t] = S.total_ext;   S.total_ext += FULL_W[t] - BASE_W[t];
        if (h > hmax) hmax = h;
    }
    if (hmax == 0 || hmax + 1 > n) { set_error("bfs_stark_commit: interpolant does not fit the FRI domain"); return BFS_ERR_BAD_ARG; }
    const u64 stride = hmax + 1;

    // ---- randomizer polynomial and codeword (brainfuck_stark.py:162-167), queued first: the GPU transforms while the host pads
    BFS_TRY(S.ensure_streams());
    // an error between a fork and its join leaves work on the side streams that the pool's stream-ordered reuse knows nothing about:
    // every early return waits for the device before the session's blocks go back
    // The stage's own blocks are DECLARED IN FRONT of the guard so that they are destroyed after it: on an early return the guard's
    // device synchronisation comes first and only then do the pinned staging block and the device blocks go back to their pools (declared
    // behind the guard -- as rpoly, stage, raw and padded were -- they were released while the copy out of `stage` and work forked onto the
    // side streams could still be in flight, and another prover thread could be handed the pinned block mid-copy: round-5 advice).
    DeviceBlock rpoly, raw, padded;
    PinnedBlock stage;
    struct Guard { bool ok = false; ~Guard() { if (!ok) (void)hipDeviceSynchronize(); } } guard;
    const u64 count = P.max_degree + 1;
    BFS_TRY(rpoly.get(3 * count * 8, stream));
    BFS_TRY(S.randomizer_cw.get(3 * n * 8, stream));
    const u64 salt_words = (3 * n + 7) / 8 * 8;
    if (rnd->base_salt_seed) BFS_TRY(S.base_salts_dev.get(salt_words * 8, stream));
    if (rnd->ext_salt_seed) BFS_TRY(S.ext_salts_dev.get(salt_words * 8, stream));
    BFS_HIP(hipEventRecord(S.fork_ev, stream));             // (the blocks above may have been released on `stream` by work still queued there)
    hipStream_t rs = S.aux[2];
    BFS_HIP(hipStreamWaitEvent(rs, S.fork_ev, 0));
    if (rnd->randomizer_seed) {
        BFS_TRY(bfs_xfe_sample_fill(rnd->randomizer_seed, rpoly.words(), count, count, rs));
    } else if (rnd->randomizer_limbs) {
        BFS_TRY(bfs_memcpy_h2d(rpoly.ptr, rnd->randomizer_limbs, 3 * count * 8, rs));
    } else { set_error("bfs_stark_commit: no randomizer polynomial"); return BFS_ERR_BAD_ARG; }
    BFS_TRY(bfs_gl_ntt(rpoly.words(), count, count, S.randomizer_cw.words(), n, P.log_n, 3, omega, offset, 1, rs));
    // the salts of both commitments are expanded from their seeds here too, next to the transforms: nothing depends on them until the leaf kernels
    if (rnd->base_salt_seed) BFS_TRY(bfs_random_fill(rnd->base_salt_seed, S.base_salts_dev.words(), salt_words, rs));
    if (rnd->ext_salt_seed) BFS_TRY(bfs_random_fill(rnd->ext_salt_seed, S.ext_salts_dev.words(), salt_words, rs));
    BFS_HIP(hipEventRecord(S.rand_ev, rs));                 // joined in front of the base commitment, which reads the codeword (and the salts)
    mark("randomizer + salts queued");

    // ---- the rows as the virtual machine wrote them go up in one copy; padding, transposition and the scan masks happen on the device
    // (bfs_trace_pad: the host loops cost 0.3-0.6 ms of a 14 ms proof with the GPU waiting)
    u64 trace_words = 0, mask_bytes = 0, raw_words = 0;
    for (int t = 0; t < NT; ++t) { trace_words += (u64)BASE_W[t] * S.height[t]; raw_words += tables[t].rows * tables[t].row_stride; }
    // masks: processor active / reads / writes, instruction product / evaluation rows, memory non-dummy rows
    const u64 hp = S.height[0], hi = S.height[1], hm = S.height[2];
    mask_bytes = 3 * hp + 2 * hi + hm;
    BFS_TRY(stage.get(raw_words * 8));
    BFS_TRY(raw.get(raw_words * 8, stream));
    BFS_TRY(padded.get(trace_words * 8 + ((mask_bytes + 7) & ~7ull), stream));
    {
        u64 at = 0;
        for (int t = 0; t < NT; ++t) {
            const u64 words = tables[t].rows * tables[t].row_stride;
            if (words) memcpy((u64*)stage.ptr + at, tables[t].values, words * 8);
            at += words;
        }
        if (raw_words) BFS_HIP(hipMemcpyAsync(raw.ptr, stage.ptr, raw_words * 8, hipMemcpyHostToDevice, stream));
        uint8_t* mk = (uint8_t*)(padded.words() + trace_words);
        bfs_trace_pad_table pt[NT];
        u64 raw_at = 0, out_at = 0;
        for (int t = 0; t < NT; ++t) {
            pt[t] = bfs_trace_pad_table{raw.words() + raw_at, tables[t].rows, tables[t].row_stride, S.height[t], padded.words() + out_at, nullptr, nullptr, nullptr, t, BASE_W[t]};
            raw_at += tables[t].rows * tables[t].row_stride;
            out_at += (u64)BASE_W[t] * S.height[t];
        }
        pt[0].d_mask0 = mk; pt[0].d_mask1 = mk + hp; pt[0].d_mask2 = mk + 2 * hp;
        pt[1].d_mask0 = mk + 3 * hp; pt[1].d_mask1 = mk + 3 * hp + hi;
        pt[2].d_mask0 = mk + 3 * hp + 2 * hi;
        BFS_TRY(bfs_trace_pad(pt, NT, stream));
    }
    // (the padded block becomes the session's trace storage: tables point into it)
    S.trace[0].release();
    S.trace[0].ptr = padded.ptr; S.trace[0].stream = stream; padded.ptr = nullptr;
    u64* d_trace[NT];
    {
        u64 at = 0;
        for (int t = 0; t < NT; ++t) { d_trace[t] = S.trace[0].words() + at; at += (u64)BASE_W[t] * S.height[t]; }
    }
    const uint8_t* d_masks = (const uint8_t*)(S.trace[0].words() + trace_words);
    mark("rows copied, padding queued");
    const double t_pad = now_ms();

    // ---- base LDE (table.py:112-148 for every table; one coset transform for all columns)
    BFS_TRY(S.coeffs.get((u64)std::max(S.total_base, 3 * S.total_ext) * stride * 8, stream));
    BFS_TRY(bfs_memset(S.coeffs.ptr, 0, (u64)S.total_base * stride * 8, stream));
    BFS_TRY(S.base_cw.get((u64)S.total_base * n * 8, stream));
    {
        // every table's chain (inverse transform over its own subgroup, randomizer correction) on a stream of its own: tables 0 and 1
        // on the side streams, the rest on the caller's
        const u64* rv = rnd->base_randomizers;
        BFS_HIP(hipEventRecord(S.fork_ev, stream));
        bool forked[2] = {false, false};
        for (int t = 0; t < NT; ++t) {
            const u64 h = S.height[t];
            const u32 w = BASE_W[t];
            if (!h) continue;
            hipStream_t st = t < 2 ? S.aux[t] : stream;
            if (t < 2) { BFS_HIP(hipStreamWaitEvent(st, S.fork_ev, 0)); forked[t] = true; }
            u64* mine = S.coeffs.words() + S.base_at[t] * stride;
            BFS_TRY(bfs_gl_ntt(d_trace[t], h, h, mine, stride, log2_exact(h), w, bfs_gl_inv(S.omicron[t]), 1, bfs_gl_inv(h % GL_P), st));
            if (NUM_RAND[t]) {
                if (!rv) { set_error("bfs_stark_commit: base randomizers missing"); return BFS_ERR_BAD_ARG; }
                BFS_TRY(bfs_poly_randomize(mine, stride, h, w, omega, rv, st));
                rv += w;
            }
            if (t < 2) BFS_HIP(hipEventRecord(S.join_ev[t], st));
        }
        for (int t = 0; t < 2; ++t) if (forked[t]) BFS_HIP(hipStreamWaitEvent(stream, S.join_ev[t], 0));
    }
    BFS_TRY(lde_all_tables(S, S.coeffs.words(), stride, S.base_cw.words(), n, P.log_n, omega, offset, /*planes=*/1, /*extension=*/false, stream));
    BFS_HIP(hipStreamWaitEvent(stream, S.rand_ev, 0));      // the randomizer codeword is ready from here on
    rpoly.release();                                        // (stream-ordered behind the join: its transform has run)
    const double t_lde = now_ms();

    // ---- commitment to the zipped base rows (brainfuck_stark.py:178-179): randomizer codeword first, then every base column
    bfs_row_column cols[32];
    u32 nc = 0;
    cols[nc++] = bfs_row_column{S.randomizer_cw.words(), 1, 0};
    for (u32 c = 0; c < S.total_base; ++c) cols[nc++] = bfs_row_column{S.base_cw.words() + (u64)c * n, 0, 0};
    BFS_TRY(S.base_nodes.get(2 * n * 64, stream));
    uint8_t root[64];
    if (rnd->base_salt_seed) {
        S.base_salts_on_device = true;                      // (filled on the randomizer's stream, joined above)
        BFS_TRY(bfs_merkle_build_rows_root(cols, nc, n, n, (const uint8_t*)S.base_salts_dev.ptr, 1, (uint8_t*)S.base_nodes.ptr, root, stream));
    } else if (rnd->base_salts) {
        S.base_salts_host.assign(rnd->base_salts, rnd->base_salts + 24 * n);
        S.base_salts_on_device = false;
        BFS_TRY(bfs_merkle_build_rows_root(cols, nc, n, n, S.base_salts_host.data(), 0, (uint8_t*)S.base_nodes.ptr, root, stream));
    } else { set_error("bfs_stark_commit: no salts for the base commitment"); return BFS_ERR_BAD_ARG; }
    // push(root) + prover_fiat_shamir(): the eleven challenges (brainfuck_stark.py:181-183)
    uint8_t seed[32];
    BFS_TRY(bfs_ps_push_digest_fiat_shamir(ps, root, seed, 32));
    BFS_TRY(bfs_sample_weights(seed, 32, 11, S.challenges));
    memcpy(out_challenges, S.challenges, sizeof S.challenges);
    const double t_tree = now_ms();

    // ---- table extension (Table.extend of every table) as prefix scans on the trace columns in HBM
    const u64* ch = S.challenges;
    auto C = [&](int i) { return ch + 3 * i; };             // a b c d e f alpha beta gamma delta eta = 0..10
    const u64 one[3] = {1, 0, 0}, zero[3] = {0, 0, 0};
    u64 ext_rows_total = 0;
    for (int t = 0; t < NT; ++t) ext_rows_total += 3ull * (FULL_W[t] - BASE_W[t]) * S.height[t];
    S.ext_trace[0].release();
    BFS_TRY(S.ext_trace[0].get(ext_rows_total * 8, stream));
    u64* d_ext[NT];
    {
        u64 at = 0;
        for (int t = 0; t < NT; ++t) { d_ext[t] = S.ext_trace[0].words() + at; at += 3ull * (FULL_W[t] - BASE_W[t]) * S.height[t]; }
    }
    BFS_TRY(S.terminals_dev.get(3 * NUM_SCANS * 8, stream));
    bfs_scan_spec specs[NUM_SCANS];
    u32 ns = 0;
    int slot_of[NUM_SCANS];
    auto add = [&](int slot, int t, int k, int kind, int before, int c1, int c2, int c3, u64 shift1, const uint8_t* mask, const u64* k0, const u64* k1,
                   const u64* k2, const u64* k3, const u64* initial) {
        const u64 h = S.height[t];
        if (!h) { memcpy(out_scan_terminals + 3 * slot, initial, 24); return; }            // no rows: the terminal is the initial value
        bfs_scan_spec& sp = specs[ns];
        memset(&sp, 0, sizeof sp);
        sp.kind = kind; sp.record_before = before;
        sp.d_x1 = c1 >= 0 ? d_trace[t] + (u64)c1 * h : nullptr;
        sp.d_x2 = c2 >= 0 ? d_trace[t] + (u64)c2 * h : nullptr;
        sp.d_x3 = c3 >= 0 ? d_trace[t] + (u64)c3 * h : nullptr;
        sp.shift1 = shift1; sp.d_mask = mask; sp.n = h;
        const u64* ks[4] = {k0, k1, k2, k3};
        for (int i = 0; i < 4; ++i) if (ks[i]) memcpy(sp.constants + 3 * i, ks[i], 24);
        memcpy(sp.initial, initial, 24);
        sp.d_out = d_ext[t] + 3ull * k * h; sp.out_stride = h;
        sp.d_terminal = S.terminals_dev.words() + 3 * slot;
        slot_of[ns++] = slot;
    };
    const u64* init0 = rnd->initials;                       // brainfuck_stark.py:184-185: one initial per permutation argument
    const u64* init1 = rnd->initials + 3;
    // processor (processor_table.py:329-427): two running products over the active rows, the input / output evaluations
    add(0, 0, 0, 0, 1, 1, 2, 3, 0, d_masks, C(6), C(0), C(1), C(2), init0);
    add(1, 0, 1, 0, 1, 0, 4, 5, 0, d_masks, C(7), C(3), C(4), C(5), init1);
    add(2, 0, 2, 1, 1, 5, -1, -1, 1, d_masks + hp, C(8), one, nullptr, nullptr, zero);      // an input symbol shows up in the NEXT row's memory value
    add(3, 0, 3, 1, 1, 5, -1, -1, 0, d_masks + 2 * hp, C(9), one, nullptr, nullptr, zero);
    // instruction (instruction_table.py:167-231): recorded AFTER the row's update
    add(4, 1, 0, 0, 0, 0, 1, 2, 0, d_masks + 3 * hp, C(6), C(0), C(1), C(2), init0);
    add(5, 1, 1, 1, 0, 0, 1, 2, 0, d_masks + 3 * hp + hi, C(10), C(0), C(1), C(2), zero);
    // memory (memory_table.py:172-206)
    add(6, 2, 0, 0, 1, 0, 1, 2, 0, d_masks + 3 * hp + 2 * hi, C(7), C(3), C(4), C(5), init1);
    // input / output (io_table.py:77-110): evaluation = evaluation * iota + symbol on every row
    add(7, 3, 0, 1, 0, 0, -1, -1, 0, nullptr, C(8), one, nullptr, nullptr, zero);
    add(8, 4, 0, 1, 0, 0, -1, -1, 0, nullptr, C(9), one, nullptr, nullptr, zero);
    if (ns) BFS_TRY(bfs_xfe_scan_device_many(specs, ns, stream));
    // one read-back: the final states, and the IO tables' value after their last REAL row (io_table.py:106-110)
    {
        bfs_gather_request req[3];
        u32 nr = 0;
        u64 got[3 * NUM_SCANS + 6];
        req[nr++] = bfs_gather_request{S.terminals_dev.words(), 3 * NUM_SCANS, 1, 0};
        int io_req[2] = {-1, -1};
        for (int k = 0; k < 2; ++k) {
            const int t = 3 + k;
            if (S.length[t]) { io_req[k] = (int)nr; req[nr++] = bfs_gather_request{d_ext[t] + (S.length[t] - 1), 3, (u32)S.height[t], 0}; }
        }
        BFS_TRY(bfs_gather(req, nr, got, stream));
        for (u32 i = 0; i < ns; ++i) memcpy(out_scan_terminals + 3 * slot_of[i], got + 3 * slot_of[i], 24);
        u64 at = 3 * NUM_SCANS;
        for (int k = 0; k < 2; ++k) {
            if (io_req[k] >= 0) { memcpy(out_io_terminals + 3 * k, got + at, 24); at += 3; }
            else memset(out_io_terminals + 3 * k, 0, 24);
        }
    }
    const double t_ext = now_ms();

    // ---- extension columns: interpolation, randomizers, support summary (read back), coset transform QUEUED -- the call returns while
    // the GPU runs it and the caller prepares terminal objects and degree bounds
    BFS_TRY(bfs_memset(S.coeffs.ptr, 0, 3ull * S.total_ext * stride * 8, stream));
    BFS_TRY(S.ext_cw.get(3ull * S.total_ext * n * 8, stream));
    {
        const u64* rv = rnd->ext_randomizers;
        BFS_HIP(hipEventRecord(S.fork_ev, stream));
        bool forked[2] = {false, false};
        for (int t = 0; t < NT; ++t) {
            const u64 h = S.height[t];
            const u32 w = 3 * (FULL_W[t] - BASE_W[t]);
            if (!h) continue;
            hipStream_t st = t < 2 ? S.aux[t] : stream;
            if (t < 2) { BFS_HIP(hipStreamWaitEvent(st, S.fork_ev, 0)); forked[t] = true; }
            u64* mine = S.coeffs.words() + 3 * S.ext_at[t] * stride;
            BFS_TRY(bfs_gl_ntt(d_ext[t], h, h, mine, stride, log2_exact(h), w, bfs_gl_inv(S.omicron[t]), 1, bfs_gl_inv(h % GL_P), st));
            if (NUM_RAND[t]) {
                if (!rv) { set_error("bfs_stark_commit: extension randomizers missing"); return BFS_ERR_BAD_ARG; }
                BFS_TRY(bfs_poly_randomize(mine, stride, h, w, omega, rv, st));
                rv += w;
            }
            if (t < 2) BFS_HIP(hipEventRecord(S.join_ev[t], st));
        }
        for (int t = 0; t < 2; ++t) if (forked[t]) BFS_HIP(hipStreamWaitEvent(stream, S.join_ev[t], 0));
    }
    {
        // Table.ext_sharing_moduli (stark_brainfuck_amd/table.py): which codeword elements of a column hold the same coefficient objects
        // in the reference (univariate.py:23-27 inside the recursive ntt), from the support of the column's interpolant
        u64 masks[3 * 16];
        BFS_TRY(bfs_poly_support(S.coeffs.words(), stride, stride, 3 * S.total_ext, masks, stream));
        for (int t = 0; t < NT; ++t)
            for (u32 c = 0; c < FULL_W[t] - BASE_W[t]; ++c) {
                const u32 col = (u32)S.ext_at[t] + c;
                u64 modulus = 0;
                if (S.height[t]) {
                    const u64 m = masks[3 * col] | masks[3 * col + 1] | masks[3 * col + 2];
                    const u64 low = m & ((1ull << 63) - 1);
                    if (m == 0) modulus = 0;
                    else if (low == 0) modulus = 1;
                    else { const u32 v = (u32)__builtin_ctzll(low); modulus = v ? n >> v : 0; }
                }
                S.ext_moduli[col] = modulus;
            }
    }
    BFS_TRY(lde_all_tables(S, S.coeffs.words(), stride, S.ext_cw.words(), n, P.log_n, omega, offset, /*planes=*/3, /*extension=*/true, stream));
    // what bfs_stark_finish needs of the caller's randomness
    S.have_ext_salt_seed = rnd->ext_salt_seed != nullptr;
    if (rnd->ext_salt_seed) memcpy(S.ext_salt_seed, rnd->ext_salt_seed, 32);
    else if (rnd->ext_salts) S.ext_salts_host.assign(rnd->ext_salts, rnd->ext_salts + 24 * n);
    else { set_error("bfs_stark_commit: no salts for the extension commitment"); return BFS_ERR_BAD_ARG; }
    // ---- commitment to the zipped extension rows (brainfuck_stark.py:197-198), on a thread of its own (see StarkSession)
    BFS_TRY(S.ext_nodes.get(2 * n * 64, stream));
    if (S.have_ext_salt_seed) {
        S.ext_salts_on_device = true;                       // (filled at the start of the call, on the randomizer's stream)
    } else {
        S.ext_salts_on_device = false;
    }
    {
        int dev = 0;
        BFS_HIP(hipGetDevice(&dev));
        StarkSession* sp = &S;
        S.ext_tree_rc = BFS_OK;
        S.post_ext_tree([sp, dev, n, stream] {
            StarkSession& T = *sp;
            if (hipSetDevice(dev) != hipSuccess) { T.ext_tree_rc = BFS_ERR_HIP; T.ext_tree_error = "hipSetDevice failed on the commitment thread"; return; }
            bfs_row_column cols[32];
            u32 nc = 0;
            for (u32 c = 0; c < T.total_ext; ++c) cols[nc++] = bfs_row_column{T.ext_cw.words() + 3ull * c * n, 1, 0};
            const uint8_t* salts = T.ext_salts_on_device ? (const uint8_t*)T.ext_salts_dev.ptr : T.ext_salts_host.data();
            T.ext_tree_rc = bfs_merkle_build_rows_root(cols, nc, n, n, salts, T.ext_salts_on_device ? 1 : 0, (uint8_t*)T.ext_nodes.ptr, T.ext_root, stream);
            if (T.ext_tree_rc != BFS_OK) T.ext_tree_error = bfs_last_error();
        });
    }
    S.committed = true;
    guard.ok = true;
    if (out_ms) {
        const double t_end = now_ms();
        out_ms[0] = t_pad - t0; out_ms[1] = t_lde - t_pad; out_ms[2] = t_tree - t_lde; out_ms[3] = t_ext - t_tree; out_ms[4] = t_end - t_ext;
    }
    return BFS_OK;
}

int bfs_stark_finish(void* session, void* ps, const uint64_t* terminal_handles, const uint64_t* terminals, const uint64_t* shifts,
                     uint32_t num_terms, int32_t base_field_id, const uint64_t* distances, uint32_t n_distances, uint64_t* out_indices,
                     uint8_t* out_weights_seed, uint64_t* out_fri_indices, double* out_ms, void* stream_) {
    BFS_TRY(check_session(session, "bfs_stark_finish"));
    StarkSession& S = *(StarkSession*)session;
    hipStream_t stream = (hipStream_t)stream_;
    if (!S.committed || stream != S.stream) { set_error("bfs_stark_finish: call bfs_stark_commit on the same stream first"); return BFS_ERR_BAD_ARG; }
    S.committed = false;
    const double t0 = now_ms();
    const bfs_stark_params& P = S.P;
    const u64 n = S.n, offset = P.offset, omega = P.omega;
    // ---- the commitment to the zipped extension rows was started by bfs_stark_commit; its root, then the terminals (:197-224)
    S.join_ext_tree();
    if (S.ext_tree_rc != BFS_OK) { set_error("%s", S.ext_tree_error.c_str()); return S.ext_tree_rc; }
    const uint8_t* root = S.ext_root;
    {
        const uint64_t h = bfs_ps_obj_bytes(ps, root, 64);
        BFS_TRY(bfs_ps_push(ps, h));
    }
    for (int k = 0; k < 5; ++k) BFS_TRY(bfs_ps_push(ps, terminal_handles[k]));
    const double t_tree = now_ms();

    // ---- weights of the non-linear combination (brainfuck_stark.py:226-243) and the combination itself (:245-298), quotients folded in
    const u32 num_base = S.total_base, num_ext = S.total_ext;
    u32 num_quot = 0;
    int nq[NT];
    for (int t = 0; t < NT; ++t) { nq[t] = bfs_air_num_quotients(t); num_quot += (u32)nq[t]; }
    num_quot += 2;                                          // the two permutation arguments (brainfuck_stark.py:62-65)
    if (num_terms != num_base + num_ext + num_quot) {
        set_error("bfs_stark_finish: %u shifts for %u terms", num_terms, num_base + num_ext + num_quot);
        return BFS_ERR_BAD_ARG;
    }
    uint8_t wseed[32];
    BFS_TRY(bfs_ps_fiat_shamir(ps, (size_t)-1, wseed, 32));
    if (out_weights_seed) memcpy(out_weights_seed, wseed, 32);
    std::vector<u64> weights(3ull * (1 + 2 * num_terms));
    BFS_TRY(bfs_sample_weights(wseed, 32, 1 + 2 * num_terms, weights.data()));
    std::vector<bfs_comb_weight> terms(num_terms);
    for (u32 s = 0; s < num_terms; ++s) {
        memcpy(terms[s].wa, weights.data() + 3 * (1 + 2 * s), 24);
        memcpy(terms[s].wb, weights.data() + 3 * (2 + 2 * s), 24);
        if (shifts[s] >> 32) { set_error("bfs_stark_finish: shift of term %u does not fit 32 bits", s); return BFS_ERR_BAD_ARG; }
        terms[s].shift = shifts[s];
    }
    // every distinct zerofier denominator of the proof, inverted together (stark_brainfuck_amd/table.py: zerofier_inverses)
    u32 z_is_power[12];
    u64 z_value[12];
    u32 nz = 0;
    auto spec_index = [&](u32 is_power, u64 value) {
        for (u32 k = 0; k < nz; ++k) if (z_is_power[k] == is_power && z_value[k] == value) return k;
        z_is_power[nz] = is_power; z_value[nz] = value;
        return nz++;
    };
    spec_index(0, 1);
    u32 z_omi[NT], z_pow[NT];
    for (int t = 0; t < NT; ++t) {
        z_omi[t] = spec_index(0, bfs_gl_inv(S.omicron[t]));
        z_pow[t] = S.height[t] ? spec_index(1, log2_exact(S.height[t])) : 0;
    }
    BFS_TRY(S.zerofiers.get((u64)nz * n * 8, stream));
    BFS_TRY(bfs_zerofier_inverses(P.log_n, offset, omega, nz, z_is_power, z_value, S.zerofiers.words(), stream));
    BFS_TRY(S.combination.get(3 * n * 8, stream));
    u32 quot_at = num_base + num_ext;
    std::vector<bfs_comb_weight> mine;
    for (int t = 0; t < NT; ++t) {
        const u32 bw = BASE_W[t], xw = FULL_W[t] - BASE_W[t];
        mine.clear();
        mine.insert(mine.end(), terms.begin() + S.base_at[t], terms.begin() + S.base_at[t] + bw);
        mine.insert(mine.end(), terms.begin() + num_base + S.ext_at[t], terms.begin() + num_base + S.ext_at[t] + xw);
        mine.insert(mine.end(), terms.begin() + quot_at, terms.begin() + quot_at + nq[t]);
        quot_at += (u32)nq[t];
        const u64* inv[3] = {S.zerofiers.words(), S.zerofiers.words() + (u64)z_omi[t] * n, S.height[t] ? S.zerofiers.words() + (u64)z_pow[t] * n : nullptr};
        u64 params[3];
        const u64* pr = nullptr;
        if (t >= 3) {                                       // io_table.py:58-60: iota^(height - length)
            const u64* iota = S.challenges + 3 * (t == 3 ? 8 : 9);
            const Xfe v = xfe_pow(Xfe{{iota[0], iota[1], iota[2]}}, S.height[t] - S.length[t]);
            params[0] = v.c[0]; params[1] = v.c[1]; params[2] = v.c[2];
            pr = params;
        }
        BFS_TRY(bfs_air_combine(t, S.base_cw.words() + S.base_at[t] * n, S.ext_cw.words() + 3 * S.ext_at[t] * n, P.log_n,
                                S.height[t] ? n / S.height[t] : 0, S.height[t], bfs_gl_inv(S.omicron[t]), offset, omega, S.challenges, terminals, pr,
                                mine.data(), t == 0 ? S.randomizer_cw.words() : nullptr, t == 0 ? weights.data() : nullptr, S.combination.words(),
                                inv, stream));
    }
    // permutation arguments (brainfuck_stark.py:62-65): processor's instruction permutation against the instruction table's, processor's
    // memory permutation against the memory table's -- extension columns 0 / 1 of table 0, 0 of tables 1 and 2
    {
        const u64* lhs0 = S.ext_cw.words() + 3 * (S.ext_at[0] + 0) * n;
        const u64* lhs1 = S.ext_cw.words() + 3 * (S.ext_at[0] + 1) * n;
        const u64* rhs0 = S.ext_cw.words() + 3 * (S.ext_at[1] + 0) * n;
        const u64* rhs1 = S.ext_cw.words() + 3 * (S.ext_at[2] + 0) * n;
        BFS_TRY(bfs_difference_combine(lhs0, rhs0, P.log_n, offset, omega, &terms[quot_at], S.combination.words(), S.zerofiers.words(), stream));
        BFS_TRY(bfs_difference_combine(lhs1, rhs1, P.log_n, offset, omega, &terms[quot_at + 1], S.combination.words(), S.zerofiers.words(), stream));
    }
    const double t_comb = now_ms();

    // ---- commitment to the combination codeword (:300-301), indices (:303-304)
    BFS_TRY(S.comb_nodes.get(2 * n * 64, stream));
    BFS_TRY(bfs_merkle_build_xfe(S.combination.words(), n, n, (uint8_t*)S.comb_nodes.ptr, stream));
    uint8_t comb_root[64];
    BFS_HIP(hipMemcpyAsync(comb_root, (const uint8_t*)S.comb_nodes.ptr + 64, 64, hipMemcpyDeviceToHost, stream));
    BFS_HIP(hipStreamSynchronize(stream));
    uint8_t iseed[32];
    BFS_TRY(bfs_ps_push_digest_fiat_shamir(ps, comb_root, iseed, 32));
    const u32 num_indices = P.security_level;
    std::vector<u64> indices(num_indices);
    {
        // brainfuck_stark.py:114-123: int.from_bytes(blake2b(randomness + bytes(i)).digest(), "big") % n  (bytes(i) = i zero bytes)
        std::vector<unsigned char> msg(iseed, iseed + 32);
        for (u32 i = 0; i < num_indices; ++i) {
            unsigned char digest[64];
            blake2b_host(msg.data(), msg.size(), digest);
            msg.push_back(0);
            u64 low = 0;
            for (int b = 56; b < 64; ++b) low = (low << 8) | digest[b];
            indices[i] = low & (n - 1);                     // n is a power of two: the residue is the low bits
        }
    }
    if (out_indices) memcpy(out_indices, indices.data(), num_indices * 8);
    const double t_ctree = now_ms();

    // ---- openings (:315-333)
    bfs_gather_request base_req[1 + NT], ext_req[NT];
    u32 nb = 0, ne = 0;
    base_req[nb++] = bfs_gather_request{S.randomizer_cw.words(), 3, (u32)n, 0};
    for (int t = 0; t < NT; ++t) base_req[nb++] = bfs_gather_request{S.base_cw.words() + S.base_at[t] * n, BASE_W[t], (u32)n, 0};
    for (int t = 0; t < NT; ++t) ext_req[ne++] = bfs_gather_request{S.ext_cw.words() + 3 * S.ext_at[t] * n, 3 * (FULL_W[t] - BASE_W[t]), (u32)n, 0};
    std::vector<u64> leaf_handles(num_indices);
    BFS_TRY(bfs_stark_push_openings(ps, base_req, nb, base_field_id, ext_req, ne, S.ext_moduli, S.total_ext, n, (const uint8_t*)S.base_nodes.ptr,
                                    S.base_salts_on_device ? (const uint8_t*)S.base_salts_dev.ptr : S.base_salts_host.data(), S.base_salts_on_device ? 1 : 0,
                                    (const uint8_t*)S.ext_nodes.ptr,
                                    S.ext_salts_on_device ? (const uint8_t*)S.ext_salts_dev.ptr : S.ext_salts_host.data(), S.ext_salts_on_device ? 1 : 0,
                                    S.combination.words(), n, (const uint8_t*)S.comb_nodes.ptr, indices.data(), num_indices, distances, n_distances,
                                    leaf_handles.data(), stream));
    const double t_open = now_ms();

    // ---- FRI on the combination codeword (:335-336): round 0 is the tree that was just built
    void* fri = bfs_fri_session_new();
    int rc = bfs_fri_session_round0_tree(fri, (const uint8_t*)S.comb_nodes.ptr, comb_root);
    if (rc == BFS_OK) rc = bfs_fri_commit(fri, ps, S.combination.words(), n, P.log_n, offset, omega, P.expansion_factor, stream);
    for (u32 a = 0; rc == BFS_OK && a < num_indices; ++a) rc = bfs_fri_session_alias(fri, ps, 0, indices[a], leaf_handles[a]);
    std::vector<u64> top(P.num_colinearity_checks ? P.num_colinearity_checks : 1);
    if (rc == BFS_OK) rc = bfs_fri_query(fri, ps, P.num_colinearity_checks, top.data(), stream);
    bfs_fri_session_free(fri);
    if (rc != BFS_OK) return rc;
    if (out_fri_indices) memcpy(out_fri_indices, top.data(), P.num_colinearity_checks * 8);
    // everything goes back to the pool (stream-ordered: nothing queued still reads it after the query's synchronisation)
    for (DeviceBlock* b : {&S.randomizer_cw, &S.trace[0], &S.ext_trace[0], &S.terminals_dev, &S.coeffs, &S.base_cw, &S.ext_cw, &S.base_nodes, &S.ext_nodes,
                           &S.base_salts_dev, &S.ext_salts_dev, &S.combination, &S.comb_nodes, &S.zerofiers})
        b->release();
    if (out_ms) {
        const double t_end = now_ms();
        out_ms[0] = t_tree - t0; out_ms[1] = t_comb - t_tree; out_ms[2] = t_ctree - t_comb; out_ms[3] = t_open - t_ctree; out_ms[4] = t_end - t_open;
    }
    return BFS_OK;
}

}  // extern "C"
