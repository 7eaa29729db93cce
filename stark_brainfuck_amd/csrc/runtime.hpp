// runtime.hpp -- host-side plumbing shared by the translation units of libbfstark_hip.so:
// error reporting (bfs_last_error), HIP call checking, cached device tables and per-stream scratch space.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "ntt_plan.hpp"

namespace bfs {

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
const char* last_error();

#define BFS_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            ::bfs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return BFS_ERR_HIP;                                                                         \
        }                                                                                               \
    } while (0)

#define BFS_TRY(expr)            \
    do {                         \
        int rc_ = (expr);        \
        if (rc_) return rc_;     \
    } while (0)

// grow-only scratch buffer keyed by (device, stream, slot); contents are only valid within one API call
int workspace(int slot, size_t bytes, hipStream_t stream, void** out);
int workspace_release(int slot, hipStream_t stream);
bool workspace_peek(int slot, hipStream_t stream, void** ptr, size_t* bytes);   // the slot's buffer if it exists (no allocation)

// Pooled HBM and pinned-host memory.  hipMalloc / hipFree cost 50-500 us and hipFree synchronises the device, which at
// 288 GB of HBM is the wrong trade: freed blocks go back to a size-class free list (8 classes per octave above 1 MiB,
// powers of two below) and are handed out again without touching the driver.  Reuse is STREAM-ORDERED, like
// hipMallocAsync: a block released on stream S may be handed to work on S immediately (earlier work on S still reading
// it is ahead in the queue); a request from another stream -- or from the stream-less bfs_malloc -- synchronises S first.
// `NO_STREAM` as the release stream means "the device was idle when the block came back".
static const hipStream_t NO_STREAM = (hipStream_t)(uintptr_t)-1;
int device_alloc(size_t bytes, hipStream_t stream, void** out);
int device_release(void* ptr, hipStream_t stream);
int device_pool_trim();                                     // hipFree every cached block (synchronises the device)
void device_pool_stats(size_t* live_bytes, size_t* cached_bytes);
int host_alloc(size_t bytes, void** out);                   // pinned, pooled the same way (no stream semantics: callers copy synchronously)
int host_release(void* ptr);

// Blocking copies between host and HBM.  Large pageable host buffers are bounced through two pinned 8 MiB buffers from
// the host pool (the memcpy of chunk k+1 overlaps the DMA of chunk k) instead of letting the runtime pin the user pages:
// that path measured ~1 GB/s, and the unpinning afterwards stalls later kernel dispatches for ~25 ms.
int copy_h2d(void* d, const void* h, size_t bytes, hipStream_t stream);
int copy_d2h(void* h, const void* d, size_t bytes, hipStream_t stream);

// a stream is about to be destroyed: wait for it, free its scratch buffers, forget it in the pools
int stream_retire(hipStream_t stream);

// upload a host table once and keep it (keyed by caller-chosen values; the cache starts over after 4096 entries, runtime.cpp)
int cached_table(uint64_t key_a, uint64_t key_b, uint64_t key_c, const u64* host, size_t count, const u64** d_out);
bool cached_table_lookup(uint64_t key_a, uint64_t key_b, uint64_t key_c, const u64** d_out);

// ---- FRI rounds on small codewords, one launch (merkle.hip: fri_round_quad_kernel) ----
struct FriFoldArgs {
    const u64* in;             // previous round's codeword (null: this round's codeword is already at cw)
    u64 in_stride, half;       // half = length of this round's codeword
    Xfe alpha;                 // fri.py:120
    u64 scal;                  // 2^-1 * offset_r^-1
    const u64* winv_lo;        // two-level powers of the round-0 omega^-1
    const u64* winv_hi;
    u32 lo_bits, round_shift;
};
constexpr u64 FRI_FUSED_MAX = 16384;     // up to 256 workgroups of 64 leaves (their roots: one top kernel)
int merkle_build_xfe_fold_launch(const FriFoldArgs& fold, u64* d_cw, u64 cw_stride, u64 n, u64* d_nodes, hipStream_t stream, u64* root_out, u64 seq);
int fri_round_fused_launch(const FriFoldArgs& fold, u64* d_cw, u64 cw_stride, u64 n, u64* d_nodes, hipStream_t stream, u64* root_out, u64 seq);

// ---- internal entry points (device pointers, current device) ----
int ntt_route_probe_info(float* us, int* route, unsigned long long* probes);
int ntt_tune(const u64* d_in, u64 in_stride, u64* d_out, u64 out_stride, u32 log_n, u32 batch, u64 root, hipStream_t stream, int* route_out);
size_t ntt_route_forget_range(const void* lo, size_t bytes, bool may_free);
void ntt_route_trim();
int ntt_launch(const u64* d_in, u64 n_in, u64 in_stride, u64* d_out, u64 out_stride, u32 log_n, u32 batch, u64 root,
               u64 shift, u64 post_scale, hipStream_t stream);

// pinned, device-visible staging for one gather call: a lease on a block of the pooled pinned-host allocator (runtime.cpp:
// mutex-guarded, size classes, no hipHostMalloc / hipHostFree on the hot path).  One lease per call, so that concurrent provers
// -- threads, or several devices driven by one process -- never share a staging buffer; the lease goes back when the call returns.
struct PinnedLease {
    void* host = nullptr;
    void* dev = nullptr;
    int get(size_t need) {
        BFS_TRY(host_alloc(need < 4096 ? 4096 : need, &host));
        BFS_HIP(hipHostGetDevicePointer(&dev, host, 0));
        return BFS_OK;
    }
    ~PinnedLease() { if (host) (void)host_release(host); }
    PinnedLease() = default;
    PinnedLease(const PinnedLease&) = delete;
    PinnedLease& operator=(const PinnedLease&) = delete;
};

}  // namespace bfs
