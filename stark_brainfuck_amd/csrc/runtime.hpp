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

// upload a host table once and keep it for the lifetime of the process (keyed by caller-chosen 128-bit key)
int cached_table(uint64_t key_a, uint64_t key_b, uint64_t key_c, const u64* host, size_t count, const u64** d_out);
bool cached_table_lookup(uint64_t key_a, uint64_t key_b, uint64_t key_c, const u64** d_out);

// ---- internal entry points (device pointers, current device) ----
int ntt_launch(const u64* d_in, u64 n_in, u64 in_stride, u64* d_out, u64 out_stride, u32 log_n, u32 batch, u64 root,
               u64 shift, u64 post_scale, hipStream_t stream);

}  // namespace bfs
