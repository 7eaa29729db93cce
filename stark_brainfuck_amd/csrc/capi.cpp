// capi.cpp -- extern "C" entry points of libbfstark_hip.so (declarations + reference citations: include/bfstark.h)
#include "../../include/bfstark.h"

#include "runtime.hpp"

#include <cstring>

namespace bfs {
int mul_pointwise_launch(const u64* a, const u64* b, u64* out, u64 n, hipStream_t stream);
int batch_inverse_launch(const u64* in, u64* out, u64 n, hipStream_t stream);
int scale_launch(const u64* in, u64* out, u64 n, u64 stride, u32 batch, u64 factor, hipStream_t stream);
int xfe_mul_pointwise_launch(const u64* a, u64 a_stride, const u64* b, u64 b_stride, u64* out, u64 out_stride, u64 n, hipStream_t stream);
int xfe_batch_inverse_launch(const u64* in, u64 in_stride, u64* out, u64 out_stride, u64 n, hipStream_t stream);
int merkle_build_xfe_launch(const u64* d_limbs, u64 limb_stride, u64 n, u64* d_nodes, hipStream_t stream, u64* root_out = nullptr, u64 seq = 0);
int merkle_build_bfe_launch(const u64* d_values, u64 n, u64* d_nodes, hipStream_t stream);
int merkle_build_bytes_launch(const u64* d_data, const u64* d_offsets, const u32* d_lengths, u64 n, u64* d_nodes, hipStream_t stream);
}

using namespace bfs;

extern "C" {

int bfs_version(void) { return 1; }
const char* bfs_last_error(void) { return last_error(); }

int bfs_device_count(int* count) { BFS_HIP(hipGetDeviceCount(count)); return BFS_OK; }
int bfs_set_device(int device) { BFS_HIP(hipSetDevice(device)); return BFS_OK; }
int bfs_malloc(void** d_ptr, size_t bytes) { return device_alloc(bytes, NO_STREAM, d_ptr); }
int bfs_free(void* d_ptr) {
    if (!d_ptr) return BFS_OK;
    BFS_HIP(hipDeviceSynchronize());             // the semantics of hipFree: nothing in flight touches the block afterwards
    return device_release(d_ptr, NO_STREAM);
}
int bfs_malloc_async(void** d_ptr, size_t bytes, void* stream) { return device_alloc(bytes, (hipStream_t)stream, d_ptr); }
int bfs_free_async(void* d_ptr, void* stream) { return device_release(d_ptr, (hipStream_t)stream); }
int bfs_pool_trim(void) { return device_pool_trim(); }
int bfs_pool_stats(size_t* live_bytes, size_t* cached_bytes) { device_pool_stats(live_bytes, cached_bytes); return BFS_OK; }
int bfs_host_alloc(void** h_ptr, size_t bytes) { return host_alloc(bytes, h_ptr); }
int bfs_host_free(void* h_ptr) { return host_release(h_ptr); }
int bfs_memcpy_h2d(void* d, const void* h, size_t bytes, void* stream) { return copy_h2d(d, h, bytes, (hipStream_t)stream); }
int bfs_memcpy_d2h(void* h, const void* d, size_t bytes, void* stream) { return copy_d2h(h, d, bytes, (hipStream_t)stream); }
int bfs_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
    BFS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return BFS_OK;
}
int bfs_memset(void* d, int value, size_t bytes, void* stream) {
    BFS_HIP(hipMemsetAsync(d, value, bytes, (hipStream_t)stream));
    return BFS_OK;
}
int bfs_stream_synchronize(void* stream) { BFS_HIP(hipStreamSynchronize((hipStream_t)stream)); return BFS_OK; }
int bfs_stream_create(void** stream) { hipStream_t st; BFS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); *stream = (void*)st; return BFS_OK; }
int bfs_stream_destroy(void* stream) {
    if (!stream) return BFS_OK;                  // the default stream is not ours to destroy
    BFS_TRY(stream_retire((hipStream_t)stream));
    BFS_HIP(hipStreamDestroy((hipStream_t)stream));
    return BFS_OK;
}
int bfs_event_create(void** event) { hipEvent_t e; BFS_HIP(hipEventCreate(&e)); *event = (void*)e; return BFS_OK; }
int bfs_event_destroy(void* event) { BFS_HIP(hipEventDestroy((hipEvent_t)event)); return BFS_OK; }
int bfs_event_record(void* event, void* stream) { BFS_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream)); return BFS_OK; }
int bfs_event_elapsed_ms(void* start, void* stop, float* ms) {
    BFS_HIP(hipEventSynchronize((hipEvent_t)stop));
    BFS_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return BFS_OK;
}

uint64_t bfs_gl_primitive_root(uint32_t log_n) { return gl_primitive_root(log_n); }
uint64_t bfs_gl_mul(uint64_t a, uint64_t b) { return gl_mul(a % GL_P, b % GL_P); }
uint64_t bfs_gl_inv(uint64_t a) { return gl_inv(a % GL_P); }
uint64_t bfs_gl_pow(uint64_t a, uint64_t e) { return gl_pow(a % GL_P, e); }

int bfs_gl_ntt(const uint64_t* d_in, uint64_t n_in, uint64_t in_stride, uint64_t* d_out, uint64_t out_stride, uint32_t log_n,
               uint32_t batch, uint64_t root, uint64_t coset_shift, uint64_t post_scale, void* stream) {
    return ntt_launch(d_in, n_in, in_stride, d_out, out_stride, log_n, batch, root, coset_shift, post_scale, (hipStream_t)stream);
}

int bfs_ntt_route_probe_info(float* us, int* route, unsigned long long* probes) { return ntt_route_probe_info(us, route, probes); }

int bfs_ntt_tune(const uint64_t* d_in, uint64_t in_stride, uint64_t* d_out, uint64_t out_stride, uint32_t log_n, uint32_t batch, uint64_t root,
                 void* stream, int* route) {
    return ntt_tune(d_in, in_stride, d_out, out_stride, log_n, batch, root, (hipStream_t)stream, route);
}

int bfs_ntt_route_forget(const void* d_ptr, size_t* forgotten) {
    BFS_HIP(hipDeviceSynchronize());             // candidate buffers nobody is routed through any more are freed: nothing may be in flight
    const size_t gone = ntt_route_forget_range(d_ptr, 0, true);
    if (forgotten) *forgotten = gone;
    return BFS_OK;
}

int bfs_gl_scale(const uint64_t* d_in, uint64_t* d_out, uint64_t n, uint64_t stride, uint32_t batch, uint64_t factor, void* stream) {
    return scale_launch(d_in, d_out, n, stride, batch, factor, (hipStream_t)stream);
}

int bfs_gl_mul_pointwise(const uint64_t* d_a, const uint64_t* d_b, uint64_t* d_out, uint64_t n, void* stream) {
    return mul_pointwise_launch(d_a, d_b, d_out, n, (hipStream_t)stream);
}

int bfs_gl_batch_inverse(const uint64_t* d_in, uint64_t* d_out, uint64_t n, void* stream) {
    return batch_inverse_launch(d_in, d_out, n, (hipStream_t)stream);
}

int bfs_xfe_mul_pointwise(const uint64_t* d_a, uint64_t a_stride, const uint64_t* d_b, uint64_t b_stride, uint64_t* d_out, uint64_t out_stride,
                          uint64_t n, void* stream) {
    return xfe_mul_pointwise_launch(d_a, a_stride, d_b, b_stride, d_out, out_stride, n, (hipStream_t)stream);
}

int bfs_xfe_batch_inverse(const uint64_t* d_in, uint64_t in_stride, uint64_t* d_out, uint64_t out_stride, uint64_t n, void* stream) {
    return xfe_batch_inverse_launch(d_in, in_stride, d_out, out_stride, n, (hipStream_t)stream);
}

static int check_nodes(const void* d_nodes) {
    if (((uintptr_t)d_nodes & 15) != 0) {
        set_error("d_nodes must be 16-byte aligned");
        return BFS_ERR_BAD_ARG;
    }
    return BFS_OK;
}

int bfs_merkle_build_xfe(const uint64_t* d_limbs, uint64_t limb_stride, uint64_t n, uint8_t* d_nodes, void* stream) {
    BFS_TRY(check_nodes(d_nodes));
    return merkle_build_xfe_launch(d_limbs, limb_stride, n, (u64*)d_nodes, (hipStream_t)stream);
}
int bfs_merkle_build_bfe(const uint64_t* d_values, uint64_t n, uint8_t* d_nodes, void* stream) {
    BFS_TRY(check_nodes(d_nodes));
    return merkle_build_bfe_launch(d_values, n, (u64*)d_nodes, (hipStream_t)stream);
}
int bfs_merkle_build_bytes(const uint8_t* d_data, const uint64_t* d_word_offsets, const uint32_t* d_lengths, uint64_t n, uint8_t* d_nodes, void* stream) {
    BFS_TRY(check_nodes(d_nodes));
    return merkle_build_bytes_launch((const u64*)d_data, d_word_offsets, d_lengths, n, (u64*)d_nodes, (hipStream_t)stream);
}
int bfs_merkle_open(const uint8_t* d_nodes, uint32_t depth, uint64_t index, uint8_t* h_path, void* stream) {
    // merkle.py:46-52: walk from the leaf to the root collecting siblings; one small copy per level
    uint64_t k = (1ull << depth) | index;
    for (uint32_t lvl = 0; k > 1; k >>= 1, ++lvl)
        BFS_HIP(hipMemcpyAsync(h_path + 64 * (size_t)lvl, d_nodes + 64 * (k ^ 1), 64, hipMemcpyDeviceToHost, (hipStream_t)stream));
    BFS_HIP(hipStreamSynchronize((hipStream_t)stream));
    return BFS_OK;
}

}  // extern "C"
