// Microbenchmark: variants of the Merkle inner-level kernel (one BLAKE2b compression per parent) on MI355X.
// Development tool (run through gpurun), not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../stark_brainfuck_amd/csrc/merkle_core.hpp"
using namespace bfs;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// V0: as shipped -- thread t hashes children 2t, 2t+1 (16 x 8-byte loads at 128 B lane stride)
__global__ void v0(const u64* child, u64* parent, u64 count) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    u64 out[8];
    merkle_parent_body(child + 16 * t, child + 16 * t + 8, 2, out);
#pragma unroll
    for (int j = 0; j < 8; ++j) parent[8 * t + j] = out[j];
}
// V1: compute only
__global__ void v1(const u64* child, u64* parent, u64 count) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    u64 m[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) m[j] = t * 0x9E3779B97F4A7C15ULL + j;
    u64 h[8];
    blake2b_init(h);
    blake2b_compress(h, m, 128, true);
    if (h[0] == 0x1234567) parent[8 * t] = h[1] ^ h[2] ^ h[3] ^ h[4] ^ h[5] ^ h[6] ^ h[7];
}
// V2: 16-byte loads
__global__ void v2(const u64* child, u64* parent, u64 count) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const u64x2* src = (const u64x2*)(child + 16 * t);
    u64 m[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) { u64x2 v = src[j]; m[2 * j] = v.x; m[2 * j + 1] = v.y; }
    u64 h[8];
    blake2b_init(h);
    blake2b_compress(h, m, 128, true);
    u64x2* dst = (u64x2*)(parent + 8 * t);
#pragma unroll
    for (int j = 0; j < 4; ++j) { u64x2 v; v.x = h[2 * j]; v.y = h[2 * j + 1]; dst[j] = v; }
}
// V3: three levels per thread: 8 children -> 4 -> 2 -> 1, everything in registers; p1/p2/p3 are the three output levels
__global__ void v3(const u64* child, u64* p1, u64* p2, u64* p3, u64 count8) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count8) return;
    const u64x2* src = (const u64x2*)(child + 64 * t);
    u64 l1[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u64 m[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) { u64x2 v = src[8 * q + j]; m[2 * j] = v.x; m[2 * j + 1] = v.y; }
        blake2b_init(l1[q]);
        blake2b_compress(l1[q], m, 128, true);
        u64x2* dst = (u64x2*)(p1 + 8 * (4 * t + q));
#pragma unroll
        for (int j = 0; j < 4; ++j) { u64x2 v; v.x = l1[q][2 * j]; v.y = l1[q][2 * j + 1]; dst[j] = v; }
    }
    u64 l2[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        u64 m[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) { m[j] = l1[2 * q][j]; m[8 + j] = l1[2 * q + 1][j]; }
        blake2b_init(l2[q]);
        blake2b_compress(l2[q], m, 128, true);
        u64x2* dst = (u64x2*)(p2 + 8 * (2 * t + q));
#pragma unroll
        for (int j = 0; j < 4; ++j) { u64x2 v; v.x = l2[q][2 * j]; v.y = l2[q][2 * j + 1]; dst[j] = v; }
    }
    u64 m[16], h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m[j] = l2[0][j]; m[8 + j] = l2[1][j]; }
    blake2b_init(h);
    blake2b_compress(h, m, 128, true);
    u64x2* dst = (u64x2*)(p3 + 8 * t);
#pragma unroll
    for (int j = 0; j < 4; ++j) { u64x2 v; v.x = h[2 * j]; v.y = h[2 * j + 1]; dst[j] = v; }
}
// V4: the compression loop is not unrolled over q (smaller code), otherwise V3
__global__ void v4(const u64* child, u64* p1, u64* p2, u64* p3, u64 count8) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count8) return;
    const u64x2* src = (const u64x2*)(child + 64 * t);
    u64 l1[4][8];
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        u64 m[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) { u64x2 v = src[8 * q + j]; m[2 * j] = v.x; m[2 * j + 1] = v.y; }
        u64 h[8];
        blake2b_init(h);
        blake2b_compress(h, m, 128, true);
        u64x2* dst = (u64x2*)(p1 + 8 * (4 * t + q));
#pragma unroll
        for (int j = 0; j < 4; ++j) { u64x2 v; v.x = h[2 * j]; v.y = h[2 * j + 1]; dst[j] = v; }
        if (h[0] == 0x12345 && q == 5) p3[0] = 1;
    }
}

template <typename F>
void timeit(const char* name, F launch, double hashes) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    printf("%-44s %8.3f ms  %7.2f G hashes/s  (VALU floor at 2.2 GHz, 1983 instr/hash: %.3f ms)\n", name, ms, hashes / ms / 1e6,
           hashes * 1983 / 64 * 4 / 1024 / 2.2e9 * 1e3);
}

int main() {
    const u64 count = 1ull << 21;            // parents
    u64 *child, *p1, *p2, *p3;
    CK(hipMalloc(&child, count * 2 * 64)); CK(hipMalloc(&p1, count * 64)); CK(hipMalloc(&p2, count * 32)); CK(hipMalloc(&p3, count * 16));
    CK(hipMemset(child, 0x5A, count * 2 * 64));
    for (int bs : {64, 128, 256, 512}) {
        char nm[64];
        snprintf(nm, 64, "v0 shipped, block %d", bs);
        timeit(nm, [&] { hipLaunchKernelGGL(v0, dim3(count / bs), dim3(bs), 0, 0, child, p1, count); }, (double)count);
    }
    timeit("v1 compute only, block 256", [&] { hipLaunchKernelGGL(v1, dim3(count / 256), dim3(256), 0, 0, child, p1, count); }, (double)count);
    for (int bs : {64, 256})  {
        char nm[64];
        snprintf(nm, 64, "v2 16-byte loads/stores, block %d", bs);
        timeit(nm, [&] { hipLaunchKernelGGL(v2, dim3(count / bs), dim3(bs), 0, 0, child, p1, count); }, (double)count);
    }
    for (int bs : {64, 256}) {
        char nm[64];
        snprintf(nm, 64, "v3 three levels per thread, block %d", bs);
        timeit(nm, [&] { hipLaunchKernelGGL(v3, dim3(count / 4 / bs), dim3(bs), 0, 0, child, p1, p2, p3, count / 4); }, (double)count * 1.75);
    }
    timeit("v4 four parents per thread (loop), block 64", [&] { hipLaunchKernelGGL(v4, dim3(count / 4 / 64), dim3(64), 0, 0, child, p1, p2, p3, count / 4); }, (double)count);
    return 0;
}
