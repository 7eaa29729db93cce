// Microbenchmarks that size the NTT design on MI355X (run through gpurun): integer VALU throughput for the modular
// multiply, and HBM access patterns of the tile passes.  Development tool, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../stark_brainfuck_amd/csrc/gl.hpp"
using namespace bfs;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int OP>
__global__ void valu_kernel(u64* out, u64 seed, int iters) {
    u64 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = seed + threadIdx.x * 977 + i * 7919 + blockIdx.x;
    u64 c = seed | 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) x[i] = (u64)(u32)x[i] * (u32)c + x[i];              // v_mad_u64_u32
            else if (OP == 1) x[i] = gl_mul(x[i], c);                         // full modular multiply
            else if (OP == 2) x[i] = gl_add(x[i], c);                         // modular add
            else if (OP == 3) x[i] = gl_sub(x[i], c);
            else if (OP == 4) x[i] = x[i] + c;                                // 64-bit add
            else if (OP == 5) { u32 lo = (u32)x[i] * (u32)c; u32 hi = __umulhi((u32)x[i], (u32)c); x[i] = ((u64)hi << 32) | lo; }
            else if (OP == 6) x[i] = gl_mul(x[i], 0x1000ULL);                 // multiply by 2^12 via the general path
            else if (OP == 7) { u32 a = (u32)x[i] & 0xFFFFFF, b = (u32)c & 0xFFFFFF; x[i] = x[i] + __umul24(a, b); }
        }
    }
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc ^= x[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int OP>
void run_valu(const char* name, u64* d_out) {
    const int blocks = 256 * 8, threads = 256, iters = 2000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(valu_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 12345ULL, 10);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(valu_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 12345ULL, iters);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double ops = (double)blocks * threads * iters * 8;
    double waveops = ops / 64;
    double cyc = ms * 1e-3 * 2.4e9 * 256 * 4 / waveops;   // SIMD-cycles per wave-level op at 2.4 GHz, 1024 SIMDs
    printf("%-28s %8.3f ms  %8.2f Gop/s  %6.2f SIMD-cycles per wave-op (if clock = 2.4 GHz)\n", name, ms, ops / ms / 1e6, cyc);
}

// ---- memory patterns
__global__ void copy16_kernel(const uint4* in, uint4* out, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void copy8_kernel(const u64* in, u64* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void read_kernel(const uint4* in, u64* out, size_t n16) {
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = in[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if (acc.x == 0x12345678 && acc.y == 1) out[0] = acc.z;
}
// the NTT column-pass pattern: tile = 256 rows x C columns (u64), row stride L elements, in place (read then write)
template <int C>
__global__ void tile_kernel(u64* data, size_t L, int rows) {
    // block handles rows x C tile; thread t: c = t % C, r0 = t / C; 16 elements per thread: r = r0 + (256/ (256/C))...
    const int tid = threadIdx.x;
    const int c = tid % C, rr = tid / C;           // rr in [0, 256/C)
    const size_t nl = L / C;
    const size_t h = blockIdx.x / nl, lch = blockIdx.x % nl;
    u64* base = data + h * ((size_t)rows * L) + lch * C + c;
    const int per = rows * C / 256;                // elements per thread
    u64 x[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) if (i < per) x[i] = base[(size_t)(rr + i * (256 / C)) * L];
#pragma unroll
    for (int i = 0; i < 64; ++i) if (i < per) base[(size_t)(rr + i * (256 / C)) * L] = x[i] + 1;
}

template <int C>
void run_tile(u64* d, size_t n, size_t L, int rows) {
    size_t blocks = n / ((size_t)rows * C);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(tile_kernel<C>, dim3(blocks), dim3(256), 0, 0, d, L, rows);
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(tile_kernel<C>, dim3(blocks), dim3(256), 0, 0, d, L, rows);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    printf("tile rows=%d C=%d (%d B segments) stride L=%zu: %.3f ms  %.1f GB/s (read+write)\n", rows, C, C * 8, L, ms, 16.0 * n / ms / 1e6);
}

int main() {
    u64* d_out; CK(hipMalloc(&d_out, 256 * 8 * 256 * 8));
    run_valu<0>("v_mad_u64_u32", d_out);
    run_valu<1>("gl_mul (full mulmod)", d_out);
    run_valu<2>("gl_add", d_out);
    run_valu<3>("gl_sub", d_out);
    run_valu<4>("u64 add", d_out);
    run_valu<5>("mul_lo + mul_hi u32", d_out);
    run_valu<6>("gl_mul by const 2^12", d_out);
    run_valu<7>("mul_u24 + add64", d_out);

    for (size_t mib : {64, 128, 256, 1024, 2048}) {
        size_t bytes = mib << 20;
        void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(copy16_kernel, dim3(2048), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16);
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(copy16_kernel, dim3(2048), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("copy16 %4zu MiB: %.3f ms  %.1f GB/s (r+w)\n", mib, ms, 2.0 * bytes / ms / 1e6);
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(copy8_kernel, dim3(2048), dim3(256), 0, 0, (const u64*)a, (u64*)b, bytes / 8);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("copy8  %4zu MiB: %.3f ms  %.1f GB/s (r+w)\n", mib, ms, 2.0 * bytes / ms / 1e6);
        // in-place re-read (does a buffer that was just written/read stay in the 256 MiB Infinity Cache?)
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(256), 0, 0, (const uint4*)b, (u64*)a, bytes / 16);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("read   %4zu MiB: %.3f ms  %.1f GB/s (repeated read of the same buffer)\n", mib, ms, 1.0 * bytes / ms / 1e6);
        CK(hipFree(a)); CK(hipFree(b));
    }
    {
        size_t n = (size_t)1 << 27;   // 1 GiB of u64: 8 columns x 2^24
        u64* d; CK(hipMalloc(&d, n * 8)); CK(hipMemset(d, 0, n * 8));
        run_tile<16>(d, n, (size_t)1 << 16, 256);   // pass 1 of 2^24: stride 2^16, 128 B segments
        run_tile<16>(d, n, (size_t)1 << 8, 256);    // pass 2: stride 2^8
        run_tile<8>(d, n, (size_t)1 << 16, 256);    // 64 B segments
        run_tile<4>(d, n, (size_t)1 << 16, 256);    // 32 B segments
        run_tile<32>(d, n, (size_t)1 << 16, 128);   // 256 B segments
        run_tile<4>(d, n, (size_t)1 << 12, 4096);   // 2-pass shape: 4096 rows x 4 cols (64 elements/thread)
        run_tile<8>(d, n, (size_t)1 << 12, 2048);   // 2048 rows x 8
        // single column (128 MiB) repeated: does the working set stay in the Infinity Cache between passes?
        run_tile<16>(d, (size_t)1 << 24, (size_t)1 << 16, 256);
        run_tile<16>(d, (size_t)1 << 24, (size_t)1 << 8, 256);
        CK(hipFree(d));
    }
    return 0;
}
