// Is long straight-line code (every instruction executed once per wave, like the unrolled NTT tile kernel) limited by
// instruction fetch?  Same number of VALU operations executed (a) as a 4096-instruction straight line per "tile",
// (b) as a 64-instruction loop body iterated 64 times.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define OP4 a0 = a0 * 3 + b0; a1 = a1 * 5 + b1; a2 = (a2 ^ b2) + a0; a3 = (a3 + b3) ^ a1;
#define OP16 OP4 OP4 OP4 OP4
#define OP64 OP16 OP16 OP16 OP16
#define OP256 OP64 OP64 OP64 OP64
#define OP1024 OP256 OP256 OP256 OP256
#define OP4096 OP1024 OP1024 OP1024 OP1024

__global__ void straight(unsigned* out, int tiles) {
    unsigned a0 = threadIdx.x, a1 = blockIdx.x, a2 = 7, a3 = 9, b0 = 1, b1 = 2, b2 = 3, b3 = 4;
    for (int t = 0; t < tiles; ++t) { OP4096 asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}
__global__ void looped(unsigned* out, int tiles) {
    unsigned a0 = threadIdx.x, a1 = blockIdx.x, a2 = 7, a3 = 9, b0 = 1, b1 = 2, b2 = 3, b3 = 4;
    for (int t = 0; t < tiles * 64; ++t) { OP64 asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}
int main() {
    unsigned* d; CK(hipMalloc(&d, 4096 * 256 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int waves_per_simd : {1, 2, 4, 8}) {
        int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD
        for (int which = 0; which < 2; ++which) {
            int tiles = 32;
            if (which == 0) hipLaunchKernelGGL(straight, dim3(blocks), dim3(256), 0, 0, d, 1); else hipLaunchKernelGGL(looped, dim3(blocks), dim3(256), 0, 0, d, 1);
            CK(hipEventRecord(a));
            if (which == 0) hipLaunchKernelGGL(straight, dim3(blocks), dim3(256), 0, 0, d, tiles); else hipLaunchKernelGGL(looped, dim3(blocks), dim3(256), 0, 0, d, tiles);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            double instr_per_wave = 4096.0 * 1.5 * tiles;   // ~1.5 VALU instructions per C statement (mul-add + xor/add)
            double cyc = ms * 1e-3 * 2.1e9 / (instr_per_wave * waves_per_simd);
            printf("%-8s waves/SIMD %d: %.3f ms  ~%.2f cycles per VALU instruction per SIMD (2.1 GHz)\n", which ? "looped" : "straight", waves_per_simd, ms, cyc);
        }
    }
    return 0;
}
