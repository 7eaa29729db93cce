// Device-vs-host check of the field primitives and of compositions of them (constants included) on MI355X.
// Exists because hipcc (ROCm 7.2) miscompiled gl_add(gl_sub(a, b), 1): a plain add of a carry was folded into the next
// add-with-carry and the merged carry-out used.  Development tool (run through gpurun); prints the number of mismatches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../stark_brainfuck_amd/csrc/gl.hpp"
using namespace bfs;
#define NOPS 24
__host__ __device__ inline void ops(u64 a, u64 b, u64* o) {
    const u64 d = gl_sub(a, b), s = gl_add(a, b), m = gl_mul(a, b);
    o[0] = s; o[1] = d; o[2] = m; o[3] = gl_neg(a);
    o[4] = gl_add(a, 1ULL); o[5] = gl_sub(a, 1ULL); o[6] = gl_sub(a, 2ULL); o[7] = gl_sub(d, 2ULL);
    o[8] = gl_add(d, 1ULL); o[9] = gl_mul(a, gl_sub(gl_sub(b, a), 2ULL));
    o[10] = gl_add(s, 1ULL); o[11] = gl_add(m, 1ULL); o[12] = gl_sub(m, 1ULL); o[13] = gl_add(d, GL_P - 1);
    o[14] = gl_add(gl_add(d, s), m); o[15] = gl_sub(gl_add(m, 7ULL), d); o[16] = gl_mul(gl_add(d, 1ULL), gl_sub(s, 1ULL));
    o[17] = gl_add(gl_neg(d), 1ULL); o[18] = gl_sub(1ULL, d); o[19] = gl_sub(0ULL, m); o[20] = gl_add(1ULL, gl_mul(d, 44ULL));
    o[21] = gl_mul_lazy(gl_mul_lazy(a, b), gl_add(d, 1ULL)) % GL_P; o[22] = gl_add(gl_sub(gl_sub(a, b), b), 1ULL); o[23] = gl_sub(gl_add(gl_add(a, 1ULL), 1ULL), b);
}
__global__ void k(const u64* in, u64* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ops(in[2 * i], in[2 * i + 1], out + NOPS * i);
}
static u64 rnd(u64& s) { s += 0x9E3779B97F4A7C15ULL; u64 z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return (z ^ (z >> 31)) % GL_P; }
int main() {
    const int n = 1 << 18;
    static u64 h[2 * n], got[NOPS * n];
    u64 s = 3;
    const u64 edges[] = {0, 1, 2, 3, 0xFFFFFFFFULL, 0x100000000ULL, 0x100000001ULL, GL_P - 1, GL_P - 2, GL_P - 3, 0xFFFFFFFF00000000ULL - 1, 0xFFFFFFFEFFFFFFFFULL, 300, 44, 0xFFFFFFFE00000001ULL, 0x8000000000000000ULL};
    const int ne = sizeof(edges) / sizeof(edges[0]);
    for (int i = 0; i < n; ++i) {
        if (i < ne * ne) { h[2 * i] = edges[i / ne]; h[2 * i + 1] = edges[i % ne]; }
        else { h[2 * i] = (i & 64) ? rnd(s) % 300 : ((i & 128) ? GL_P - 1 - rnd(s) % 300 : rnd(s)); h[2 * i + 1] = (i & 32) ? rnd(s) % 300 : rnd(s); }
    }
    u64 *d, *dout;
    if (hipMalloc(&d, sizeof(h)) != hipSuccess || hipMalloc(&dout, sizeof(got)) != hipSuccess) { printf("no device\n"); return 2; }
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, dout, n);
    (void)hipMemcpy(got, dout, sizeof(got), hipMemcpyDeviceToHost);
    int total = 0;
    for (int j = 0; j < NOPS; ++j) {
        int bad = 0;
        for (int i = 0; i < n; ++i) {
            u64 w[NOPS];
            ops(h[2 * i], h[2 * i + 1], w);
            if (w[j] != got[NOPS * i + j]) { if (bad < 1) printf("op %d a=%llx b=%llx got %llx want %llx\n", j, (unsigned long long)h[2 * i], (unsigned long long)h[2 * i + 1], (unsigned long long)got[NOPS * i + j], (unsigned long long)w[j]); ++bad; }
        }
        total += bad;
    }
    printf("mismatches: %d\n", total);
    return total != 0;
}
