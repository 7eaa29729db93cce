// HBM roof of the NTT column pass' access shape (development tool): 8 columns x 2^24 u64, a workgroup of 256 threads moves a tile
// of 256 rows x 128 bytes (row stride 2^16 elements) from `src` to `dst` with no arithmetic, holding the LDS footprint of the real
// kernel (21.5 KiB -> 6 workgroups per CU).  Variants: 8 or 16 bytes per lane and access, ordinary or non-temporal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int W16, int NT, int LINEAR>
__global__ void __launch_bounds__(256) tile(const u64* src, u64* dst, u64* sink) {
    extern __shared__ u64 lds[];
    const unsigned tid = threadIdx.x;
    const size_t col = (size_t)blockIdx.y << 24;
    if (tid == 999) lds[tid] = 1;
    if (W16 == 0) {
        // lane: c = tid & 15 (column), o = tid >> 4; rows o + 16 d
        const unsigned c = tid & 15, o = tid >> 4;
        const size_t base = LINEAR ? col + (size_t)blockIdx.x * 4096 + tid : col + (size_t)blockIdx.x * 16 + c + ((size_t)o << 16);
        const size_t step = LINEAR ? 256 : (size_t)16 << 16;
        u64 x[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) x[d] = NT ? __builtin_nontemporal_load(src + base + d * step) : src[base + d * step];
#pragma unroll
        for (int d = 0; d < 16; ++d) { if (NT) __builtin_nontemporal_store(x[d] + 1, dst + base + d * step); else dst[base + d * step] = x[d] + 1; }
    } else {
        // lane: c2 = tid & 7 (pair of columns), o = tid >> 3 (32 row classes); rows o + 32 d, d < 8
        const unsigned c2 = tid & 7, o = tid >> 3;
        const size_t base = LINEAR ? col + (size_t)blockIdx.x * 4096 + 2 * tid : col + (size_t)blockIdx.x * 16 + 2 * c2 + ((size_t)o << 16);
        const size_t step = LINEAR ? 512 : (size_t)32 << 16;
        u64x2 x[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) x[d] = NT ? __builtin_nontemporal_load((const u64x2*)(src + base + d * step)) : *(const u64x2*)(src + base + d * step);
#pragma unroll
        for (int d = 0; d < 8; ++d) { u64x2 v = x[d]; v.x += 1; if (NT) __builtin_nontemporal_store(v, (u64x2*)(dst + base + d * step)); else *(u64x2*)(dst + base + d * step) = v; }
    }
    if (lds[0] == 0x1234567 && tid == 0) sink[0] = 1;
}

template <int W16, int NT, int LINEAR>
void run(const u64* s, u64* d, u64* sink, const char* tag) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t lds = 22016;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((tile<W16, NT, LINEAR>), dim3(4096, 8), dim3(256), lds, 0, s, d, sink);
    CK(hipEventRecord(a));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((tile<W16, NT, LINEAR>), dim3(4096, 8), dim3(256), lds, 0, s, d, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    printf("%-44s %7.1f us  %5.2f TB/s read+write\n", tag, ms * 1e3, 2.0 * (1ull << 30) / ms / 1e9);
}

int main() {
    u64 *s, *d, *sink;
    CK(hipMalloc(&s, 1ull << 30)); CK(hipMalloc(&d, 1ull << 30)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(s, 1, 1ull << 30)); CK(hipMemset(d, 0, 1ull << 30));
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0, 0>(s, d, sink, "tile 256 x 128 B,  8 B/lane, ordinary");
        run<0, 1, 0>(s, d, sink, "tile 256 x 128 B,  8 B/lane, non-temporal");
        run<1, 0, 0>(s, d, sink, "tile 256 x 128 B, 16 B/lane, ordinary");
        run<1, 1, 0>(s, d, sink, "tile 256 x 128 B, 16 B/lane, non-temporal");
        run<0, 0, 1>(s, d, sink, "linear,            8 B/lane, ordinary");
        run<0, 1, 1>(s, d, sink, "linear,            8 B/lane, non-temporal");
        run<1, 0, 1>(s, d, sink, "linear,           16 B/lane, ordinary");
        run<1, 1, 1>(s, d, sink, "linear,           16 B/lane, non-temporal");
    }
    return 0;
}
