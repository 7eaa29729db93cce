// HBM access-pattern microbenchmark for the NTT tile passes (development tool).
// Pattern: the array is a matrix of `rows x L` u64; a workgroup owns all `rows` rows x SEG bytes of columns and
// reads / writes it in 16-byte pieces per lane.  Reports GB/s for read-only, write-only and read+write(in place).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0 read, 1 write, 2 read+write in place.  NT: nontemporal loads/stores
template <int SEG16 /* 16-byte units per row segment */, int ROWS, int THREADS, int MODE, int NT>
__global__ void __launch_bounds__(THREADS) tile16(uint4* data, size_t L16 /* row length in 16-byte units */, u64* sink) {
    constexpr int ITEMS = SEG16 * ROWS / THREADS;     // 16-byte items per thread
    const int tid = threadIdx.x;
    const int c = tid % SEG16, r0 = tid / SEG16;
    constexpr int RSTEP = THREADS / SEG16;
    const size_t nl = L16 / SEG16;
    const size_t h = blockIdx.x / nl, lch = blockIdx.x % nl;
    uint4* base = data + h * ((size_t)ROWS * L16) + lch * SEG16 + c;
    uint4 x[ITEMS];
    if (MODE != 1) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint4* p = base + (size_t)(r0 + i * RSTEP) * L16;
            if (NT) { v4u t = __builtin_nontemporal_load((const v4u*)p); x[i] = make_uint4(t.x, t.y, t.z, t.w); } else x[i] = *p;
        }
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) x[i] = make_uint4(tid, i, 3, 4);
    }
    if (MODE != 0) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            uint4 v = x[i]; v.x += 1;
            uint4* p = base + (size_t)(r0 + i * RSTEP) * L16;
            if (NT) { v4u t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, (v4u*)p); } else *p = v;
        }
    } else {
        unsigned acc = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) acc ^= x[i].x ^ x[i].y ^ x[i].z ^ x[i].w;
        if (acc == 0x12345679u) sink[0] = acc;
    }
}

template <int SEG16, int ROWS, int THREADS, int MODE, int NT>
float run(uint4* d, size_t bytes, size_t L16, u64* sink, int reps = 5) {
    size_t blocks = bytes / 16 / ((size_t)SEG16 * ROWS);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((tile16<SEG16, ROWS, THREADS, MODE, NT>), dim3(blocks), dim3(THREADS), 0, 0, d, L16, sink);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((tile16<SEG16, ROWS, THREADS, MODE, NT>), dim3(blocks), dim3(THREADS), 0, 0, d, L16, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

template <int SEG16, int ROWS, int THREADS>
void suite(uint4* d, size_t bytes, size_t L16, u64* sink, const char* tag) {
    float r = run<SEG16, ROWS, THREADS, 0, 0>(d, bytes, L16, sink), w = run<SEG16, ROWS, THREADS, 1, 0>(d, bytes, L16, sink),
          rw = run<SEG16, ROWS, THREADS, 2, 0>(d, bytes, L16, sink), rwnt = run<SEG16, ROWS, THREADS, 2, 1>(d, bytes, L16, sink);
    printf("%-10s seg %4d B rows %4d thr %4d tile %3d KB: read %6.0f  write %6.0f  r+w %6.0f  r+w(nt) %6.0f GB/s\n", tag, SEG16 * 16, ROWS, THREADS,
           SEG16 * 16 * ROWS / 1024, bytes / r / 1e6, bytes / w / 1e6, 2.0 * bytes / rw / 1e6, 2.0 * bytes / rwnt / 1e6);
}

int main() {
    u64* sink; CK(hipMalloc(&sink, 64));
    for (size_t mib : {1024, 128}) {
        size_t bytes = mib << 20;
        uint4* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 0, bytes));
        printf("---- array %zu MiB\n", mib);
        const size_t n16 = bytes / 16;
        // pass-1 like: rows far apart (L = total / rows)
        suite<8, 256, 256>(d, bytes, n16 / 256 / (mib == 1024 ? 8 : 1), sink, "far");
        suite<16, 256, 256>(d, bytes, n16 / 256 / (mib == 1024 ? 8 : 1), sink, "far");
        suite<16, 256, 1024>(d, bytes, n16 / 256 / (mib == 1024 ? 8 : 1), sink, "far");
        suite<32, 256, 1024>(d, bytes, n16 / 256 / (mib == 1024 ? 8 : 1), sink, "far");
        suite<64, 256, 1024>(d, bytes, n16 / 256 / (mib == 1024 ? 8 : 1), sink, "far");
        suite<32, 128, 256>(d, bytes, n16 / 128 / (mib == 1024 ? 8 : 1), sink, "far");
        suite<64, 64, 256>(d, bytes, n16 / 64 / (mib == 1024 ? 8 : 1), sink, "far");
        suite<128, 64, 1024>(d, bytes, n16 / 64 / (mib == 1024 ? 8 : 1), sink, "far");
        // last-pass like: rows close together (L = 256 elements = 128 x 16 B)
        suite<8, 256, 256>(d, bytes, 128, sink, "near");
        suite<32, 256, 1024>(d, bytes, 128, sink, "near");
        // 2-pass shapes: 4096 rows
        suite<4, 4096, 1024>(d, bytes, 2048, sink, "4096r");
        suite<8, 4096, 1024>(d, bytes, 2048, sink, "4096r");
        CK(hipFree(d));
    }
    return 0;
}
