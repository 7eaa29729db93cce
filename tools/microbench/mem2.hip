// HBM access-pattern microbenchmark, round 2: can narrow row segments (32 / 64 B) be made efficient by putting the workgroups
// that share a 128-byte line on the SAME XCD (blockIdx % 8 = XCD) so that the line crosses the fabric once?  (development tool)
// Pattern as mem.hip: the array is rows x L u64; a workgroup owns ROWS rows x SEG bytes and moves them in 16-byte pieces per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// REMAP: chunk index such that the 128/SEGB workgroups sharing a line are b, b+8, b+16, ... (same XCD, dispatched together)
template <int SEG16, int ROWS, int THREADS, int MODE, int REMAP>
__global__ void __launch_bounds__(THREADS) tile16(uint4* data, size_t L16, u64* sink) {
    constexpr int ITEMS = SEG16 * ROWS / THREADS;
    constexpr int SHARE = (SEG16 >= 8) ? 1 : 8 / SEG16;          // workgroups per 128-byte line
    const int tid = threadIdx.x;
    const int c = tid % SEG16, r0 = tid / SEG16;
    constexpr int RSTEP = THREADS / SEG16;
    const size_t nl = L16 / SEG16;
    size_t b = blockIdx.x;
    if (REMAP && SHARE > 1) {
        const size_t grp = b / (8 * SHARE), within = b % (8 * SHARE);
        const size_t xcd = within % 8, part = within / 8;
        b = (grp * 8 + xcd) * SHARE + part;
    }
    const size_t h = b / nl, lch = b % nl;
    uint4* base = data + h * ((size_t)ROWS * L16) + lch * SEG16 + c;
    uint4 x[ITEMS];
    if (MODE != 1) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) x[i] = base[(size_t)(r0 + i * RSTEP) * L16];
    } else {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) x[i] = make_uint4(tid, i, 3, 4);
    }
    if (MODE != 0) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) { uint4 v = x[i]; v.x += 1; base[(size_t)(r0 + i * RSTEP) * L16] = v; }
    } else {
        unsigned acc = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) acc ^= x[i].x ^ x[i].y ^ x[i].z ^ x[i].w;
        if (acc == 0x12345679u) sink[0] = acc;
    }
}

template <int SEG16, int ROWS, int THREADS, int MODE, int REMAP>
float run(uint4* d, size_t bytes, size_t L16, u64* sink, int reps = 5) {
    size_t blocks = bytes / 16 / ((size_t)SEG16 * ROWS);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((tile16<SEG16, ROWS, THREADS, MODE, REMAP>), dim3(blocks), dim3(THREADS), 0, 0, d, L16, sink);
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((tile16<SEG16, ROWS, THREADS, MODE, REMAP>), dim3(blocks), dim3(THREADS), 0, 0, d, L16, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

template <int SEG16, int ROWS, int THREADS>
void suite(uint4* d, size_t bytes, size_t L16, u64* sink, const char* tag) {
    float r0 = run<SEG16, ROWS, THREADS, 0, 0>(d, bytes, L16, sink), w0 = run<SEG16, ROWS, THREADS, 1, 0>(d, bytes, L16, sink), rw0 = run<SEG16, ROWS, THREADS, 2, 0>(d, bytes, L16, sink);
    float r1 = run<SEG16, ROWS, THREADS, 0, 1>(d, bytes, L16, sink), w1 = run<SEG16, ROWS, THREADS, 1, 1>(d, bytes, L16, sink), rw1 = run<SEG16, ROWS, THREADS, 2, 1>(d, bytes, L16, sink);
    printf("%-8s seg %4d B rows %4d thr %4d tile %3d KB | plain: read %5.0f write %5.0f r+w %5.0f | xcd-paired: read %5.0f write %5.0f r+w %5.0f GB/s\n", tag,
           SEG16 * 16, ROWS, THREADS, SEG16 * 16 * ROWS / 1024, bytes / r0 / 1e6, bytes / w0 / 1e6, 2.0 * bytes / rw0 / 1e6, bytes / r1 / 1e6, bytes / w1 / 1e6,
           2.0 * bytes / rw1 / 1e6);
}

int main() {
    u64* sink; CK(hipMalloc(&sink, 64));
    for (size_t mib : {1024, 128}) {
        size_t bytes = mib << 20;
        uint4* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 0, bytes));
        printf("---- array %zu MiB (columns of 2^24 u64: row length 4096 u64 = 2048 x 16 B for the 4096-row shapes)\n", mib);
        // two-pass shapes of a 2^24 transform: 4096 rows x 4096 columns per column of the batch
        suite<2, 4096, 1024>(d, bytes, 2048, sink, "4096r");
        suite<2, 4096, 512>(d, bytes, 2048, sink, "4096r");
        suite<4, 4096, 1024>(d, bytes, 2048, sink, "4096r");
        suite<8, 4096, 1024>(d, bytes, 2048, sink, "4096r");
        // 2048 / 1024 rows (11 / 10-bit digits)
        suite<2, 2048, 512>(d, bytes, 4096, sink, "2048r");
        suite<4, 2048, 1024>(d, bytes, 4096, sink, "2048r");
        suite<4, 1024, 512>(d, bytes, 8192, sink, "1024r");
        suite<8, 1024, 1024>(d, bytes, 8192, sink, "1024r");
        // three-pass shapes: 256 rows, 128 / 256 / 512-byte segments, rows 2^16 u64 apart
        suite<8, 256, 256>(d, bytes, 32768, sink, "256r");
        suite<16, 256, 512>(d, bytes, 32768, sink, "256r");
        suite<16, 256, 256>(d, bytes, 32768, sink, "256r");
        suite<32, 256, 512>(d, bytes, 32768, sink, "256r");
        CK(hipFree(d));
    }
    return 0;
}
