// waitvalue.hip -- how long after a host write does a pre-queued kernel start, when it sits behind hipStreamWaitValue64 on a word of
// pinned host memory -- against launching the kernel at that moment (development tool; round-5 question: could the FRI rounds be queued
// ahead, with the host only publishing each challenge?).  hipcc --offload-arch=gfx950 -O3 -o waitvalue waitvalue.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void mark_kernel(volatile uint64_t* done, const volatile uint64_t* value, uint64_t seq) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { const uint64_t v = *value; __threadfence_system(); *done = seq + (v & 0); }
}

int main() {
    uint64_t *flag, *done, *value;
    CK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc((void**)&done, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc((void**)&value, 64, hipHostMallocMapped | hipHostMallocCoherent));
    uint64_t *d_flag, *d_done, *d_value;
    CK(hipHostGetDevicePointer((void**)&d_flag, flag, 0));
    CK(hipHostGetDevicePointer((void**)&d_done, done, 0));
    CK(hipHostGetDevicePointer((void**)&d_value, value, 0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *flag = 0; *done = 0; *value = 7;
    const int reps = 200;
    std::vector<double> a, b;
    // (a) launch at the moment the value is known
    for (int r = 1; r <= reps; ++r) {
        CK(hipStreamSynchronize(st));
        const double t0 = now_us();
        hipLaunchKernelGGL(mark_kernel, dim3(1), dim3(64), 0, st, d_done, d_value, (uint64_t)r);
        while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != (uint64_t)r) {}
        a.push_back(now_us() - t0);
    }
    // (b) the kernel is queued ahead behind a wait on `flag`; the host publishes the value and sets the flag
    hipError_t e = hipStreamWaitValue64(st, d_flag, 1, hipStreamWaitValueEq, 0xFFFFFFFFFFFFFFFFull);
    if (e != hipSuccess) { printf("hipStreamWaitValue64 on mapped host memory: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); }
    else {
        *flag = 1;
        CK(hipStreamSynchronize(st));
        for (int r = 1; r <= reps; ++r) {
            *flag = 0; *done = 0;
            CK(hipStreamWaitValue64(st, d_flag, (uint64_t)r, hipStreamWaitValueEq, 0xFFFFFFFFFFFFFFFFull));
            hipLaunchKernelGGL(mark_kernel, dim3(1), dim3(64), 0, st, d_done, d_value, (uint64_t)r);
            const double spin = now_us();
            while (now_us() - spin < 30.0) {}                  // the queue has drained to the wait by now
            const double t0 = now_us();
            *value = (uint64_t)r;
            __atomic_store_n(flag, (uint64_t)r, __ATOMIC_RELEASE);
            while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != (uint64_t)r) {}
            b.push_back(now_us() - t0);
        }
    }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("launch when the value is known -> kernel's write visible on the host: median %.1f us (best %.1f)\n", a[a.size() / 2], a[0]);
    if (!b.empty()) printf("kernel queued behind hipStreamWaitValue64, host sets the flag -> write visible:  median %.1f us (best %.1f)\n", b[b.size() / 2], b[0]);
    return 0;
}
