// Latency of ONE BLAKE2b compression spread over four lanes (blake2b_quad.hpp) for a lone wave: the unit the upper levels of every
// Merkle tree and the late FRI rounds are made of.  A chain of dependent compressions (digest fed back as the message) per quad.
// Measured (MI355X): 1.49 us per compression incl. two barriers and the LDS hand-over; requesting the message words of round R+1 from
// LDS before the arithmetic of round R: 1.38 us here, nothing measurable in the FRI round kernel (whose 16-20 us for a tiny round are
// launch ~5, fold ~2, leaf encoding on one lane ~3, leaf hash (2-3 blocks) ~4.5, 1.7 per level, fence 0.5) -- not kept.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I stark_brainfuck_amd/csrc -o tools/microbench/b2lat tools/microbench/b2lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "blake2b_quad.hpp"
using namespace bfs;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int VARIANT>
__global__ void __launch_bounds__(64) chain(u64* out, int steps) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ u64 msg[16][16];          // per quad: 16 message words
    const u32 lane = threadIdx.x, quad = lane >> 2, j = lane & 3;
    const QuadLane q = quad_lane(lane);
    for (int i = 0; i < 4; ++i) msg[quad][4 * j + i] = 0x0123456789abcdefULL * (lane + 1) + i;
    __syncthreads();
    u64 hl, hh;
    for (int s = 0; s < steps; ++s) {
        blake2b_init_quad(q, hl, hh);
        blake2b_compress_quad(q, hl, hh, msg[quad], 128, true);
        __syncthreads();
        msg[quad][j] = hl; msg[quad][4 + j] = hh;      // digest -> left child of the next "parent"
        __syncthreads();
    }
    out[lane] = hl ^ hh;
#endif
}

template <int VARIANT>
void run(const char* tag, u64* d_out) {
    const int steps = 2000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(chain<VARIANT>, dim3(1), dim3(64), 0, 0, d_out, 10);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(chain<VARIANT>, dim3(1), dim3(64), 0, 0, d_out, steps);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    u64 h[64]; CK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
    printf("%-28s %.3f us per compression   (check %016llx)\n", tag, 1e3 * ms / steps, (unsigned long long)h[5]);
}

int main() {
    u64* d_out; CK(hipMalloc(&d_out, 64 * 8));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("quad compress", d_out);
    }
    return 0;
}
