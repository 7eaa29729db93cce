// limb_block.hip -- prices the radix-16 butterfly block + twiddle product of the NTT tile kernels in its two forms (round 6, the
// "ceiling on paper first, then a microbenchmark" step of the round-5 verdict):
//   A  two 32-bit words per element, carry arithmetic: dif<16, lazy> + gl_mul            (ntt_core.hpp, gl.hpp: the shipped form)
//   B  four signed limbs of weight 2^24, carry-free adds, rotations for 2^(24 k): limb_split + limb_dif<16> + limb_mul   (gl_limb.hpp)
// Every thread holds 16 elements and runs ITERS x (block, 16 products with table entries from LDS); both kernels compute the same
// values (checked on the host and between the kernels).  Reported: ns per 16-element block chip-wide, blocks/s, and what that makes
// of one 8 x 2^24 pass' arithmetic (2 blocks per element-16 per pass).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I stark_brainfuck_amd/csrc -I tools/microbench -o tools/microbench/limb_block tools/microbench/limb_block.hip && ./limb_block
#define BFS_GL_SUB4
#include "ntt_core.hpp"
#include "gl_limb.hpp"   // tools/microbench/gl_limb.hpp
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

using namespace bfs;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int TW_N = 256;

template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) block_words(const u64* in, u64* out, const u64* twg, int iters) {
    __shared__ u64 tw[TW_N];
    tw[threadIdx.x] = twg[threadIdx.x];
    __syncthreads();
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    u64 x[16];
    BFS_UNROLL
    for (int m = 0; m < 16; ++m) x[m] = in[base + m];
    for (int it = 0; it < iters; ++it) {
        dif<16, true>(x);
        const u32 e0 = (threadIdx.x * 7 + it * 13) & (TW_N - 1);
        BFS_UNROLL
        for (int m = 0; m < 16; ++m) x[m] = gl_mul(x[m], tw[(e0 + 17 * m) & (TW_N - 1)]);
    }
    BFS_UNROLL
    for (int m = 0; m < 16; ++m) out[base + m] = x[m];
}

template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) block_limbs(const u64* in, u64* out, const LimbTw* twg, int iters) {
    __shared__ LimbTw tw[TW_N];
    tw[threadIdx.x] = twg[threadIdx.x];
    __syncthreads();
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    u64 x[16];
    BFS_UNROLL
    for (int m = 0; m < 16; ++m) x[m] = in[base + m];
    for (int it = 0; it < iters; ++it) {
        L4 y[16];
        BFS_UNROLL
        for (int m = 0; m < 16; ++m) y[m] = limb_split(x[m]);
        limb_dif<16>(y);
        const u32 e0 = (threadIdx.x * 7 + it * 13) & (TW_N - 1);
        BFS_UNROLL
        for (int m = 0; m < 16; ++m) x[m] = limb_mul(y[m], tw[(e0 + 17 * m) & (TW_N - 1)]);
    }
    BFS_UNROLL
    for (int m = 0; m < 16; ++m) out[base + m] = x[m];
}

// block without product: limb form joined back (what a pass' last stage costs when no twiddle follows) against the words form
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) block_limbs_join(const u64* in, u64* out, int iters) {
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    u64 x[16];
    BFS_UNROLL
    for (int m = 0; m < 16; ++m) x[m] = in[base + m];
    for (int it = 0; it < iters; ++it) {
        L4 y[16];
        BFS_UNROLL
        for (int m = 0; m < 16; ++m) y[m] = limb_split(x[m]);
        limb_dif<16>(y);
        BFS_UNROLL
        for (int m = 0; m < 16; ++m) x[m] = limb_join(y[m]);
    }
    BFS_UNROLL
    for (int m = 0; m < 16; ++m) out[base + m] = x[m];
}
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) block_words_plain(const u64* in, u64* out, int iters) {
    const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    u64 x[16];
    BFS_UNROLL
    for (int m = 0; m < 16; ++m) x[m] = in[base + m];
    for (int it = 0; it < iters; ++it) dif<16, false>(x);
    BFS_UNROLL
    for (int m = 0; m < 16; ++m) out[base + m] = x[m];
}

static u64 rnd(u64& s) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return s;
}

static int host_check() {
    u64 s = 0x9E3779B97F4A7C15ull;
    int bad = 0;
    for (int trial = 0; trial < 20000; ++trial) {
        u64 x[16], y[16];
        for (int m = 0; m < 16; ++m) {
            u64 v = rnd(s) % GL_P;
            if (trial % 5 == 1) v = (m & 1) ? GL_P - 1 - (rnd(s) & 3) : (rnd(s) & 3);
            if (trial % 5 == 2) v = GL_P - 1;
            if (trial % 5 == 3) v = (m & 2) ? 0xFFFFFFFF00000000ull : 0x00000000FFFFFFFFull;
            x[m] = y[m] = v;
        }
        u64 w[16];
        for (int m = 0; m < 16; ++m) w[m] = trial % 7 == 0 ? GL_P - 1 : rnd(s) % GL_P;
        dif<16, false>(x);
        L4 l[16];
        for (int m = 0; m < 16; ++m) l[m] = limb_split(y[m]);
        limb_dif<16>(l);
        for (int m = 0; m < 16; ++m) {
            for (int i = 0; i < 4; ++i)
                if (l[m].l[i] >= (1u << 30)) { ++bad; }
            const u64 a = gl_mul(x[m], w[m]);
            const u64 b = limb_mul(l[m], limb_tw_of(w[m]));
            const u64 c = limb_join(l[m]);
            if (a != b || c != x[m]) {
                if (bad < 5) printf("host mismatch trial %d m %d: words %016llx limbs %016llx ; join %016llx block %016llx\n", trial, m,
                                    (unsigned long long)a, (unsigned long long)b, (unsigned long long)c, (unsigned long long)x[m]);
                ++bad;
            }
        }
    }
    printf("host check: %d mismatches over 20000 blocks (random, next to 0 and p, all p - 1, word edges)\n", bad);
    return bad;
}

template <typename F>
static float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); launch();
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return t[t.size() / 2];
}

int main() {
    if (host_check()) return 1;
    int dev_count = 0;
    if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count == 0) { printf("no GPU: host check only\n"); return 0; }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = 400;
    u64 s = 12345;
    std::vector<u64> tw(TW_N);
    std::vector<LimbTw> ltw(TW_N);
    for (int i = 0; i < TW_N; ++i) { tw[i] = rnd(s) % GL_P; ltw[i] = limb_tw_of(tw[i]); }
    u64 *dtw; LimbTw* dltw;
    CK(hipMalloc(&dtw, TW_N * 8)); CK(hipMalloc(&dltw, TW_N * sizeof(LimbTw)));
    CK(hipMemcpy(dtw, tw.data(), TW_N * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dltw, ltw.data(), TW_N * sizeof(LimbTw), hipMemcpyHostToDevice));
    printf("%s, %d CUs; %d iterations of (radix-16 block + 16 twiddle products) per thread\n", prop.name, cus, iters);
    for (int wg_per_cu : {4, 5, 6, 8}) {
        const int blocks = cus * wg_per_cu;
        const size_t n = (size_t)blocks * 256 * 16;
        std::vector<u64> h(n);
        for (auto& v : h) v = rnd(s) % GL_P;
        u64 *din, *da, *db;
        CK(hipMalloc(&din, n * 8)); CK(hipMalloc(&da, n * 8)); CK(hipMalloc(&db, n * 8));
        CK(hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice));
        const float ta = time_ms([&] { hipLaunchKernelGGL(block_words<6>, dim3(blocks), dim3(256), 0, 0, din, da, dtw, iters); }, 7);
        const float tb = time_ms([&] { hipLaunchKernelGGL(block_limbs<4>, dim3(blocks), dim3(256), 0, 0, din, db, dltw, iters); }, 7);
        const float tb5 = time_ms([&] { hipLaunchKernelGGL(block_limbs<5>, dim3(blocks), dim3(256), 0, 0, din, db, dltw, iters); }, 7);
        const float tb6 = time_ms([&] { hipLaunchKernelGGL(block_limbs<6>, dim3(blocks), dim3(256), 0, 0, din, db, dltw, iters); }, 7);
        hipLaunchKernelGGL(block_limbs<4>, dim3(blocks), dim3(256), 0, 0, din, db, dltw, iters);
        CK(hipDeviceSynchronize());
        std::vector<u64> ha(n), hb(n);
        CK(hipMemcpy(ha.data(), da, n * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), db, n * 8, hipMemcpyDeviceToHost));
        size_t diff = 0;
        for (size_t i = 0; i < n; ++i) diff += ha[i] != hb[i];
        const double blk = (double)blocks * 256 * iters;              // 16-element blocks
        printf("  %d workgroups per CU: words %7.3f ms = %6.1f ps per element-block ; limbs(4 waves) %7.3f ms, (5) %7.3f, (6) %7.3f -> best ratio %.3f ; %zu of %zu outputs differ\n",
               wg_per_cu, ta, 1e9 * ta / (blk * 16), tb, tb5, tb6, std::min(tb, std::min(tb5, tb6)) / ta, diff, n);
        const float tpa = time_ms([&] { hipLaunchKernelGGL(block_words_plain<6>, dim3(blocks), dim3(256), 0, 0, din, da, iters); }, 7);
        const float tpb = time_ms([&] { hipLaunchKernelGGL(block_limbs_join<5>, dim3(blocks), dim3(256), 0, 0, din, db, iters); }, 7);
        CK(hipMemcpy(ha.data(), da, n * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hb.data(), db, n * 8, hipMemcpyDeviceToHost));
        diff = 0;
        for (size_t i = 0; i < n; ++i) diff += ha[i] != hb[i];
        printf("     block alone, canonical out: words %7.3f ms ; limbs + join %7.3f ms (ratio %.3f) ; %zu differ\n", tpa, tpb, tpb / tpa, diff);
        CK(hipFree(din)); CK(hipFree(da)); CK(hipFree(db));
    }
    return 0;
}
