// valu_rate.hip -- issue cost of the integer VALU opcode classes the NTT kernels are made of, on gfx950 (development tool; run through
// gpurun: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate).
//
// Question (round-4 verdict, next #5b): the roofline accounting prices one wave64 VALU instruction at 4 SIMD cycles (16 lanes per
// clock), while the micro-architecture guide has a "2 cycles (SIMD-32)" line.  Which is it for the opcodes this code uses, and is
// v_lshl_add_u64 -- a 64-bit add without a carry-out -- cheaper than the v_add_co / v_addc_co pair?
//
// Method: every thread runs a long block of ONE instruction class over EIGHT independent dependency chains (so that instruction
// latency never limits a wave), W one-wave workgroups per SIMD (W = 1, 2, 4, 6, 8) on every SIMD of the chip, kernels of milliseconds
// (every wave resident for the whole kernel).  Reported: the chip-wide rate in wave-instructions per second (HIP events), the same per
// SIMD and per cycle of the in-kernel counter (s_memtime; its rate is measured against the events: median wave ticks / kernel time),
// and the ticks one wave spends per instruction.  If a SIMD retires one wave64 instruction per 4 cycles the per-SIMD figure saturates
// at 0.25 per cycle; at 2 cycles (32 lanes per clock) at 0.5.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

enum Op { ADD_U32, XOR_B32, ADD_CO_PAIR, LSHL_ADD_U64, CNDMASK, MAD_U64_U32, SUBB_SGPR, MUL_LO_U32, ALIGNBIT, ADD_LAZY4, PERM_B32, ALIGNBYTE, LSHL_OR, XOR3, ADD3, LSHLREV, CNDMASK_CMP, MIX_XA, MIX_G, OPS };
static const char* NAMES[OPS] = {"v_add_u32", "v_xor_b32", "v_add_co_u32 + v_addc_co_u32 (vcc)", "v_lshl_add_u64", "v_cndmask_b32 (vcc)",
                                 "v_mad_u64_u32", "v_sub_co + s_nop 1 + v_subb_co (SGPR pair)", "v_mul_lo_u32", "v_alignbit_b32",
                                 "gl_add_lazy shape: add_co, addc_co, cndmask, add_co... (4 instr)", "v_perm_b32", "v_alignbyte_b32", "v_lshl_or_b32",
                                 "v_bitop3_b32 (three-input xor)", "v_add3_u32", "v_lshlrev_b32", "v_cmp_lt_u32 + v_cndmask_b32 (vcc written every time)",
                                 "mix: v_xor_b32, v_alignbit_b32 alternating (class prices 2 + 4)",
                                 "mix, a quarter of BLAKE2b's G: v_lshl_add_u64, v_xor_b32 x 2, v_alignbit_b32 x 2 (class prices 4 + 2 + 2 + 4 + 4)"};
static const int INSTR_PER_STEP[OPS] = {1, 1, 2, 1, 1, 1, 2, 1, 1, 4, 1, 1, 1, 1, 1, 1, 2, 2, 5};

// one step = the class applied to chain k.  Everything is inline asm (volatile) so the compiler neither folds nor reorders the work;
// the loop is unrolled by hand through the macro below.
template <int OP>
__device__ __forceinline__ void step(uint32_t& lo, uint32_t& hi, uint32_t c) {
    if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo) : "v"(c));
    else if (OP == XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(lo) : "v"(c));
    else if (OP == ADD_CO_PAIR) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(lo), "+v"(hi) : "v"(c) : "vcc");
    else if (OP == LSHL_ADD_U64) {
        uint64_t x = ((uint64_t)hi << 32) | lo, y = c;
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(y));
        lo = (uint32_t)x; hi = (uint32_t)(x >> 32);
    } else if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(lo) : "v"(c) : "vcc");
    else if (OP == MAD_U64_U32) {
        uint64_t x = ((uint64_t)hi << 32) | lo;
        asm volatile("v_mad_u64_u32 %0, s[40:41], %1, %2, %0" : "+v"(x) : "v"(lo), "v"(c) : "s40", "s41");
        lo = (uint32_t)x; hi = (uint32_t)(x >> 32);
    } else if (OP == SUBB_SGPR) asm volatile("v_sub_co_u32 %0, s[42:43], %0, %2\n\ts_nop 1\n\tv_subb_co_u32 %1, s[42:43], %1, %2, s[42:43]"
                                             : "+v"(lo), "+v"(hi) : "v"(c) : "s42", "s43");
    else if (OP == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo) : "v"(c));
    else if (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(lo) : "v"(hi));
    else if (OP == PERM_B32) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(hi), "v"(c));
    else if (OP == ALIGNBYTE) asm volatile("v_alignbyte_b32 %0, %0, %1, 3" : "+v"(lo) : "v"(hi));
    else if (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(lo) : "v"(hi));
    else if (OP == XOR3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(lo) : "v"(hi), "v"(c));
    else if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(lo) : "v"(hi), "v"(c));
    else if (OP == LSHLREV) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(lo));
    else if (OP == CNDMASK_CMP) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(lo) : "v"(hi), "v"(c) : "vcc");
    else if (OP == MIX_XA) asm volatile("v_xor_b32 %0, %0, %2\n\tv_alignbit_b32 %1, %1, %0, 7" : "+v"(lo), "+v"(hi) : "v"(c));
    else if (OP == MIX_G) {
        uint64_t x = ((uint64_t)hi << 32) | lo, y = c;
        asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(y));
        lo = (uint32_t)x; hi = (uint32_t)(x >> 32);
        asm volatile("v_xor_b32 %0, %0, %2\n\tv_xor_b32 %1, %1, %2\n\tv_alignbit_b32 %0, %0, %1, 24\n\tv_alignbit_b32 %1, %1, %0, 24" : "+v"(lo), "+v"(hi) : "v"(c));
    }
    else if (OP == ADD_LAZY4) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_add_co_u32 %0, vcc, %0, %1"
                                           : "+v"(lo), "+v"(hi) : "v"(c) : "vcc");
}

constexpr int CHAINS = 8, ROUNDS = 256;              // ROUNDS * CHAINS steps per thread, fully unrolled (2048 steps)

constexpr int OUTER = 200;                          // the unrolled block is repeated: kernels of milliseconds, every wave resident throughout

template <int OP>
__global__ void __launch_bounds__(64) rate_kernel(uint64_t* cycles, uint32_t* sink, uint32_t seed, int outer) {
    uint32_t lo[CHAINS], hi[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) { lo[k] = seed * (k + 3) + threadIdx.x; hi[k] = seed ^ (k * 2654435761u); }
    const uint32_t c = seed | 1;
    __builtin_amdgcn_s_barrier();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < outer; ++it) {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
            for (int k = 0; k < CHAINS; ++k) step<OP>(lo[k], hi[k], c);
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) acc ^= lo[k] ^ hi[k];
    sink[(size_t)blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(int waves_per_simd, uint64_t* d_cycles, uint32_t* d_sink, int num_cu) {
    // one-wave workgroups, waves_per_simd * 4 * num_cu of them: with kernels of milliseconds they are all resident at once, W per
    // SIMD (a CU takes up to 8 waves per SIMD; the dispatcher fills CUs and SIMDs evenly)
    const int blocks = waves_per_simd * 4 * num_cu;
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(64), 0, 0, d_cycles, d_sink, 12345u, 20);     // warm-up (code cache, clocks)
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(64), 0, 0, d_cycles, d_sink, 777u, OUTER);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<uint64_t> h(blocks);
    CK(hipMemcpy(h.data(), d_cycles, blocks * sizeof(uint64_t), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[blocks / 2];
    const double instr = (double)OUTER * ROUNDS * CHAINS * INSTR_PER_STEP[OP];
    // chip-wide: wave-instructions per second, and what one SIMD spends per wave-instruction if the clock was `mhz`
    const double rate = (double)blocks * instr / (ms * 1e-3);
    const double tick_mhz = med / (ms * 1e3);            // the median wave runs (almost) the whole kernel
    printf("  W = %d: kernel %7.3f ms, %6.1f G wave-instr/s chip-wide = %5.3f per SIMD per tick-counter cycle (counter %.0f MHz); one wave: %.2f ticks per instruction\n",
           waves_per_simd, ms, rate / 1e9, rate / (4.0 * num_cu) / (tick_mhz * 1e6), tick_mhz, med / instr);
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

template <int OP>
void sweep(uint64_t* d_cycles, uint32_t* d_sink, int num_cu) {
    printf("%s\n", NAMES[OP]);
    for (int w : {1, 2, 4, 6, 8}) run<OP>(w, d_cycles, d_sink, num_cu);
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int num_cu = prop.multiProcessorCount;
    printf("%s, %d CUs, clockRate %d kHz (the tick counter below is s_memtime: it runs at a constant rate -- see the calibration line)\n", prop.name, num_cu, prop.clockRate);
    uint64_t* d_cycles; uint32_t* d_sink;
    CK(hipMalloc(&d_cycles, 8 * 4 * num_cu * sizeof(uint64_t)));
    CK(hipMalloc(&d_sink, (size_t)8 * 4 * num_cu * 64 * sizeof(uint32_t)));
    if (argc > 1 && argv[1][0] == 'm') {        // "mix": do the class prices add up when the classes alternate?
        sweep<XOR_B32>(d_cycles, d_sink, num_cu);
        sweep<ALIGNBIT>(d_cycles, d_sink, num_cu);
        sweep<MIX_XA>(d_cycles, d_sink, num_cu);
        sweep<MIX_G>(d_cycles, d_sink, num_cu);
        return 0;
    }
    sweep<ADD_U32>(d_cycles, d_sink, num_cu);
    sweep<XOR_B32>(d_cycles, d_sink, num_cu);
    sweep<ADD_CO_PAIR>(d_cycles, d_sink, num_cu);
    sweep<LSHL_ADD_U64>(d_cycles, d_sink, num_cu);
    sweep<CNDMASK>(d_cycles, d_sink, num_cu);
    sweep<MAD_U64_U32>(d_cycles, d_sink, num_cu);
    sweep<SUBB_SGPR>(d_cycles, d_sink, num_cu);
    sweep<MUL_LO_U32>(d_cycles, d_sink, num_cu);
    sweep<ALIGNBIT>(d_cycles, d_sink, num_cu);
    sweep<ADD_LAZY4>(d_cycles, d_sink, num_cu);
    sweep<PERM_B32>(d_cycles, d_sink, num_cu);
    sweep<ALIGNBYTE>(d_cycles, d_sink, num_cu);
    sweep<LSHL_OR>(d_cycles, d_sink, num_cu);
    sweep<XOR3>(d_cycles, d_sink, num_cu);
    sweep<ADD3>(d_cycles, d_sink, num_cu);
    sweep<LSHLREV>(d_cycles, d_sink, num_cu);
    sweep<CNDMASK_CMP>(d_cycles, d_sink, num_cu);
    return 0;
}
