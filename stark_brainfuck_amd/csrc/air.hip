// air.hip -- the STARK-specific kernels around the transforms:
//   * randomized trace interpolation         Table.interpolate_columns   /root/reference/code/table.py:112-136
//   * boundary / transition / terminal quotient codewords   table.py:148-168, 176-236, 249-281  (constraints: air_generated.hpp)
//   * difference quotients of the permutation arguments      permutation_argument.py:9-18
//   * the non-linear combination codeword                    brainfuck_stark.py:236-300
// Points of the FRI domain are x_i = offset * omega^i (fri.py:20-21); codewords are column-major: base column c at
// base[c * n + i], extension column c as three limb planes at ext[(3 c + limb) * n + i].
#include <algorithm>
#include <cstddef>
#include <vector>

#include "air_generated.hpp"
#include "lazy.hpp"
#include "runtime.hpp"

namespace bfs {

int ntt_power_tables(u64 root, u32 log_n, const u64** lo, const u64** hi, u32* lo_bits);

// ---- randomized interpolation ------------------------------------------------------------------------------------
// The reference interpolates each trace column over {omicron^i} plus one extra point (omega, random value r) with a
// generic subproduct-tree routine (ntt.py:82-161).  The interpolant is unique, so it equals
//     f = f0 + c * (X^h - 1),   f0 = INTT_h(column),   c = (r - f0(omega)) / (omega^h - 1)
// Two small kernels: evaluate f0 at `point` (strided Horner, several workgroups per column), then patch f[0] and f[h].
struct RandomizerValues {
    u64 v[64];           // the random values travel in the kernel arguments: no copy, no synchronisation
};

// Stage 1, grid (parts, columns): thread g of a column's T = 256 * parts threads sums f_k point^k over k = g, g + T, ... by Horner in
// point^T (consecutive lanes read consecutive coefficients), times point^g; the block's sum goes to partial[column][part].
__global__ void __launch_bounds__(256) poly_evaluate_partial_kernel(const u64* coeffs, u64 stride, u64 h, u64 point, u64* partial, u32 first) {
    __shared__ u64 part[256];
    const u64* f = coeffs + (u64)(first + blockIdx.y) * stride;
    const u32 t = threadIdx.x;
    const u64 T = (u64)gridDim.x * 256, g = (u64)blockIdx.x * 256 + t;
    u64 acc = 0;
    if (g < h) {
        const u64 step = gl_pow(point, T);
        u64 k = g + (h - 1 - g) / T * T;                     // the largest index of this thread's progression
        for (;; k -= T) {
            acc = gl_add(gl_mul(acc, step), f[k]);
            if (k == g) break;
        }
        acc = gl_mul(acc, gl_pow(point, g));
    }
    part[t] = acc;
    __syncthreads();
    for (u32 s = 128; s > 0; s >>= 1) {
        if (t < s) part[t] = gl_add(part[t], part[t + s]);
        __syncthreads();
    }
    if (t == 0) partial[(u64)blockIdx.y * gridDim.x + blockIdx.x] = part[0];
}

// Stage 2, one thread per column: c = (r - f0(point)) / (point^h - 1), then f[0] -= c and f[h] = c.
__global__ void poly_randomize_finish_kernel(u64* coeffs, u64 stride, u64 h, u64 inv_den, RandomizerValues r, const u64* partial, u32 parts,
                                             u32 first, u32 count) {
    const u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= count) return;
    u64 sum = 0;
    for (u32 p = 0; p < parts; ++p) sum = gl_add(sum, partial[(u64)b * parts + p]);
    u64* f = coeffs + (u64)(first + b) * stride;
    const u64 c = gl_mul(gl_sub(r.v[b], sum), inv_den);
    f[0] = gl_sub(f[0], c);
    f[h] = c;
}

// Which coefficient indices of a polynomial are non-zero, condensed to what the prover needs (table.ext_sharing_moduli): bit 63 =
// the constant coefficient is non-zero; the low bits = OR over the non-zero indices j > 0 of (j & -j), so the lowest set bit is 2^v
// with v the 2-adic valuation common to all of them.  grid (parts, polynomials); `out` starts at zero.
__global__ void __launch_bounds__(256) coefficient_support_kernel(const u64* coeffs, u64 stride, u64 len, u64* out) {
    __shared__ u64 part[256];
    const u64* f = coeffs + (u64)blockIdx.y * stride;
    u64 acc = 0;
    for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < len; j += (u64)gridDim.x * 256)
        if (f[j] != 0) acc |= j ? (j & (0 - j)) : (1ull << 63);
    part[threadIdx.x] = acc;
    __syncthreads();
    for (u32 s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] |= part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && part[0]) atomicOr((unsigned long long*)(out + blockIdx.y), (unsigned long long)part[0]);
}

// ---- quotients -----------------------------------------------------------------------------------------------------
struct AirArgs {
    const u64* base;
    const u64* ext;
    u64* out;
    u64 n;
    u64 first, count;    // the points this launch works on: [first, first + count) of the domain (the whole domain, or a rank's rows)
    u64 unit_distance;
    u64 height;          // 0: the transition zerofier has no inverse and the reference multiplies by 0 (table.py:181-184)
    u32 log_height;
    u64 omicron_inv;
    u64 offset;
    const u64* w_lo;
    const u64* w_hi;
    u32 lo_bits;
    Xfe ch[11];
    Xfe tm[5];
    Xfe pr[1];
};

template <int TABLE> struct AirShape;
template <> struct AirShape<0> { static constexpr int BW = airgen::PROCESSOR_BASE_WIDTH, XW = airgen::PROCESSOR_EXT_WIDTH, NB = airgen::PROCESSOR_NUM_BOUNDARY, NT = airgen::PROCESSOR_NUM_TRANSITION, NZ = airgen::PROCESSOR_NUM_TERMINAL; };
template <> struct AirShape<1> { static constexpr int BW = airgen::INSTRUCTION_BASE_WIDTH, XW = airgen::INSTRUCTION_EXT_WIDTH, NB = airgen::INSTRUCTION_NUM_BOUNDARY, NT = airgen::INSTRUCTION_NUM_TRANSITION, NZ = airgen::INSTRUCTION_NUM_TERMINAL; };
template <> struct AirShape<2> { static constexpr int BW = airgen::MEMORY_BASE_WIDTH, XW = airgen::MEMORY_EXT_WIDTH, NB = airgen::MEMORY_NUM_BOUNDARY, NT = airgen::MEMORY_NUM_TRANSITION, NZ = airgen::MEMORY_NUM_TERMINAL; };
template <> struct AirShape<3> { static constexpr int BW = airgen::INPUT_BASE_WIDTH, XW = airgen::INPUT_EXT_WIDTH, NB = airgen::INPUT_NUM_BOUNDARY, NT = airgen::INPUT_NUM_TRANSITION, NZ = airgen::INPUT_NUM_TERMINAL; };
template <> struct AirShape<4> { static constexpr int BW = airgen::OUTPUT_BASE_WIDTH, XW = airgen::OUTPUT_EXT_WIDTH, NB = airgen::OUTPUT_NUM_BOUNDARY, NT = airgen::OUTPUT_NUM_TRANSITION, NZ = airgen::OUTPUT_NUM_TERMINAL; };

// the NEXT row of a thread kept in LDS ([value][thread], 64-bit words) and fetched where the constraint code uses it: 38 fewer live
// registers in the processor table's combine kernel (A/B switch BFS_COMBINE_LDS_NEXT, profiles/r03/ab_combine_lds_next.txt)
struct LdsNextBase {
    const u64* p;
    __device__ __forceinline__ u64 operator[](int k) const { return p[k * 256]; }
};
struct LdsNextExt {
    const u64* p;
    __device__ __forceinline__ Xfe operator[](int k) const { return Xfe{{p[(3 * k) * 256], p[(3 * k + 1) * 256], p[(3 * k + 2) * 256]}}; }
};

template <int TABLE, class Sink, class BN, class XN>
__device__ __forceinline__ void air_eval(const u64* bc, BN bn, const Xfe* xc, XN xn, const AirArgs& a, Sink& sink) {
    if constexpr (TABLE == 0) airgen::air_processor(bc, bn, xc, xn, a.ch, a.tm, a.pr, sink);
    else if constexpr (TABLE == 1) airgen::air_instruction(bc, bn, xc, xn, a.ch, a.tm, a.pr, sink);
    else if constexpr (TABLE == 2) airgen::air_memory(bc, bn, xc, xn, a.ch, a.tm, a.pr, sink);
    else if constexpr (TABLE == 3) airgen::air_input(bc, bn, xc, xn, a.ch, a.tm, a.pr, sink);
    else airgen::air_output(bc, bn, xc, xn, a.ch, a.tm, a.pr, sink);
}

// zerofier inverses at x with ONE field inversion (Montgomery's trick over a = x - 1, b = x - omicron^-1, c = x^h - 1):
//   boundary 1 / a (table.py:153-155), terminal 1 / b (:253-256), transition b / c (:180-188; 0 for an empty table)
struct Zerofiers {
    u64 boundary, transition, terminal;
    __device__ __forceinline__ Zerofiers(const AirArgs& a, u64 x) {
        const u64 za = gl_sub(x, 1), xo = gl_sub(x, a.omicron_inv);
        u64 zc = 1;
        if (a.height != 0) {
            u64 xh = x;
            for (u32 s = 0; s < a.log_height; ++s) xh = gl_sqr(xh);
            zc = gl_sub(xh, 1);
        }
        const u64 ab = gl_mul(za, xo);
        const u64 iabc = gl_inv(gl_mul(ab, zc));
        const u64 iab = gl_mul(iabc, zc);
        boundary = gl_mul(iab, xo);
        terminal = gl_mul(iab, za);
        transition = a.height != 0 ? gl_mul(xo, gl_mul(iabc, ab)) : 0;
    }
    // from precomputed inverse codewords (zerofier_inverses_kernel): no inversion here
    __device__ __forceinline__ Zerofiers(const AirArgs& a, u64 x, u64 inv_x_minus_1, u64 inv_x_minus_omicron_inv, u64 inv_xh_minus_1) {
        boundary = inv_x_minus_1;
        terminal = inv_x_minus_omicron_inv;
        transition = a.height != 0 ? gl_mul(gl_sub(x, a.omicron_inv), inv_xh_minus_1) : 0;
    }
    template <int TABLE, int Q>
    __device__ __forceinline__ u64 of() const {
        typedef AirShape<TABLE> S;
        return Q < S::NB ? boundary : (Q < S::NB + S::NT ? transition : terminal);
    }
};

// the values leave as soon as the generated code has them: scaled by the zerofier inverse and stored ...
template <int TABLE>
struct QuotientStore {
    u64* out;
    u64 n, i;
    Zerofiers z;
    template <int Q> __device__ __forceinline__ void put(const Xfe& v) {
        const Xfe r = xfe_scale(v, z.template of<TABLE, Q>());
#pragma unroll
        for (int l = 0; l < 3; ++l) out[(u64)(3 * Q + l) * n + i] = r.c[l];
    }
    template <int Q> __device__ __forceinline__ void put_base(u64 v) {
        out[(u64)(3 * Q) * n + i] = gl_mul(v, z.template of<TABLE, Q>());
        out[(u64)(3 * Q + 1) * n + i] = 0;
        out[(u64)(3 * Q + 2) * n + i] = 0;
    }
};

template <int TABLE>
__global__ void __launch_bounds__(256) air_quotient_kernel(const AirArgs a) {
    typedef AirShape<TABLE> S;
    // one point per thread, no grid-stride loop: a loop would let the compiler hoist the (loop-invariant) challenges and weights out
    // of it into registers -- hundreds of them
    const u64 i = a.first + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.first + a.count) {
        u64 j = i + a.unit_distance;
        if (j >= a.n) j -= a.n;
        u64 bc[S::BW], bn[S::BW];
        Xfe xc[S::XW], xn[S::XW];
#pragma unroll
        for (int c = 0; c < S::BW; ++c) { bc[c] = a.base[(u64)c * a.n + i]; bn[c] = a.base[(u64)c * a.n + j]; }
#pragma unroll
        for (int c = 0; c < S::XW; ++c)
#pragma unroll
            for (int l = 0; l < 3; ++l) { xc[c].c[l] = a.ext[(u64)(3 * c + l) * a.n + i]; xn[c].c[l] = a.ext[(u64)(3 * c + l) * a.n + j]; }
        const u64 x = gl_mul(a.offset, tw_pow(a.w_lo, a.w_hi, a.lo_bits, i));
        QuotientStore<TABLE> sink{a.out, a.n, i, Zerofiers(a, x)};
        air_eval<TABLE>(bc, bn, xc, xn, a, sink);
    }
}

// ---- zerofier inverses for all tables at once --------------------------------------------------------------------
// Every table divides by x - 1 (boundary), x - omicron^-1 (terminal) and x^h - 1 (transition); with five tables and two
// permutation arguments that was seven field inversions per point (~2300 instructions each), a third of the combination's
// arithmetic.  One kernel computes all distinct denominators of a proof at a point and inverts them together (Montgomery's
// trick: one inversion + 3 (K - 1) multiplications for K <= 12 values); the table kernels read the codewords.
struct ZerofierSpecs {
    u32 count;
    u32 is_power[12];     // 0: x - value      1: x^(2^value) - 1
    u64 value[12];
};

__global__ void __launch_bounds__(256) zerofier_inverses_kernel(ZerofierSpecs sp, u64* out, u64 n, u64 offset, const u64* w_lo, const u64* w_hi,
                                                                u32 lo_bits, u64 first, u64 count) {
    const u64 i = first + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= first + count) return;
    const u64 x = gl_mul(offset, tw_pow(w_lo, w_hi, lo_bits, i));
    u64 v[12], prefix[12];
    u64 running = 1;
#pragma unroll
    for (u32 k = 0; k < 12; ++k) {
        if (k < sp.count) {
            if (sp.is_power[k]) {
                u64 xh = x;
                for (u32 s = 0; s < (u32)sp.value[k]; ++s) xh = gl_sqr(xh);
                v[k] = gl_sub(xh, 1);
            } else {
                v[k] = gl_sub(x, sp.value[k]);
            }
            prefix[k] = running;
            running = gl_mul(running, v[k]);
        }
    }
    u64 inv = gl_inv(running);
#pragma unroll
    for (int k = 11; k >= 0; --k) {
        if ((u32)k < sp.count) {
            out[(u64)k * n + i] = gl_mul(inv, prefix[k]);
            inv = gl_mul(inv, v[k]);
        }
    }
}

// ---- quotients folded straight into the non-linear combination ---------------------------------------------------
// The prover never opens a quotient codeword (the verifier recomputes quotient VALUES from the opened trace rows,
// brainfuck_stark.py:470-560), so the production path does not write them: one kernel per table evaluates the constraints at a
// point, divides by the zerofiers and adds the table's share of the combination
//     sum over its base columns, extension columns and quotients s of (wa_s + wb_s x^shift_s) * value_s
// to the accumulator -- with the trace values it already holds in registers.  Against air_quotient_kernel + combination_kernel
// this saves writing and re-reading 60 extension codewords and re-reading the 49 trace columns (13 GB at N = 2^22).
// x^shift = offset^shift * omega^(i * shift mod n): two table loads and two multiplications instead of a square-and-multiply chain.
struct CombW {
    Xfe wa, wb;
    u64 offset_pow;      // offset^shift
    u64 shift;
};

// Round 3: the sum is taken as  sum_kind z_kind * [ sum_k wa_k v_k  +  sum_runs x^shift_run * sum_{k in run} wb_k v_k ]  with the
// inner sums accumulated UNREDUCED (lazy.hpp: a weight is a 3 x 3 matrix of residues laid out by the host, the lanes only
// multiply-accumulate into 64-bit columns) and x^shift applied once per run of terms that share it -- all columns of a table do
// (their bound is the interpolant degree), and neighbouring quotients of one kind with the same generic degree do
// (air_generated.hpp: *_Q_DEGREE).  The launcher compares the actual shifts and takes the GROUPED = false instantiation (every
// term its own run) when a run's shifts differ, which crafted challenges can cause.  Per extension term that is 18 products of
// 8 instructions where the reduced form spent 14 modular multiplications of 19 and 12 modular additions.
template <int TABLE>
struct CombineLayout {
    typedef AirShape<TABLE> S;
    static constexpr int NC = S::BW + S::XW, NQ = S::NB + S::NT + S::NZ, NTERM = NC + NQ;
    static constexpr bool q_ext(int q) {
        if constexpr (TABLE == 0) return airgen::PROCESSOR_Q_EXT[q];
        else if constexpr (TABLE == 1) return airgen::INSTRUCTION_Q_EXT[q];
        else if constexpr (TABLE == 2) return airgen::MEMORY_Q_EXT[q];
        else if constexpr (TABLE == 3) return airgen::INPUT_Q_EXT[q];
        else return airgen::OUTPUT_Q_EXT[q];
    }
    static constexpr int q_degree(int q) {
        if constexpr (TABLE == 0) return airgen::PROCESSOR_Q_DEGREE[q];
        else if constexpr (TABLE == 1) return airgen::INSTRUCTION_Q_DEGREE[q];
        else if constexpr (TABLE == 2) return airgen::MEMORY_Q_DEGREE[q];
        else if constexpr (TABLE == 3) return airgen::INPUT_Q_DEGREE[q];
        else return airgen::OUTPUT_Q_DEGREE[q];
    }
    // term k: base columns, extension columns, quotients -- the order of the reference's `terms` list restricted to one table
    static constexpr bool is_ext(int k) { return k < S::BW ? false : (k < NC ? true : q_ext(k - NC)); }
    static constexpr int kind(int k) { return k < NC ? 0 : (k - NC < S::NB ? 1 : (k - NC < S::NB + S::NT ? 2 : 3)); }      // columns, boundary, transition, terminal
    static constexpr bool continues_run(int k) {          // generic expectation: term k has the shift of term k - 1
        if (k == 0 || kind(k) != kind(k - 1)) return false;
        return kind(k) == 0 ? true : q_degree(k - NC) == q_degree(k - 1 - NC);
    }
    static constexpr int words(int k) { return 2 * (is_ext(k) ? LAZY_W_EXT : LAZY_W_BASE); }
    static constexpr int offset(int k) {
        int o = 0;
        for (int j = 0; j < k; ++j) o += words(j);
        return o;
    }
    static constexpr int NW = offset(NTERM);
};

// Per-term data the lanes read: staged in LDS by the kernel.  As plain kernel arguments the compiler fetched all of it (~600 dwords for
// the processor table) with scalar loads at the top of the kernel and spilled it to vector-register lanes at once: 709 v_writelane +
// 715 v_readlane of 11 600 VALU instructions per thread (profiles/r03/ab_combine_lds_weights.txt).  LDS reads of a wave-uniform
// address are broadcasts, and the compiler places them next to their use.
template <int TABLE>
struct CombineTerms {
    u64 w[CombineLayout<TABLE>::NW];               // per term: wa then wb * offset^shift, 3 words each (base value) or 7 each (lazy.hpp)
    u32 shift[(CombineLayout<TABLE>::NTERM + 1) & ~1];
};

template <int TABLE>
struct AirCombineArgs {
    AirArgs a;
    const u64* randomizer;   // non-null: the accumulator starts from w0 * randomizer (first kernel of a proof)
    Xfe w0;
    u64* acc;                // three limb planes of n
    const u64 *inv_boundary, *inv_terminal, *inv_transition;   // codewords of 1/(x - 1), 1/(x - omicron^-1), 1/(x^h - 1), or all null
    CombineTerms<TABLE> terms;
};

// The constraints arrive kind by kind (boundary, transition, terminal); all quotients of a kind share their zerofier inverse,
// so the weighted VALUES of a kind are summed first and the inverse is applied once.  x^shift = offset^shift omega^(i shift): the
// host folds offset^shift into wb, and the product of a run's sum with omega^(i shift) goes into the kind's unreduced sum as well.
template <int TABLE, bool GROUPED>
struct CombineSink {
    typedef CombineLayout<TABLE> Lay;
    const AirArgs& a;
    const CombineTerms<TABLE>& T;   // in LDS
    u64 i;
    Zerofiers z;
    Xfe acc;
    LazyX sa, sb;            // sum over the kind so far of wa_k v_k and of the closed runs' omega^(i shift) * sum; sum of wb_k v_k over the run so far

    __device__ __forceinline__ u64 omega_pow(int k) const {
        return tw_pow(a.w_lo, a.w_hi, a.lo_bits, (i * T.shift[k]) & (a.n - 1));
    }
    template <int K> __device__ __forceinline__ void close_run() {           // K: any term of the run that ends
        lazyx_mac_scale(sa, lazyx_reduce(sb), omega_pow(K));
        sb = lazyx_zero();
    }
    template <int KIND> __device__ __forceinline__ void close_kind() {
        const Xfe kind_sum = lazyx_reduce(sa);
        sa = lazyx_zero();
        if constexpr (KIND == 0) acc = xfe_add(acc, kind_sum);
        else acc = xfe_add(acc, xfe_scale(kind_sum, KIND == 1 ? z.boundary : (KIND == 2 ? z.transition : z.terminal)));
    }
    template <int K> __device__ __forceinline__ void open() {                // what ends where term K begins
        if constexpr (K > 0) {
            constexpr bool new_kind = Lay::kind(K) != Lay::kind(K - 1);
            if constexpr (new_kind || !GROUPED || !Lay::continues_run(K)) close_run<K - 1>();
            if constexpr (new_kind) close_kind<Lay::kind(K - 1)>();
        }
    }
    template <int K> __device__ __forceinline__ void term_base(u64 v) {
        static_assert(!Lay::is_ext(K), "layout and generated code disagree");
        open<K>();
        lazyx_mac_base(sa, T.w + Lay::offset(K), v);
        lazyx_mac_base(sb, T.w + Lay::offset(K) + LAZY_W_BASE, v);
    }
    template <int K> __device__ __forceinline__ void term_ext(const Xfe& v) {
        static_assert(Lay::is_ext(K), "layout and generated code disagree");
        open<K>();
        lazyx_mac_ext(sa, T.w + Lay::offset(K), v);
        lazyx_mac_ext(sb, T.w + Lay::offset(K) + LAZY_W_EXT, v);
    }
    __device__ __forceinline__ void finish() {
        close_run<Lay::NTERM - 1>();
        close_kind<Lay::kind(Lay::NTERM - 1)>();
    }
    // the generated code's interface
    template <int Q> __device__ __forceinline__ void put(const Xfe& v) { term_ext<Lay::NC + Q>(v); }
    template <int Q> __device__ __forceinline__ void put_base(u64 v) { term_base<Lay::NC + Q>(v); }
};

template <int TABLE, int C, class Sink>
__device__ __forceinline__ void combine_columns(Sink& sink, const u64* bc, const Xfe* xc) {
    typedef AirShape<TABLE> S;
    if constexpr (C < S::BW) { sink.template term_base<C>(bc[C]); combine_columns<TABLE, C + 1>(sink, bc, xc); }
    else if constexpr (C < S::BW + S::XW) { sink.template term_ext<C>(xc[C - S::BW]); combine_columns<TABLE, C + 1>(sink, bc, xc); }
}

// waves per SIMD asked of the register allocator: with the per-term data in LDS the scheduler otherwise pulls the reads far ahead of
// their use and fills all 256 registers the block size allows (tables 1, 3, 4: 73-126 -> 250)
#ifndef BFS_COMBINE_WAVES
#define BFS_COMBINE_WAVES 3, 3, 4, 4, 4
#endif
#ifndef BFS_COMBINE_LDS_NEXT
#define BFS_COMBINE_LDS_NEXT 1       // bit t: table t keeps its next row in LDS (LdsNextBase / LdsNextExt)
#endif
constexpr int combine_waves(int table) {
    constexpr int w[5] = {BFS_COMBINE_WAVES};
    return w[table];
}
template <int TABLE, bool GROUPED>
__global__ void __launch_bounds__(256, combine_waves(TABLE)) air_combine_kernel(const AirCombineArgs<TABLE> A) {
    typedef AirShape<TABLE> S;
    __shared__ CombineTerms<TABLE> T;
    {
        // the raw kernel-argument segment, read with a per-thread index (a vector load; indexing A itself at run time would make the
        // compiler copy the whole struct to scratch)
        const u64* karg = (const u64*)__builtin_amdgcn_kernarg_segment_ptr();
        constexpr u32 first = offsetof(AirCombineArgs<TABLE>, terms) / 8, words = sizeof(CombineTerms<TABLE>) / 8;
        static_assert(offsetof(AirCombineArgs<TABLE>, terms) % 8 == 0 && sizeof(CombineTerms<TABLE>) % 8 == 0, "staged as 64-bit words");
        u64* dst = reinterpret_cast<u64*>(&T);
        for (u32 k = threadIdx.x; k < words; k += 256) dst[k] = karg[first + k];
    }
    __syncthreads();
    const AirArgs& a = A.a;
    const u64 i = a.first + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.first + a.count) {
        u64 j = i + a.unit_distance;
        if (j >= a.n) j -= a.n;
        constexpr bool LDS_NEXT = ((BFS_COMBINE_LDS_NEXT) >> TABLE) & 1;
        __shared__ u64 next_row[LDS_NEXT ? (S::BW + 3 * S::XW) * 256 : 1];
        u64 bc[S::BW], bn[LDS_NEXT ? 1 : S::BW];
        Xfe xc[S::XW], xn[LDS_NEXT ? 1 : S::XW];
#pragma unroll
        for (int c = 0; c < S::BW; ++c) {
            bc[c] = a.base[(u64)c * a.n + i];
            if constexpr (LDS_NEXT) next_row[c * 256 + threadIdx.x] = a.base[(u64)c * a.n + j];
            else bn[c] = a.base[(u64)c * a.n + j];
        }
#pragma unroll
        for (int c = 0; c < S::XW; ++c)
#pragma unroll
            for (int l = 0; l < 3; ++l) {
                xc[c].c[l] = a.ext[(u64)(3 * c + l) * a.n + i];
                if constexpr (LDS_NEXT) next_row[(S::BW + 3 * c + l) * 256 + threadIdx.x] = a.ext[(u64)(3 * c + l) * a.n + j];
                else xn[c].c[l] = a.ext[(u64)(3 * c + l) * a.n + j];
            }
        Xfe acc;
        if (A.randomizer) acc = xfe_mul(A.w0, Xfe{{A.randomizer[i], A.randomizer[a.n + i], A.randomizer[2 * a.n + i]}});
        else acc = Xfe{{A.acc[i], A.acc[a.n + i], A.acc[2 * a.n + i]}};
        const u64 x = gl_mul(a.offset, tw_pow(a.w_lo, a.w_hi, a.lo_bits, i));
        CombineSink<TABLE, GROUPED> sink{a, T, i,
                                         A.inv_boundary ? Zerofiers(a, x, A.inv_boundary[i], A.inv_terminal[i], a.height != 0 ? A.inv_transition[i] : 0)
                                                        : Zerofiers(a, x),
                                         acc, lazyx_zero(), lazyx_zero()};
        combine_columns<TABLE, 0>(sink, bc, xc);
        if constexpr (LDS_NEXT) air_eval<TABLE>(bc, LdsNextBase{next_row + threadIdx.x}, xc, LdsNextExt{next_row + S::BW * 256 + threadIdx.x}, a, sink);
        else air_eval<TABLE>(bc, (const u64*)bn, xc, (const Xfe*)xn, a, sink);
        sink.finish();
        A.acc[i] = sink.acc.c[0];
        A.acc[a.n + i] = sink.acc.c[1];
        A.acc[2 * a.n + i] = sink.acc.c[2];
    }
}

// acc += (wa + wb x^shift) * (lhs - rhs) / (x - 1)      (the difference quotient of a permutation argument, folded the same way)
__global__ void difference_combine_kernel(const u64* lhs, const u64* rhs, u64* acc, u64 n, u64 offset, const u64* w_lo, const u64* w_hi,
                                          u32 lo_bits, CombW w, const u64* inv_x_minus_1, u64 first, u64 count) {
    for (u64 i = first + (u64)blockIdx.x * blockDim.x + threadIdx.x; i < first + count; i += (u64)gridDim.x * blockDim.x) {
        const u64 z = inv_x_minus_1 ? inv_x_minus_1[i] : gl_inv(gl_sub(gl_mul(offset, tw_pow(w_lo, w_hi, lo_bits, i)), 1));
        const u64 xs = gl_mul(w.offset_pow, tw_pow(w_lo, w_hi, lo_bits, (i * w.shift) & (n - 1)));
        const Xfe weight = xfe_add(w.wa, xfe_scale(w.wb, xs));
        Xfe q;
#pragma unroll
        for (int l = 0; l < 3; ++l) q.c[l] = gl_mul(gl_sub(lhs[(u64)l * n + i], rhs[(u64)l * n + i]), z);
        const Xfe r = xfe_add(Xfe{{acc[i], acc[n + i], acc[2 * n + i]}}, xfe_mul(weight, q));
        acc[i] = r.c[0];
        acc[n + i] = r.c[1];
        acc[2 * n + i] = r.c[2];
    }
}

// (lhs - rhs) / (x - 1) for two extension codewords (three limb planes of stride n each)
__global__ void difference_quotient_kernel(const u64* lhs, const u64* rhs, u64* out, u64 n, u64 offset, const u64* w_lo, const u64* w_hi, u32 lo_bits) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 x = gl_mul(offset, tw_pow(w_lo, w_hi, lo_bits, i));
        const u64 z = gl_inv(gl_sub(x, 1));
#pragma unroll
        for (int l = 0; l < 3; ++l) out[(u64)l * n + i] = gl_mul(gl_sub(lhs[(u64)l * n + i], rhs[(u64)l * n + i]), z);
    }
}

// ---- non-linear combination -------------------------------------------------------------------------------------
// sum over sources s of (wa_s + wb_s * x^shift_s) * codeword_s[i], plus w0 * randomizer[i]   (brainfuck_stark.py:236-300)
struct CombSrc {
    const u64* ptr;      // base: n values; extension: three planes of n
    u32 is_ext, pad;
    u64 shift;
    Xfe wa, wb;
};

__global__ void __launch_bounds__(256) combination_kernel(const CombSrc* srcs, u32 count, const u64* randomizer, Xfe w0, u64* out, u64 n,
                                                          u64 offset, const u64* w_lo, const u64* w_hi, u32 lo_bits) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 x = gl_mul(offset, tw_pow(w_lo, w_hi, lo_bits, i));
        Xfe acc = xfe_mul(w0, Xfe{{randomizer[i], randomizer[n + i], randomizer[2 * n + i]}});
        u64 shift = ~0ull, xs = 0;
        for (u32 s = 0; s < count; ++s) {
            const CombSrc& src = srcs[s];
            if (src.shift != shift) {          // sources arrive grouped by shift (host side): one power per distinct shift
                shift = src.shift;
                xs = gl_pow(x, shift);
            }
            const Xfe w = xfe_add(src.wa, xfe_scale(src.wb, xs));
            if (src.is_ext) acc = xfe_add(acc, xfe_mul(w, Xfe{{src.ptr[i], src.ptr[n + i], src.ptr[2 * n + i]}}));
            else acc = xfe_add(acc, xfe_scale(w, src.ptr[i]));
        }
        out[i] = acc.c[0];
        out[n + i] = acc.c[1];
        out[2 * n + i] = acc.c[2];
    }
}

static u32 grid_per_point(u64 n) { return (u32)((n + 255) / 256); }

static u32 grid_for(u64 n) {
    u64 g = (n + 255) / 256;
    return (u32)(g > 8192 ? 8192 : (g ? g : 1));
}

static Xfe xfe_from(const u64* p) { return Xfe{{p[0], p[1], p[2]}}; }

}  // namespace bfs

using namespace bfs;

extern "C" {

int bfs_poly_randomize(uint64_t* d_coeffs, uint64_t stride, uint64_t h, uint32_t batch, uint64_t point, const uint64_t* h_values, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0) return BFS_OK;
    if (h == 0 || stride < h + 1) { set_error("bfs_poly_randomize: need h >= 1 and stride >= h + 1"); return BFS_ERR_BAD_ARG; }
    const u64 den = gl_sub(gl_pow(point, h), 1);
    if (den == 0) { set_error("bfs_poly_randomize: the extra point lies on the interpolation subgroup"); return BFS_ERR_BAD_ARG; }
    u32 parts = (u32)((h + 1023) / 1024);
    if (parts > 64) parts = 64;
    void* w = nullptr;
    BFS_TRY(workspace(3, (size_t)64 * parts * sizeof(u64), stream, &w));
    for (u32 first = 0; first < batch; first += 64) {
        const u32 count = batch - first < 64 ? batch - first : 64;
        RandomizerValues r{};
        for (u32 k = 0; k < count; ++k) r.v[k] = h_values[first + k] % GL_P;
        hipLaunchKernelGGL(poly_evaluate_partial_kernel, dim3(parts, count), dim3(256), 0, stream, (const u64*)d_coeffs, stride, h, point, (u64*)w, first);
        hipLaunchKernelGGL(poly_randomize_finish_kernel, dim3((count + 63) / 64), dim3(64), 0, stream, d_coeffs, stride, h, gl_inv(den), r,
                           (const u64*)w, parts, first, count);
        BFS_HIP(hipGetLastError());
    }
    return BFS_OK;
}

int bfs_poly_support(const uint64_t* d_coeffs, uint64_t stride, uint64_t len, uint32_t batch, uint64_t* h_masks, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0) return BFS_OK;
    void* w = nullptr;
    BFS_TRY(workspace(3, (size_t)batch * sizeof(u64), stream, &w));
    BFS_HIP(hipMemsetAsync(w, 0, (size_t)batch * sizeof(u64), stream));
    u32 parts = (u32)((len + 2047) / 2048);
    if (parts > 64) parts = 64;
    hipLaunchKernelGGL(coefficient_support_kernel, dim3(parts ? parts : 1, batch), dim3(256), 0, stream, d_coeffs, stride, len, (u64*)w);
    BFS_HIP(hipGetLastError());
    BFS_HIP(hipMemcpyAsync(h_masks, w, (size_t)batch * sizeof(u64), hipMemcpyDeviceToHost, stream));
    BFS_HIP(hipStreamSynchronize(stream));
    return BFS_OK;
}

static int fill_air_args(AirArgs& a, int table, const uint64_t* d_base, const uint64_t* d_ext, uint32_t log_n, uint64_t unit_distance,
                         uint64_t height, uint64_t omicron_inv, uint64_t offset, uint64_t omega, const uint64_t* h_challenges,
                         const uint64_t* h_terminals, const uint64_t* h_params, const char* who) {
    if (table < 0 || table > 4) { set_error("%s: table index %d", who, table); return BFS_ERR_BAD_ARG; }
    if (height & (height - 1)) { set_error("%s: table height must be zero or a power of two", who); return BFS_ERR_NOT_POW2; }
    a.base = d_base; a.ext = d_ext; a.out = nullptr;
    a.n = 1ull << log_n;
    a.unit_distance = unit_distance % a.n;
    a.height = height;
    a.log_height = 0;
    while ((1ull << a.log_height) < height) ++a.log_height;
    a.omicron_inv = omicron_inv; a.offset = offset;
    BFS_TRY(ntt_power_tables(omega, log_n, &a.w_lo, &a.w_hi, &a.lo_bits));
    for (int i = 0; i < 11; ++i) a.ch[i] = xfe_from(h_challenges + 3 * i);
    a.first = 0;
    a.count = a.n;
    for (int i = 0; i < 5; ++i) a.tm[i] = xfe_from(h_terminals + 3 * i);
    a.pr[0] = h_params ? xfe_from(h_params) : Xfe{{1, 0, 0}};
    return BFS_OK;
}

int bfs_air_quotients(int table, const uint64_t* d_base, const uint64_t* d_ext, uint64_t* d_out, uint32_t log_n, uint64_t unit_distance,
                      uint64_t height, uint64_t omicron_inv, uint64_t offset, uint64_t omega, const uint64_t* h_challenges,
                      const uint64_t* h_terminals, const uint64_t* h_params, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    AirArgs a{};
    BFS_TRY(fill_air_args(a, table, d_base, d_ext, log_n, unit_distance, height, omicron_inv, offset, omega, h_challenges, h_terminals,
                          h_params, "bfs_air_quotients"));
    a.out = d_out;
    const u32 grid = grid_per_point(a.n);
    switch (table) {
        case 0: hipLaunchKernelGGL(air_quotient_kernel<0>, dim3(grid), dim3(256), 0, stream, a); break;
        case 1: hipLaunchKernelGGL(air_quotient_kernel<1>, dim3(grid), dim3(256), 0, stream, a); break;
        case 2: hipLaunchKernelGGL(air_quotient_kernel<2>, dim3(grid), dim3(256), 0, stream, a); break;
        case 3: hipLaunchKernelGGL(air_quotient_kernel<3>, dim3(grid), dim3(256), 0, stream, a); break;
        default: hipLaunchKernelGGL(air_quotient_kernel<4>, dim3(grid), dim3(256), 0, stream, a); break;
    }
    BFS_HIP(hipGetLastError());
    return BFS_OK;
}

}  // extern "C"

static CombW comb_weight(const bfs_comb_weight& w, u64 offset) {
    CombW r;
    r.wa = xfe_from(w.wa); r.wb = xfe_from(w.wb);
    r.shift = w.shift;
    r.offset_pow = gl_pow(offset, w.shift);
    return r;
}

template <int TABLE>
static int air_combine_launch(const AirArgs& a, const bfs_comb_weight* h_weights, const uint64_t* d_randomizer, const uint64_t* h_w0,
                              uint64_t* d_acc, const uint64_t* const* d_inverses, hipStream_t stream) {
    typedef CombineLayout<TABLE> Lay;
    AirCombineArgs<TABLE> A{};
    A.a = a;
    A.randomizer = d_randomizer;
    A.w0 = d_randomizer ? xfe_from(h_w0) : Xfe{{0, 0, 0}};
    A.acc = d_acc;
    if (d_inverses && d_inverses[0] && d_inverses[1] && (a.height == 0 || d_inverses[2])) {
        A.inv_boundary = d_inverses[0]; A.inv_terminal = d_inverses[1]; A.inv_transition = d_inverses[2];
    }
    bool grouped = true;
    u64 offset_pow = 1;
    for (int k = 0; k < Lay::NTERM; ++k) {
        const bfs_comb_weight& w = h_weights[k];
        if (w.shift >> 32) { set_error("bfs_air_combine: shift does not fit 32 bits"); return BFS_ERR_BAD_ARG; }
        A.terms.shift[k] = (u32)w.shift;
        offset_pow = (k > 0 && w.shift == h_weights[k - 1].shift) ? offset_pow : gl_pow(a.offset, w.shift);
        if (Lay::continues_run(k) && w.shift != h_weights[k - 1].shift) grouped = false;     // not the generic degree pattern
        u64* out = A.terms.w + Lay::offset(k);
        const Xfe wb = xfe_scale(xfe_from(w.wb), offset_pow);                // x^shift = offset^shift * omega^(i shift)
        if (Lay::is_ext(k)) {
            lazy_weight_matrix(xfe_from(w.wa), out);
            lazy_weight_matrix(wb, out + LAZY_W_EXT);
        } else {
            for (int l = 0; l < 3; ++l) { out[l] = w.wa[l]; out[LAZY_W_BASE + l] = wb.c[l]; }
        }
    }
    static_assert(sizeof(A) <= 4096, "kernel arguments");
    if (grouped) hipLaunchKernelGGL((air_combine_kernel<TABLE, true>), dim3(grid_per_point(a.count)), dim3(256), 0, stream, A);
    else hipLaunchKernelGGL((air_combine_kernel<TABLE, false>), dim3(grid_per_point(a.count)), dim3(256), 0, stream, A);
    BFS_HIP(hipGetLastError());
    return BFS_OK;
}

extern "C" {

static int check_rows(const char* who, uint32_t log_n, uint64_t first_row, uint64_t num_rows) {
    const u64 n = 1ull << log_n;
    if (first_row > n || num_rows > n - first_row) {
        set_error("%s: rows [%llu, %llu + %llu) are not inside the domain of %llu points", who, (unsigned long long)first_row,
                  (unsigned long long)first_row, (unsigned long long)num_rows, (unsigned long long)n);
        return BFS_ERR_BAD_ARG;
    }
    return BFS_OK;
}

int bfs_air_combine(int table, const uint64_t* d_base, const uint64_t* d_ext, uint32_t log_n, uint64_t unit_distance, uint64_t height,
                    uint64_t omicron_inv, uint64_t offset, uint64_t omega, const uint64_t* h_challenges, const uint64_t* h_terminals,
                    const uint64_t* h_params, const bfs_comb_weight* h_weights, const uint64_t* d_randomizer,
                    const uint64_t* h_randomizer_weight, uint64_t* d_acc, const uint64_t* const* d_zerofier_inverses, void* stream_) {
    return bfs_air_combine_rows(table, d_base, d_ext, log_n, unit_distance, height, omicron_inv, offset, omega, h_challenges, h_terminals, h_params,
                                h_weights, d_randomizer, h_randomizer_weight, d_acc, d_zerofier_inverses, 0, log_n > 32 ? 0 : 1ull << log_n, stream_);
}

int bfs_air_combine_rows(int table, const uint64_t* d_base, const uint64_t* d_ext, uint32_t log_n, uint64_t unit_distance, uint64_t height,
                         uint64_t omicron_inv, uint64_t offset, uint64_t omega, const uint64_t* h_challenges, const uint64_t* h_terminals,
                         const uint64_t* h_params, const bfs_comb_weight* h_weights, const uint64_t* d_randomizer,
                         const uint64_t* h_randomizer_weight, uint64_t* d_acc, const uint64_t* const* d_zerofier_inverses, uint64_t first_row,
                         uint64_t num_rows, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (log_n > 32) { set_error("bfs_air_combine: log_n"); return BFS_ERR_BAD_ARG; }
    BFS_TRY(check_rows("bfs_air_combine_rows", log_n, first_row, num_rows));
    if (num_rows == 0) return BFS_OK;
    AirArgs a{};
    BFS_TRY(fill_air_args(a, table, d_base, d_ext, log_n, unit_distance, height, omicron_inv, offset, omega, h_challenges, h_terminals,
                          h_params, "bfs_air_combine"));
    a.first = first_row;
    a.count = num_rows;
    switch (table) {
        case 0: return air_combine_launch<0>(a, h_weights, d_randomizer, h_randomizer_weight, d_acc, d_zerofier_inverses, stream);
        case 1: return air_combine_launch<1>(a, h_weights, d_randomizer, h_randomizer_weight, d_acc, d_zerofier_inverses, stream);
        case 2: return air_combine_launch<2>(a, h_weights, d_randomizer, h_randomizer_weight, d_acc, d_zerofier_inverses, stream);
        case 3: return air_combine_launch<3>(a, h_weights, d_randomizer, h_randomizer_weight, d_acc, d_zerofier_inverses, stream);
        default: return air_combine_launch<4>(a, h_weights, d_randomizer, h_randomizer_weight, d_acc, d_zerofier_inverses, stream);
    }
}

int bfs_zerofier_inverses(uint32_t log_n, uint64_t offset, uint64_t omega, uint32_t count, const uint32_t* h_is_power, const uint64_t* h_values,
                          uint64_t* d_out, void* stream_) {
    return bfs_zerofier_inverses_rows(log_n, offset, omega, count, h_is_power, h_values, d_out, 0, log_n > 32 ? 0 : 1ull << log_n, stream_);
}

int bfs_zerofier_inverses_rows(uint32_t log_n, uint64_t offset, uint64_t omega, uint32_t count, const uint32_t* h_is_power, const uint64_t* h_values,
                               uint64_t* d_out, uint64_t first_row, uint64_t num_rows, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (count == 0) return BFS_OK;
    if (count > 12 || log_n > 32) { set_error("bfs_zerofier_inverses: at most 12 denominators, log_n <= 32"); return BFS_ERR_BAD_ARG; }
    BFS_TRY(check_rows("bfs_zerofier_inverses_rows", log_n, first_row, num_rows));
    if (num_rows == 0) return BFS_OK;
    const u64 n = 1ull << log_n;
    ZerofierSpecs sp{};
    sp.count = count;
    for (u32 k = 0; k < count; ++k) {
        sp.is_power[k] = h_is_power[k] ? 1u : 0u;
        sp.value[k] = h_values[k];
        if (h_is_power[k] && h_values[k] > 32) { set_error("bfs_zerofier_inverses: exponent 2^%llu", (unsigned long long)h_values[k]); return BFS_ERR_BAD_ARG; }
        // the denominators must not vanish on the coset: offset not in the subgroup they cut out (the reference asserts nothing
        // here and would raise on the division); cheap host check for the linear ones
        if (!h_is_power[k] && gl_pow(gl_mul(h_values[k] % GL_P, gl_inv(offset)), n) == 1) {
            set_error("bfs_zerofier_inverses: x - %llu vanishes on the evaluation domain", (unsigned long long)h_values[k]);
            return BFS_ERR_BAD_ARG;
        }
    }
    const u64 *lo, *hi;
    u32 lo_bits;
    BFS_TRY(ntt_power_tables(omega, log_n, &lo, &hi, &lo_bits));
    hipLaunchKernelGGL(zerofier_inverses_kernel, dim3(grid_per_point(num_rows)), dim3(256), 0, stream, sp, d_out, n, offset, lo, hi, lo_bits, first_row,
                       num_rows);
    BFS_HIP(hipGetLastError());
    return BFS_OK;
}

int bfs_difference_combine(const uint64_t* d_lhs, const uint64_t* d_rhs, uint32_t log_n, uint64_t offset, uint64_t omega,
                           const bfs_comb_weight* h_weight, uint64_t* d_acc, const uint64_t* d_inv_x_minus_1, void* stream_) {
    return bfs_difference_combine_rows(d_lhs, d_rhs, log_n, offset, omega, h_weight, d_acc, d_inv_x_minus_1, 0, log_n > 32 ? 0 : 1ull << log_n, stream_);
}

int bfs_difference_combine_rows(const uint64_t* d_lhs, const uint64_t* d_rhs, uint32_t log_n, uint64_t offset, uint64_t omega,
                                const bfs_comb_weight* h_weight, uint64_t* d_acc, const uint64_t* d_inv_x_minus_1, uint64_t first_row,
                                uint64_t num_rows, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (log_n > 32) { set_error("bfs_difference_combine: log_n"); return BFS_ERR_BAD_ARG; }
    BFS_TRY(check_rows("bfs_difference_combine_rows", log_n, first_row, num_rows));
    if (num_rows == 0) return BFS_OK;
    const u64 n = 1ull << log_n;
    if (h_weight->shift >> 32) { set_error("bfs_difference_combine: shift does not fit 32 bits"); return BFS_ERR_BAD_ARG; }
    const u64 *lo, *hi;
    u32 lo_bits;
    BFS_TRY(ntt_power_tables(omega, log_n, &lo, &hi, &lo_bits));
    hipLaunchKernelGGL(difference_combine_kernel, dim3(grid_for(num_rows)), dim3(256), 0, stream, d_lhs, d_rhs, d_acc, n, offset, lo, hi, lo_bits,
                       comb_weight(*h_weight, offset), d_inv_x_minus_1, first_row, num_rows);
    BFS_HIP(hipGetLastError());
    return BFS_OK;
}

// the constraints of one table at ONE point, on the host (the verifier: brainfuck_stark.py:470-560 evaluates every constraint
// polynomial at the opened rows; table.py:283-311).  The same generated straight-line code the kernels run, compiled for the host.
int bfs_air_evaluate(int table, const uint64_t* base_row, const uint64_t* base_next, const uint64_t* ext_row, const uint64_t* ext_next,
                     const uint64_t* h_challenges, const uint64_t* h_terminals, const uint64_t* h_params, uint64_t* out) {
    if (table < 0 || table > 4) { set_error("bfs_air_evaluate: table index %d", table); return BFS_ERR_BAD_ARG; }
    Xfe ch[11], tm[5], pr[1], xc[4], xn[4], res[32];
    u64 bc[8], bn[8];
    // every operand is reduced on the way in: opened rows and terminals come out of a proof somebody else wrote, and the field
    // primitives below assume canonical residues (the reference reduces in every operation, algebra.py:89-99) -- round-4 advice
    auto canon = [](const uint64_t* l) { return Xfe{{l[0] % GL_P, l[1] % GL_P, l[2] % GL_P}}; };
    for (int i = 0; i < 11; ++i) ch[i] = canon(h_challenges + 3 * i);
    for (int i = 0; i < 5; ++i) tm[i] = canon(h_terminals + 3 * i);
    pr[0] = h_params ? canon(h_params) : Xfe{{1, 0, 0}};
    auto run = [&](auto shape, auto fn) {
        typedef decltype(shape) S;
        static_assert(S::BW <= 8 && S::XW <= 4 && S::NB + S::NT + S::NZ <= 32, "row buffers");
        for (int c = 0; c < S::BW; ++c) { bc[c] = base_row[c] % GL_P; bn[c] = base_next ? base_next[c] % GL_P : 0; }
        for (int c = 0; c < S::XW; ++c) { xc[c] = canon(ext_row + 3 * c); xn[c] = ext_next ? canon(ext_next + 3 * c) : Xfe{{0, 0, 0}}; }
        fn(bc, bn, xc, xn, ch, tm, pr, res);
        for (int q = 0; q < S::NB + S::NT + S::NZ; ++q)
            for (int l = 0; l < 3; ++l) out[3 * q + l] = res[q].c[l];
    };
    switch (table) {
        case 0: run(AirShape<0>(), airgen::air_processor_values); break;
        case 1: run(AirShape<1>(), airgen::air_instruction_values); break;
        case 2: run(AirShape<2>(), airgen::air_memory_values); break;
        case 3: run(AirShape<3>(), airgen::air_input_values); break;
        default: run(AirShape<4>(), airgen::air_output_values); break;
    }
    return BFS_OK;
}

int bfs_air_counts(int table, int counts[3]) {
    switch (table) {
        case 0: counts[0] = AirShape<0>::NB; counts[1] = AirShape<0>::NT; counts[2] = AirShape<0>::NZ; return BFS_OK;
        case 1: counts[0] = AirShape<1>::NB; counts[1] = AirShape<1>::NT; counts[2] = AirShape<1>::NZ; return BFS_OK;
        case 2: counts[0] = AirShape<2>::NB; counts[1] = AirShape<2>::NT; counts[2] = AirShape<2>::NZ; return BFS_OK;
        case 3: counts[0] = AirShape<3>::NB; counts[1] = AirShape<3>::NT; counts[2] = AirShape<3>::NZ; return BFS_OK;
        case 4: counts[0] = AirShape<4>::NB; counts[1] = AirShape<4>::NT; counts[2] = AirShape<4>::NZ; return BFS_OK;
    }
    set_error("bfs_air_counts: table index %d", table);
    return BFS_ERR_BAD_ARG;
}

int bfs_air_num_quotients(int table) {
    switch (table) {
        case 0: return AirShape<0>::NB + AirShape<0>::NT + AirShape<0>::NZ;
        case 1: return AirShape<1>::NB + AirShape<1>::NT + AirShape<1>::NZ;
        case 2: return AirShape<2>::NB + AirShape<2>::NT + AirShape<2>::NZ;
        case 3: return AirShape<3>::NB + AirShape<3>::NT + AirShape<3>::NZ;
        case 4: return AirShape<4>::NB + AirShape<4>::NT + AirShape<4>::NZ;
    }
    return -1;
}

int bfs_difference_quotient(const uint64_t* d_lhs, const uint64_t* d_rhs, uint64_t* d_out, uint32_t log_n, uint64_t offset, uint64_t omega, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const u64 n = 1ull << log_n;
    const u64 *lo, *hi;
    u32 lo_bits;
    BFS_TRY(ntt_power_tables(omega, log_n, &lo, &hi, &lo_bits));
    hipLaunchKernelGGL(difference_quotient_kernel, dim3(grid_for(n)), dim3(256), 0, stream, d_lhs, d_rhs, d_out, n, offset, lo, hi, lo_bits);
    BFS_HIP(hipGetLastError());
    return BFS_OK;
}

int bfs_combination(const bfs_comb_source* h_sources, uint32_t count, const uint64_t* d_randomizer, const uint64_t* h_randomizer_weight,
                    uint64_t* d_out, uint32_t log_n, uint64_t offset, uint64_t omega, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const u64 n = 1ull << log_n;
    static_assert(sizeof(CombSrc) == sizeof(bfs_comb_source), "layout of bfs_comb_source");
    void* w = nullptr;
    BFS_TRY(workspace(3, (size_t)(count ? count : 1) * sizeof(CombSrc), stream, &w));
    // the sum does not depend on the order of its terms: group the sources by shift so the kernel raises x to each distinct
    // shift once per point (11 distinct values among the 75 sources of a Brainfuck proof)
    std::vector<CombSrc> sorted((const CombSrc*)h_sources, (const CombSrc*)h_sources + count);
    std::stable_sort(sorted.begin(), sorted.end(), [](const CombSrc& l, const CombSrc& r) { return l.shift < r.shift; });
    if (count) BFS_HIP(hipMemcpyAsync(w, sorted.data(), (size_t)count * sizeof(CombSrc), hipMemcpyHostToDevice, stream));
    const u64 *lo, *hi;
    u32 lo_bits;
    BFS_TRY(ntt_power_tables(omega, log_n, &lo, &hi, &lo_bits));
    hipLaunchKernelGGL(combination_kernel, dim3(grid_for(n)), dim3(256), 0, stream, (const CombSrc*)w, count, d_randomizer,
                       xfe_from(h_randomizer_weight), d_out, n, offset, lo, hi, lo_bits);
    BFS_HIP(hipGetLastError());
    BFS_HIP(hipStreamSynchronize(stream));      // h_sources may be reused by the caller
    return BFS_OK;
}

}  // extern "C"
