// rows.hip -- commitment to "zipped" codewords: leaf i is the TUPLE of the i-th elements of several codewords plus a salt,
// the reference's SaltedMerkle(list(zip(*codewords)))  (/root/reference/code/brainfuck_stark.py:178-179,197-198 with
// salted_merkle.py:22-47): leaf preimage = pickle.dumps(tuple) || pickle.dumps(salt), 0.8-1.3 KB per row.
//
// The bytes of a row's pickle are a fixed skeleton with the integers of the row spliced in, and the skeleton depends only on
// how many coefficients each extension element of the row stores (0..3, trailing zeros are dropped: extension_field.py:6-9):
// the memo indices that later back-references use shift with that count.  So:
//   1. a kernel computes every row's PATTERN (2 bits per extension column);
//   2. the host builds one TEMPLATE per pattern that actually occurs (normally one or two) by running the generic pickle
//      emitter (refpickle.hpp) on a row of sentinel values and cutting the result around the integer opcodes;
//   3. the leaf kernel expands the template of each row -- the template is flattened on the host into steps of at most 8 preimage
//      bytes (constant bytes carried in the step, integers encoded on the fly as BININT1/2, BININT or LONG1 like CPython's save_long,
//      frame length patched in) which a wave walks in lockstep -- and feeds the bytes straight into BLAKE2b: a 200-byte buffer per
//      lane in LDS, compressed by the whole wave whenever the lane furthest ahead has filled it.  The preimage never exists in memory.
#include <algorithm>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "blake2b.hpp"
#include "leaf_encode.hpp"
#include "refpickle.hpp"
#include "rows_core.hpp"
#include "rows_generated.hpp"
#include "runtime.hpp"

namespace bfs {

int merkle_inner_launch(u64* d_nodes, u32 depth, u64 n_leaves, hipStream_t stream, u64* root_out, u64 seq);

constexpr int ROW_MAX_COLS = 32;

// host-side description of a template: constant byte runs (in a word pool) and the places where a row's integers go
struct RowSeg {
    u32 kind;
    u32 a;      // CONST: word offset into the pool (segments start on a 64-bit word); INT: column
    u32 b;      // CONST: length in bytes; INT: limb
    u32 pad;
};
// (RowStep, the flattened form the kernel walks: rows_core.hpp)
struct RowTemplate {
    u32 code, first_step, num_steps, tuple_const_bytes, salt_bytes;
    u32 first_int, num_ints;     // into RowArgs::ints: (column | limb << 8) of every integer of the row, in preimage order
    u32 pad;
};
struct RowArgs {
    const u64* const* columns;   // device array of column pointers
    const u32* is_ext;           // device array
    u32 ncols;
    u32 n_ext;                   // extension columns among them ...
    u32 ext_cols[4];             // ... and which they are, one byte each, in column order
    u64 n;                       // rows to hash
    u64 limb_stride;             // distance between the limb planes of an extension column (n, or the full column length when
                                 // the rows are a range of longer columns)
    const u64* salts;            // n x 3 words, or null
    const RowTemplate* templates;
    u32 num_templates;
    const RowStep* steps;
    const u32* ints;
    u64* digests;                // n x 8 words
    u64* pattern_set;            // pattern kernel output: open-addressing set of the codes present (PATTERN_SLOTS entries, ~0 = empty)
    u32* error;                  // [0]: a row without a template, [1]: the pattern set overflowed
    u32 skip, skip_code;         // interpreter kernel, skip != 0: rows of pattern skip_code were hashed by a generated kernel
    const u64* block0_states;    // generated kernels: BLAKE2b's state behind block 0 of the preimage, 8 words per possible length of the
                                 // row's integers (2 .. 11 bytes each), first entry = all of them two bytes long
};

// 2 bits per extension column: how many coefficients its element of row i stores (trailing zero limbs are dropped).  The top limbs
// of all extension columns are requested together (the column pointers come through the scalar unit, RowArgs::ext_cols lists the
// columns): one memory latency for the common row whose top limbs are all non-zero, not one per column.
__device__ __forceinline__ u32 row_pattern(const RowArgs& a, u64 i) {
    typedef const u64* const __attribute__((address_space(4)))* Columns;
    typedef const u64 __attribute__((address_space(1)))* Words;
    const Columns columns = (Columns)a.columns;
    u64 top[16];
#pragma unroll
    for (u32 j = 0; j < 16; ++j) {
        top[j] = 1;
        if (j < a.n_ext) top[j] = ((Words)columns[(a.ext_cols[j / 4] >> (8 * (j % 4))) & 0xFF])[2 * a.limb_stride + i];
    }
    u32 code = 0;
#pragma unroll
    for (u32 j = 0; j < 16; ++j) {
        if (j < a.n_ext) {
            u32 k = 3;
            if (top[j] == 0) {
                const Words p = (Words)columns[(a.ext_cols[j / 4] >> (8 * (j % 4))) & 0xFF];
                k = p[a.limb_stride + i] ? 2u : (p[i] ? 1u : 0u);
            }
            code |= k << (2 * j);
        }
    }
    return code;
}

// Which patterns occur?  Neighbouring rows almost always share theirs, so a lane only goes to the set when its code differs
// from the lane below, and the set is probed with a plain load before any atomic (after the first few waves every probe hits).
constexpr u32 PATTERN_SLOTS = 1024;
__global__ void row_pattern_kernel(const RowArgs a) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < a.n;
    const u32 code = valid ? row_pattern(a, i) : 0u;
    const u32 below = __shfl_up(code, 1);
    if (!valid || ((threadIdx.x & 63) != 0 && below == code)) return;
    u32 slot = (code * 2654435761u) >> 22;                      // 10 bits
    for (u32 tries = 0; tries < PATTERN_SLOTS; ++tries, slot = (slot + 1) & (PATTERN_SLOTS - 1)) {
        u64 seen = __atomic_load_n(a.pattern_set + slot, __ATOMIC_RELAXED);
        if (seen == ~0ull) seen = atomicCAS((unsigned long long*)(a.pattern_set + slot), ~0ull, (unsigned long long)code);
        if (seen == ~0ull || seen == code) return;
    }
    atomicOr(a.error + 1, 1u);
}

// One thread per row, 256 rows per workgroup.  The rows of a wave (almost always) share one template, so the walk over the template is
// WAVE-UNIFORM: step descriptors (with the constant bytes of the skeleton inside them) come through the scalar unit, the integers of
// a column are one coalesced load per wave, and a lane's own work per step is one unaligned LDS store at its own byte position
// (rows_core.hpp; the first version kept a segment cursor per lane: 50-80 divergent VALU instructions per 8 bytes, 53 % of the
// BLAKE2b floor).  One compression site in the code (the final, padded block goes through it as well), so the kernel stays inside
// the instruction cache.
constexpr u32 LEAF_THREADS = 256;
// the template is read through the scalar unit: constant address space + wave-uniform index = s_load (the data was written by a
// copy that completed before the launch, which is what the scalar cache needs)
typedef const RowStep __attribute__((address_space(4)))* ConstSteps;
typedef const u32 __attribute__((address_space(4)))* ConstInts;
typedef const u64* const __attribute__((address_space(4)))* ConstColumns;
typedef const u64 __attribute__((address_space(1)))* GlobalWords;
__device__ __forceinline__ u32 uniform32(u32 x) { return (u32)__builtin_amdgcn_readfirstlane(x); }   // (the builtin returns a signed int)
#ifndef BFS_ROW_WAVES
#define BFS_ROW_WAVES 4          // 128 VGPRs (20 bytes of scratch): with 38 KiB of LDS per workgroup the fourth wave per SIMD is there to be had
#endif
__global__ void __launch_bounds__(LEAF_THREADS, BFS_ROW_WAVES) row_leaves_kernel(const RowArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char blk[ROW_LANE_BYTES * LEAF_THREADS];
    const u32 lane = threadIdx.x;
    unsigned char* const buf = blk + ROW_LANE_BYTES * lane;
    const u64 i = (u64)blockIdx.x * LEAF_THREADS + lane;
    if (i >= a.n) return;
    const u32 code = row_pattern(a, i);
    if (a.skip && code == a.skip_code) return;
    u32 mine = ~0u;
    for (u32 t = 0; t < a.num_templates; ++t)
        if (a.templates[t].code == code) mine = t;
    if (mine == ~0u) { atomicOr(a.error, 1u); return; }
    for (u32 t = 0; t < a.num_templates; ++t) {      // t is wave-uniform; lanes with another template wait their turn
        if (mine != t) continue;
        const RowTemplate* tp = a.templates + t;
        // every lane here reads the same template: its fields go to scalar registers, and what hangs off them is read through the
        // scalar unit
        const u32 nsteps = uniform32(tp->num_steps), nints = uniform32(tp->num_ints);
        const u32 tuple_const_bytes = uniform32(tp->tuple_const_bytes), salt_bytes = uniform32(tp->salt_bytes);
        const ConstSteps steps = (ConstSteps)(a.steps + uniform32(tp->first_step));
        const ConstInts ints = (ConstInts)(a.ints + uniform32(tp->first_int));
        const ConstColumns columns = (ConstColumns)a.columns;
        auto row_int = [&](u32 packed) -> u64 {
            const GlobalWords col = (GlobalWords)columns[packed & 0xFF];
            return col[(u64)((packed >> 8) & 0xFF) * a.limb_stride + i];
        };
        // pass 1: length of the tuple pickle = constant bytes + the integer opcodes of this row
        u32 int_bytes = 0;
        {
            u32 j = 0;
            for (; j + 8 <= nints; j += 8) {         // eight loads in flight
                u64 w[8];
#pragma unroll
                for (u32 q = 0; q < 8; ++q) w[q] = row_int(ints[j + q]);
#pragma unroll
                for (u32 q = 0; q < 8; ++q) int_bytes += pickle_int_len(w[q]);
            }
            for (; j < nints; ++j) int_bytes += pickle_int_len(row_int(ints[j]));
        }
        const u32 tuple_len = tuple_const_bytes + int_bytes;
        // pass 2: expand the template into the lane's buffer, compressing block-synchronously.  ROW_UNROLL steps per turn of the loop
        // (the template is padded with empty steps): their descriptors are one scalar load.
        RowLane st8;
        row_lane_init(st8, tuple_len + salt_bytes);
        u64 int_hi = 0;             // the integer being written: bytes 8.. of its opcode, and how many they are
        u32 int_hi_bytes = 0;
        // the next integers of the row, already requested; slot = integer index mod 3 (the step says which), so a slot is only ever
        // touched by its own loads -- shifting values between registers would have to wait for loads still in flight
        u64 r0 = 0, r1 = 0, r2 = 0;
        if (nints) {
            r0 = row_int(ints[0]);
            r1 = row_int(ints[nints > 1 ? 1 : 0]);
            r2 = row_int(ints[nints > 2 ? 2 : 0]);
        }
        u32 k = 0;                  // wave-uniform step counter
        bool input_done = false;
        auto take = [&](const RowStep& st) {
            u64 data;
            u32 nb;
            if (st.kind == SEG_CONST) {
                data = st.data;
                nb = st.a;
            } else if (st.kind == SEG_INT) {
                const u32 slot = (u32)(st.data >> 24) & 0xFF, pf = (u32)(st.data >> 8) & 0xFFFF;
                u64 v;
                if (slot == 0) { v = r0; r0 = row_int(pf); }
                else if (slot == 1) { v = r1; r1 = row_int(pf); }
                else { v = r2; r2 = row_int(pf); }
                u32 len;
                if (__all(v >= (1ull << 31))) row_long1_opcode(v, data, int_hi, len);   // field elements are almost never small: no divergent branches
                else row_int_opcode(v, data, int_hi, len);
                nb = len < 8 ? len : 8;
                int_hi_bytes = len > 8 ? len - 8 : 0;
            } else if (st.kind == SEG_INT_HI) {       // what did not fit into the first eight bytes of the integer's opcode
                data = int_hi;
                nb = int_hi_bytes;
            } else if (st.kind == SEG_FRAMELEN) {
                data = (u64)tuple_len - 11;
                nb = 8;
            } else {
                data = a.salts[3 * i + st.a];
                nb = 8;
            }
            row_lane_put(st8, buf, data, nb);
        };
        while (true) {
            k = uniform32(k);       // (the same in every lane; the compiler cannot tell)
            // ---- the compression site
            bool want;
            if (!input_done) {
                want = row_lane_wants_mid(st8, __any(row_lane_full(st8)));
            } else {
                want = row_lane_wants_end(st8);
                if (!__any(want)) break;
            }
            if (want) row_lane_compress(st8, buf, input_done);
            if (input_done) continue;
            if (k < nsteps) {
                // the descriptors of the next ROW_UNROLL steps: one scalar load, straight into scalar registers
                RowStep st[ROW_UNROLL];
#pragma unroll
                for (u32 u = 0; u < ROW_UNROLL; ++u) { st[u].kind = steps[k + u].kind; st[u].a = steps[k + u].a; st[u].data = steps[k + u].data; }
#pragma unroll
                for (u32 u = 0; u < ROW_UNROLL; ++u) take(st[u]);
                k += ROW_UNROLL;
            } else {
                row_lane_finish(st8, buf);
                input_done = true;
            }
        }
        u64* out = a.digests + i * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = st8.h[j];
    }
}

// ---- the same walk as straight-line code, for the layouts rows_generated.hpp knows ----------------------------------------------
// row_leaves_kernel spends a third of its issue time on the scalar control flow of the interpreter: one taken branch per ten vector
// instructions, each of which restarts the wave's instruction fetch (profiles/r05/ab_rows_stalls.txt).  For a template known when the
// library is built, tools/gen_rows.py unrolls the walk: constants are immediates, every store of a segment has a fixed offset from the
// lane's position, and the only branch per segment is the (almost never taken) "is some lane full?".  The compression stays ONE site:
// Layout::segments is a coroutine -- it returns with `resume` = the next segment when the wave must compress, and is entered again
// through its switch.  Lanes whose row has another pattern than the layout's (a shorter extension element) leave at once: the
// interpreter kernel, launched behind this one when the remembered pattern set has more than one entry, hashes them.
#ifndef BFS_ROWGEN_PREFETCH
#define BFS_ROWGEN_PREFETCH 4          // integers of the row requested ahead of the one being written
#endif
template <class G>
struct RowGenLane {
    RowLane st8;
    unsigned char* buf;
    const RowArgs& a;
    u64 i;
    u32 tuple_len;
    u32 variant;                // which of the layout's patterns the rows are (wave-uniform: a kernel argument)
    u64 r[BFS_ROWGEN_PREFETCH];
    __device__ __forceinline__ RowGenLane(const RowArgs& a_, unsigned char* buf_, u64 i_, u32 variant_) : buf(buf_), a(a_), i(i_), tuple_len(0), variant(variant_) {}
    __device__ __forceinline__ u64 row_int(u32 packed) const {
        const GlobalWords col = (GlobalWords)((ConstColumns)a.columns)[packed & 0xFF];
        return col[(u64)((packed >> 8) & 0xFF) * a.limb_stride + i];
    }
    // (the constant is made where it is stored: left to itself the compiler hoists all of them out of the loop around the compression
    // site, as loop invariants, and then spills them -- 428 bytes of scratch per lane)
    __device__ __forceinline__ void st(u32 off, u64 data) {
        u32 lo = (u32)data, hi = (u32)(data >> 32);
        asm volatile("" : "+v"(lo), "+v"(hi));
        row_store8(buf + st8.pos + off, lo | ((u64)hi << 32));
    }
    __device__ __forceinline__ void adv(u32 n) { st8.pos += n; }
    __device__ __forceinline__ void framelen(u32 off) { row_store8(buf + st8.pos + off, (u64)tuple_len - 11); }
    __device__ __forceinline__ void salt(u32 off, u32 w) { row_store8(buf + st8.pos + off, a.salts[3 * i + w]); }
    // integer K of the row (V < 0: of the part all variants share; else of variant V's own part)
    template <u32 K, int V = -1> __device__ __forceinline__ void integer() {
        constexpr u32 D = BFS_ROWGEN_PREFETCH;
        const u64 v = r[K % D];
        if constexpr (V >= 0) {
            if constexpr (K + D < G::NUM_INTS[V >= 0 ? V : 0]) r[K % D] = row_int(G::INTS[V >= 0 ? V : 0][K + D]);
        } else if constexpr (K + D < G::COMMON_INTS) {
            r[K % D] = row_int(G::INTS[0][K + D]);
        } else {
            if (K + D < G::NUM_INTS[variant]) r[K % D] = row_int(G::INTS[variant][K + D]);      // (wave-uniform)
        }
        u64 lo, hi;
        u32 len;
        if (__all(v >= (1ull << 31))) row_long1_opcode(v, lo, hi, len);     // field elements are almost never small
        else row_int_opcode(v, lo, hi, len);
        row_store8(buf + st8.pos, lo);
        row_store8(buf + st8.pos + 8, hi);      // (zeros above the opcode's length; what follows overwrites them)
        st8.pos += len;
    }
    __device__ __forceinline__ bool full() const { return __any(st8.pos > ROW_LANE_BYTES - 16); }
    __device__ __forceinline__ void finish() { row_lane_finish(st8, buf); }
};

template <class G>
__global__ void __launch_bounds__(LEAF_THREADS, BFS_ROW_WAVES) row_leaves_generated_kernel(const RowArgs a, const u32 variant_, const u32 others_follow) {
    __shared__ __attribute__((aligned(16))) unsigned char blk[ROW_LANE_BYTES * LEAF_THREADS];
    const u32 lane = threadIdx.x;
    const u64 i = (u64)blockIdx.x * LEAF_THREADS + lane;
    if (i >= a.n) return;
    const u32 variant = uniform32(variant_);
    RowGenLane<G> c(a, blk + ROW_LANE_BYTES * lane, i, variant);
    // pass 1: the length of the tuple pickle, all integers of the row requested at once.  A row is this kernel's when it has the
    // variant's pattern, i.e. stores exactly the integers the variant lists: the top coefficient of every listed element is not zero
    // (a limb below a listed one may be), and every extension element the variant does not list in full is zero above what is listed.
    {
        if (row_pattern(a, i) != G::CODES[variant]) {
            if (!others_follow) atomicOr(a.error, 1u);
            return;
        }
        const u32 nints = G::NUM_INTS[variant];
        u64 w[G::MAX_INTS];
#pragma unroll
        for (u32 q = 0; q < G::MAX_INTS; ++q) w[q] = c.row_int(q < G::COMMON_INTS ? G::INTS[0][q] : G::INTS[variant][q]);   // (padded with the last one)
        u32 int_bytes = 0;
#pragma unroll
        for (u32 q = 0; q < G::MAX_INTS; ++q) int_bytes += (q < G::COMMON_INTS || q < nints) ? pickle_int_len(w[q]) : 0u;
        c.tuple_len = G::TUPLE_CONST_BYTES[variant] + int_bytes;
#pragma unroll
        for (u32 q = 0; q < BFS_ROWGEN_PREFETCH; ++q) c.r[q] = w[q < G::COMMON_INTS ? q : 0];
        static_assert(BFS_ROWGEN_PREFETCH <= G::COMMON_INTS, "the first integers are the same in every variant");
    }
    row_lane_init(c.st8, c.tuple_len + G::SALT_BYTES);
    // Block 0 is constants and the frame length (the first integer of a row comes ~50 bytes later): its compression is a table look-up by
    // length, and the walk starts with the bytes of the segment that straddles byte 128.
    u32 resume = 0;
    {
        const u64* ms = a.block0_states + (size_t)(c.tuple_len - G::TUPLE_CONST_BYTES[variant] - 2 * G::NUM_INTS[variant]) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) c.st8.h[j] = ms[j];
        c.st8.consumed = 128;
        c.st8.hashed_any = 1;
        G::enter_after_block0(c, resume);
    }
    while (true) {
        resume = uniform32(resume);             // (the same in every lane; the compiler cannot tell)
        if (resume < G::NUM_SEGMENTS) G::segments(c, resume);
        const bool at_end = uniform32(resume) >= G::NUM_SEGMENTS;
        // ---- the compression site
        const bool want = at_end ? row_lane_wants_end(c.st8) : row_lane_wants_mid(c.st8, true);
        if (at_end && !__any(want)) break;
        if (want) row_lane_compress(c.st8, c.buf, at_end);
    }
    u64* out = a.digests + i * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = c.st8.h[j];
}

// Salts can also be made on the device: 64-byte block j of the stream is BLAKE2b-512(seed || j), seed = 32 bytes of os.urandom.
// (The reference draws urandom(24) per leaf, salted_merkle.py:25; any cryptographic stream serves.  Tests that need the
// reference's exact bytes pass host salts instead.)
__global__ void random_fill_kernel(u64 s0, u64 s1, u64 s2, u64 s3, u64* out, u64 nblocks) {
    const u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nblocks) return;
    u64 m[16] = {s0, s1, s2, s3, j, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    u64 h[8];
    blake2b_init(h);
    blake2b_compress(h, m, 40, true);
#pragma unroll
    for (int k = 0; k < 8; ++k) out[8 * j + k] = h[k];
}

// ExtensionField.sample (extension_field.py:100-111) of 27 pseudo-random bytes per element, on the device.  The byte stream is the first
// 63 bytes of every 64-byte block BLAKE2b-512(seed || b), b = 0, 1, ...; element i takes stream bytes [27 i, 27 i + 27): limb j is the
// big-endian integer of its j-th run of 9 bytes, reduced mod p (2^64 = 2^32 - 1) -- seven limbs per compression.
__global__ void xfe_sample_kernel(u64 s0, u64 s1, u64 s2, u64 s3, u64* out, u64 count, u64 stride) {
    const u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (7 * b >= 3 * count) return;
    u64 m[16] = {s0, s1, s2, s3, b, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    u64 h[8];
    blake2b_init(h);
    blake2b_compress(h, m, 40, true);
#pragma unroll
    for (u32 k = 0; k < 7; ++k) {
        const u64 limb = 7 * b + k;
        if (limb >= 3 * count) break;
        // bytes 9 k .. 9 k + 8 of the block: the first is the most significant
        const u32 at = 9 * k, w = (at + 1) / 8, sh = 8 * ((at + 1) % 8);
        const u64 top = (h[at / 8] >> (8 * (at % 8))) & 0xFF;
        const u64 low = __builtin_bswap64(sh ? (h[w] >> sh) | (h[w + 1] << (64 - sh)) : h[w]);      // bytes 1..8 as a big-endian integer
        const u64 v = gl_add(gl_mul(top, GL_EPS), low >= GL_P ? low - GL_P : low);
        out[(limb % 3) * stride + limb / 3] = v;
    }
}

// ---- host: one template per pattern -------------------------------------------------------------------------------
struct HostTemplates {
    std::vector<RowTemplate> templates;
    std::vector<RowStep> steps;      // what the kernel reads
    std::vector<u32> ints;           // idem
    std::vector<RowSeg> segs;        // scratch of build_template
    std::vector<u64> pool;           // idem
    int generated = -1;              // the layout of rows_generated.hpp one of the templates is (by hash), or -1 ...
    u32 generated_variant = 0;       // ... which of its patterns ...
    u32 generated_code = 0;          // ... and the pattern itself
    std::vector<u64> block0_states;  // ... and BLAKE2b's state behind block 0 of its rows, per length of the row's integers (RowArgs::block0_states)
};

static std::mutex g_template_mu;
static std::atomic<u64> g_generated_launches{0};      // how often a kernel of rows_generated.hpp was used (the tests ask)
static std::map<std::string, std::shared_ptr<const HostTemplates>> g_template_cache;
static std::map<std::string, std::vector<u32>> g_last_codes;      // column layout -> the row patterns its last commitment had

static void add_const(HostTemplates& ht, const std::string& bytes, size_t from, size_t to) {
    if (to <= from) return;
    RowSeg sg{SEG_CONST, (u32)ht.pool.size(), (u32)(to - from), 0};
    const size_t words = (to - from + 7) / 8;
    const size_t base = ht.pool.size();
    ht.pool.resize(base + words, 0);
    memcpy(&ht.pool[base], bytes.data() + from, to - from);
    ht.segs.push_back(sg);
}

// sentinel for (column, limb): above 2^63 so that it is written as a 11-byte LONG1, never equal to a constant of the skeleton
static u64 sentinel(u32 column, u32 limb) { return 0xE000000000000000ULL | ((u64)column << 8) | limb | 0x5A5A0000ULL; }

static int build_template(const bfs_row_column* cols, u32 ncols, u32 code, bool salted, HostTemplates& ht) {
    rp::World world;
    rp::Pickler pickler(&world);
    pickler.record_ints = true;
    std::vector<rp::Ref> items(ncols);
    std::map<u64, std::pair<u32, u32>> slot_of;
    u32 shift = 0;
    for (u32 c = 0; c < ncols; ++c) {
        if (cols[c].is_ext) {
            const u32 k = (code >> shift) & 3u;
            shift += 2;
            u64 l[3] = {0, 0, 0};
            for (u32 j = 0; j < k; ++j) { l[j] = sentinel(c, j); slot_of[l[j]] = {c, j}; }
            items[c] = world.xfe_compact(l);
        } else {
            const u64 v = sentinel(c, 0);
            slot_of[v] = {c, 0};
            items[c] = world.bfe_in(v, world.base_field(cols[c].field_id));
        }
    }
    const std::string s = pickler.dumps(rp::mk_tuple(items));
    if (s.size() < 11 || (unsigned char)s[2] != 0x95) { set_error("row pickle without a frame"); return BFS_ERR_BAD_ARG; }
    RowTemplate t{};
    t.code = code;
    const size_t first_seg = ht.segs.size();
    size_t pos = 0;
    u32 const_bytes = 0;
    auto flush = [&](size_t to) { add_const(ht, s, pos, to); const_bytes += (u32)(to - pos); pos = to; };
    flush(3);                                          // PROTO 4, FRAME opcode
    ht.segs.push_back(RowSeg{SEG_FRAMELEN, 0, 0, 0});
    pos = 11;
    const_bytes += 8;
    for (const auto& m : pickler.int_marks) {
        auto it = slot_of.find(m.value);
        if (it == slot_of.end()) continue;             // a constant of the skeleton (the modulus, p)
        flush(m.offset);
        ht.segs.push_back(RowSeg{SEG_INT, it->second.first, it->second.second, 0});
        pos = m.offset + m.length;
    }
    flush(s.size());
    t.tuple_const_bytes = const_bytes;
    if (salted) {
        unsigned char salt[24];
        for (int j = 0; j < 24; ++j) salt[j] = (unsigned char)(0xA0 + j);
        const std::string ss = pickler.dumps(rp::mk_bytes(salt, 24));
        const size_t at = ss.find(std::string((const char*)salt, 24));
        if (at == std::string::npos) { set_error("salt pickle layout"); return BFS_ERR_BAD_ARG; }
        add_const(ht, ss, 0, at);
        ht.segs.push_back(RowSeg{SEG_SALT, 0, 0, 0});
        add_const(ht, ss, at + 24, ss.size());
        t.salt_bytes = (u32)ss.size();
    }
    // flatten: one step per (at most) 8 bytes of the preimage, constants carried in the step itself
    t.first_step = (u32)ht.steps.size();
    t.first_int = (u32)ht.ints.size();
    for (size_t k = first_seg; k < ht.segs.size(); ++k)
        if (ht.segs[k].kind == SEG_INT) ht.ints.push_back(ht.segs[k].a | (ht.segs[k].b << 8));
    t.num_ints = (u32)ht.ints.size() - t.first_int;
    u32 int_index = 0;
    for (size_t k = first_seg; k < ht.segs.size(); ++k) {
        const RowSeg& sg = ht.segs[k];
        if (sg.kind == SEG_CONST) {
            for (u32 w = 0; 8 * w < sg.b; ++w) ht.steps.push_back(RowStep{SEG_CONST, std::min<u32>(8u, sg.b - 8 * w), ht.pool[sg.a + w]});
        } else if (sg.kind == SEG_INT) {
            const u32 pf = ht.ints[t.first_int + std::min(int_index + ROW_PREFETCH, t.num_ints - 1)];    // what to request at this integer
            ++int_index;
            ht.steps.push_back(RowStep{SEG_INT, sg.a, (u64)sg.b | ((u64)pf << 8) | ((u64)((int_index - 1) % ROW_PREFETCH) << 24)});
            ht.steps.push_back(RowStep{SEG_INT_HI, sg.a, sg.b});
        } else if (sg.kind == SEG_FRAMELEN) {
            ht.steps.push_back(RowStep{SEG_FRAMELEN, 0, 0});
        } else {
            for (u32 w = 0; w < 3; ++w) ht.steps.push_back(RowStep{SEG_SALT, w, 0});
        }
    }
    while ((ht.steps.size() - t.first_step) % ROW_STEP_PAD) ht.steps.push_back(RowStep{SEG_CONST, 0, 0});   // the kernel takes several steps per turn
    t.num_steps = (u32)ht.steps.size() - t.first_step;
    ht.templates.push_back(t);
    return BFS_OK;
}

// What a template IS, as one number: FNV-1a over its steps, the integers they name and its two lengths.  A generated leaf kernel
// (rows_generated.hpp) carries the hash of the template it was unrolled from and is used for exactly the templates that hash alike.
static u64 template_hash(const HostTemplates& ht, const RowTemplate& t) {
    u64 h = 0xcbf29ce484222325ull;
    auto mix = [&h](u64 v) { for (int b = 0; b < 8; ++b) { h ^= (v >> (8 * b)) & 0xFF; h *= 0x100000001b3ull; } };
    mix(t.num_steps); mix(t.num_ints); mix(t.tuple_const_bytes); mix(t.salt_bytes);
    for (u32 k = 0; k < t.num_steps; ++k) { const RowStep& st = ht.steps[t.first_step + k]; mix(st.kind | ((u64)st.a << 32)); mix(st.data); }
    for (u32 k = 0; k < t.num_ints; ++k) mix(ht.ints[t.first_int + k]);
    return h;
}

template <class G>
static bool generated_match(HostTemplates& ht, int layout) {
    for (const RowTemplate& t : ht.templates) {
        const u64 h = template_hash(ht, t);
        for (u32 v = 0; v < G::NUM_VARIANTS; ++v)
            if (h == G::HASHES[v] && t.code == G::CODES[v]) {
                ht.generated = layout; ht.generated_variant = v; ht.generated_code = t.code;
                // every length the row's integers can have together: 2 .. 11 bytes each (leaf_encode.hpp: pickle_int_len)
                const u32 n = G::NUM_INTS[v];
                ht.block0_states.resize((size_t)(9 * n + 1) * 8);
                for (u32 extra = 0; extra <= 9 * n; ++extra) {
                    const u64 tuple_len = (u64)G::TUPLE_CONST_BYTES[v] + 2 * n + extra;
                    unsigned char block[128];
                    memcpy(block, G::BLOCK0, 128);
                    const u64 frame = tuple_len - 11;                       // (as SEG_FRAMELEN writes it: rows_core.hpp)
                    memcpy(block + 3, &frame, 8);
                    u64 m[16], st[8];
                    memcpy(m, block, 128);
                    blake2b_init(st);
                    blake2b_compress(st, m, 128, false);
                    memcpy(&ht.block0_states[(size_t)extra * 8], st, 64);
                }
                return true;
            }
    }
    return false;
}
// one of the templates is a pattern of a layout rows_generated.hpp has a kernel for? (the first that is)
static void find_generated(HostTemplates& ht) {
    static_assert(rowgen::NUM_LAYOUTS == 2, "one line per generated layout");
    if (generated_match<rowgen::Layout0>(ht, 0)) return;
    generated_match<rowgen::Layout1>(ht, 1);
}

}  // namespace bfs

using namespace bfs;

extern "C" uint64_t bfs_row_generated_launches(void) { return g_generated_launches.load(std::memory_order_relaxed); }

// tools/gen_rows.py and the tests: the flattened template of one row pattern of a column layout (d_values are not looked at).
// out_header = {num_steps, num_ints, tuple_const_bytes, salt_bytes}; a step is two words: kind | a << 32, data.
extern "C" int bfs_row_template_steps(const bfs_row_column* columns, uint32_t ncols, uint32_t code, int salted, uint32_t out_header[4],
                                      uint64_t* out_steps, uint32_t steps_cap, uint32_t* out_ints, uint32_t ints_cap, uint64_t* out_hash) {
    if (ncols == 0 || ncols > ROW_MAX_COLS) { set_error("bfs_row_template_steps: 1..%d columns", ROW_MAX_COLS); return BFS_ERR_BAD_ARG; }
    HostTemplates ht;
    BFS_TRY(build_template(columns, ncols, code, salted != 0, ht));
    const RowTemplate& t = ht.templates[0];
    if (t.num_steps > steps_cap || t.num_ints > ints_cap) { set_error("bfs_row_template_steps: %u steps, %u integers do not fit", t.num_steps, t.num_ints); return BFS_ERR_BAD_ARG; }
    out_header[0] = t.num_steps; out_header[1] = t.num_ints; out_header[2] = t.tuple_const_bytes; out_header[3] = t.salt_bytes;
    for (u32 k = 0; k < t.num_steps; ++k) { out_steps[2 * k] = ht.steps[k].kind | ((u64)ht.steps[k].a << 32); out_steps[2 * k + 1] = ht.steps[k].data; }
    for (u32 k = 0; k < t.num_ints; ++k) out_ints[k] = ht.ints[k];
    if (out_hash) *out_hash = template_hash(ht, t);
    return BFS_OK;
}

extern "C" int bfs_random_fill(const uint8_t seed[32], uint64_t* d_out, uint64_t nwords, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (nwords % 8) { set_error("bfs_random_fill: the word count must be a multiple of 8"); return BFS_ERR_BAD_ARG; }
    if (nwords == 0) return BFS_OK;
    u64 s[4];
    memcpy(s, seed, 32);
    const u64 nblocks = nwords / 8;
    hipLaunchKernelGGL(random_fill_kernel, dim3((u32)((nblocks + 255) / 256)), dim3(256), 0, stream, s[0], s[1], s[2], s[3], d_out, nblocks);
    BFS_HIP(hipGetLastError());
    return BFS_OK;
}

extern "C" int bfs_xfe_sample_fill(const uint8_t seed[32], uint64_t* d_out, uint64_t count, uint64_t limb_stride, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (count == 0) return BFS_OK;
    u64 s[4];
    memcpy(s, seed, 32);
    const u64 blocks = (3 * count + 6) / 7;
    hipLaunchKernelGGL(xfe_sample_kernel, dim3((u32)((blocks + 255) / 256)), dim3(256), 0, stream, s[0], s[1], s[2], s[3], d_out, count, limb_stride);
    BFS_HIP(hipGetLastError());
    return BFS_OK;
}

static int build_rows(const bfs_row_column* columns, uint32_t ncols, uint64_t n, uint64_t limb_stride, const uint8_t* salts, int salts_on_device,
                      uint8_t* d_nodes, uint8_t* h_root, void* stream_);

extern "C" int bfs_merkle_build_rows(const bfs_row_column* columns, uint32_t ncols, uint64_t n, const uint8_t* salts, int salts_on_device,
                                     uint8_t* d_nodes, void* stream_) {
    return build_rows(columns, ncols, n, n, salts, salts_on_device, d_nodes, nullptr, stream_);
}

extern "C" int bfs_merkle_build_rows_range(const bfs_row_column* columns, uint32_t ncols, uint64_t n, uint64_t limb_stride, const uint8_t* salts,
                                           int salts_on_device, uint8_t* d_nodes, void* stream_) {
    return build_rows(columns, ncols, n, limb_stride, salts, salts_on_device, d_nodes, nullptr, stream_);
}

extern "C" int bfs_merkle_build_rows_root(const bfs_row_column* columns, uint32_t ncols, uint64_t n, uint64_t limb_stride, const uint8_t* salts,
                                          int salts_on_device, uint8_t* d_nodes, uint8_t h_root[64], void* stream_) {
    return build_rows(columns, ncols, n, limb_stride, salts, salts_on_device, d_nodes, h_root, stream_);
}

// A small proof spends as long around the leaf kernel as in it, so the host traffic is packed: the column description, the cleared
// error words and the empty pattern set go up in ONE copy from a pinned block, the templates in another, and the error word comes back
// together with the root (h_root given: the caller's next step is root(), brainfuck_stark.py:179).
static int build_rows(const bfs_row_column* columns, uint32_t ncols, uint64_t n, uint64_t limb_stride, const uint8_t* salts, int salts_on_device,
                      uint8_t* d_nodes, uint8_t* h_root, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (limb_stride < n) { set_error("bfs_merkle_build_rows_range: limb_stride < n"); return BFS_ERR_BAD_ARG; }
    const uint8_t* h_salts = salts_on_device ? nullptr : salts;
    const bool salted = salts != nullptr;
    if (n == 0) return BFS_OK;
    if (((uintptr_t)d_nodes & 15) != 0) { set_error("d_nodes must be 16-byte aligned"); return BFS_ERR_BAD_ARG; }
    if (ncols == 0 || ncols > ROW_MAX_COLS) { set_error("bfs_merkle_build_rows: 1..%d columns", ROW_MAX_COLS); return BFS_ERR_BAD_ARG; }
    u32 next = 0;
    for (u32 c = 0; c < ncols; ++c) next += columns[c].is_ext ? 1 : 0;
    if (next > 16) { set_error("bfs_merkle_build_rows: at most 16 extension columns"); return BFS_ERR_BAD_ARG; }
    u32 depth = 0;
    while ((1ull << depth) < n) ++depth;
    const u64 npo2 = 1ull << depth;

    // device-side description of the columns, scratch for patterns / salts / flags
    const size_t salt_words = h_salts ? (size_t)3 * n : 0;      // staging only for host salts
    const size_t set_bytes = PATTERN_SLOTS * sizeof(u64);
    const size_t cols_bytes = (ncols * sizeof(u64*) + 15) & ~(size_t)15, ext_bytes = (ncols * sizeof(u32) + 15) & ~(size_t)15;
    const size_t head_bytes = cols_bytes + ext_bytes + 16 + set_bytes;       // what the one upload carries
    const size_t fixed_bytes = head_bytes + salt_words * sizeof(u64) + 64;
    void* w = nullptr;
    BFS_TRY(workspace(5, fixed_bytes, stream, &w));
    char* base = (char*)w;
    const u64** d_cols = (const u64**)base;                     base += cols_bytes;
    u32* d_ext = (u32*)base;                                    base += ext_bytes;
    u32* d_err = (u32*)base;                                    base += 16;
    u64* d_set = (u64*)base;                                    base += set_bytes;
    u64* d_salts = (u64*)base;
    PinnedLease stage;
    BFS_TRY(stage.get(head_bytes + 96));
    {
        char* h = (char*)stage.host;
        memset(h, 0, cols_bytes + ext_bytes + 16);
        for (u32 c = 0; c < ncols; ++c) {
            ((const u64**)h)[c] = columns[c].d_values;
            ((u32*)(h + cols_bytes))[c] = columns[c].is_ext ? 1u : 0u;
        }
        memset(h + cols_bytes + ext_bytes + 16, 0xFF, set_bytes);
    }
    BFS_HIP(hipMemcpyAsync(w, stage.host, head_bytes, hipMemcpyHostToDevice, stream));
    if (h_salts) BFS_TRY(copy_h2d(d_salts, h_salts, salt_words * sizeof(u64), stream));   // test mode: the reference's byte stream

    RowArgs a{};
    a.columns = d_cols; a.is_ext = d_ext; a.ncols = ncols; a.n = n; a.limb_stride = limb_stride;
    for (u32 c = 0; c < ncols; ++c)
        if (columns[c].is_ext) { a.ext_cols[a.n_ext / 4] |= c << (8 * (a.n_ext % 4)); ++a.n_ext; }
    a.salts = h_salts ? d_salts : (salted ? (const u64*)salts : nullptr);
    a.digests = (u64*)d_nodes + npo2 * 8;
    a.pattern_set = d_set; a.error = d_err;

    // Which row patterns occur is a property of the data, but the same few come back proof after proof (a pattern is the degree class of
    // every extension element of the row: "all of degree 2" and a handful around all-zero columns).  So the set found for this column
    // layout LAST time is tried first: no pattern kernel, no read-back in the middle of the call.  The leaf kernel checks every row's
    // pattern against the templates it was given and raises the error word for a row without one; then -- and the first time a layout
    // is seen -- the patterns are collected (row_pattern_kernel) and the leaves hashed again.  BFS_ROWS_SPECULATE=0 always collects.
    std::string layout((const char*)&ncols, sizeof ncols);
    for (u32 c = 0; c < ncols; ++c) { const int32_t d[2] = {columns[c].is_ext ? 1 : 0, columns[c].field_id}; layout.append((const char*)d, sizeof d); }
    layout.push_back(salted ? 1 : 0);
    static const bool speculate = [] { const char* e = getenv("BFS_ROWS_SPECULATE"); return !(e && e[0] == '0'); }();
    const u64* set = (const u64*)((char*)stage.host + cols_bytes + ext_bytes);          // (the pinned block takes the answers too)
    char* back = (char*)stage.host + head_bytes;      // (behind the head block, which may have to go up a second time)
    for (int attempt = 0; attempt < 2; ++attempt) {
        std::vector<u32> codes;
        bool guessed = false;
        if (attempt == 0 && speculate) {
            std::lock_guard<std::mutex> lock(g_template_mu);
            auto it = g_last_codes.find(layout);
            if (it != g_last_codes.end()) { codes = it->second; guessed = true; }
        }
        if (!guessed) {
            // 1. patterns present in this batch of rows: only the (small) set comes back -- a pageable n-word copy would be pinned
            // and unpinned by the runtime, and the unmapping stalls the next dispatches for ~25 ms (profiles/r01/README.md)
            if (attempt == 1) BFS_HIP(hipMemcpyAsync(w, stage.host, head_bytes, hipMemcpyHostToDevice, stream));     // cleared error words, empty set
            hipLaunchKernelGGL(row_pattern_kernel, dim3((u32)((n + 255) / 256)), dim3(256), 0, stream, a);
            BFS_HIP(hipGetLastError());
            BFS_HIP(hipMemcpyAsync((void*)set, d_err, 16 + set_bytes, hipMemcpyDeviceToHost, stream));   // error words sit right before the set
            BFS_HIP(hipStreamSynchronize(stream));
            if (((const u32*)set)[1]) { set_error("bfs_merkle_build_rows: more than %u distinct row patterns", PATTERN_SLOTS); return BFS_ERR_BAD_ARG; }
            for (u32 k = 0; k < PATTERN_SLOTS; ++k)
                if (set[2 + k] != ~0ull) codes.push_back((u32)set[2 + k]);
            std::sort(codes.begin(), codes.end());
            std::lock_guard<std::mutex> lock(g_template_mu);
            if (g_last_codes.size() >= 256) g_last_codes.clear();
            g_last_codes[layout] = codes;
        }

        // 2. one template per pattern.  Making a template means pickling a row of sentinels -- ~30 us each: the finished sets are
        // remembered per (column layout, patterns).
        std::string key = layout;
        key.append((const char*)codes.data(), codes.size() * sizeof(u32));
        std::shared_ptr<const HostTemplates> cached;
        {
            std::lock_guard<std::mutex> lock(g_template_mu);
            auto it = g_template_cache.find(key);
            if (it != g_template_cache.end()) cached = it->second;
        }
        if (!cached) {
            std::shared_ptr<HostTemplates> fresh(new HostTemplates());
            for (u32 code : codes) BFS_TRY(build_template(columns, ncols, code, salted, *fresh));
            find_generated(*fresh);
            std::lock_guard<std::mutex> lock(g_template_mu);
            if (g_template_cache.size() >= 256) g_template_cache.clear();
            g_template_cache[key] = fresh;
            cached = fresh;
        }
        const HostTemplates& ht = *cached;
        const size_t tbytes = ht.templates.size() * sizeof(RowTemplate), sbytes = ht.steps.size() * sizeof(RowStep);
        const size_t ibytes = (ht.ints.size() * sizeof(u32) + 7) & ~(size_t)7, mbytes = ht.block0_states.size() * sizeof(u64);
        void* tw = nullptr;
        BFS_TRY(workspace(6, tbytes + sbytes + ibytes + mbytes + 64, stream, &tw));
        char* tb = (char*)tw;
        PinnedLease tstage;
        BFS_TRY(tstage.get(tbytes + sbytes + ibytes + mbytes + 64));
        memcpy(tstage.host, ht.templates.data(), tbytes);
        memcpy((char*)tstage.host + tbytes, ht.steps.data(), sbytes);
        memcpy((char*)tstage.host + tbytes + sbytes, ht.ints.data(), ht.ints.size() * sizeof(u32));
        if (mbytes) memcpy((char*)tstage.host + tbytes + sbytes + ibytes, ht.block0_states.data(), mbytes);
        BFS_HIP(hipMemcpyAsync(tb, tstage.host, tbytes + sbytes + ibytes + mbytes, hipMemcpyHostToDevice, stream));
        a.block0_states = (const u64*)(tb + tbytes + sbytes + ibytes);
        a.templates = (const RowTemplate*)tb;
        a.num_templates = (u32)ht.templates.size();
        a.steps = (const RowStep*)(tb + tbytes);
        a.ints = (const u32*)(tb + tbytes + sbytes);

        // 3. leaf digests, then the tree.  A layout rows_generated.hpp knows goes through its straight-line kernel; rows of any other
        // pattern of that layout (and every other layout) through the interpreter.  BFS_ROWS_GENERATED=0: the interpreter only.
        static const bool use_generated = [] { const char* e = getenv("BFS_ROWS_GENERATED"); return !(e && e[0] == '0'); }();
        const dim3 grid((u32)((n + LEAF_THREADS - 1) / LEAF_THREADS)), block(LEAF_THREADS);
        const int gen = use_generated ? ht.generated : -1;
        static const bool log_rows = [] { const char* e = getenv("BFS_ROWS_LOG"); return e && e[0] == '1'; }();
        if (log_rows) {
            std::string line;
            for (const RowTemplate& t : ht.templates) { char b[16]; snprintf(b, sizeof b, " %x", t.code); line += b; }
            fprintf(stderr, "[bfs] row leaves: %u columns (%u extension), %llu rows, patterns%s%s, generated layout %d\n", ncols, a.n_ext,
                    (unsigned long long)n, line.c_str(), guessed ? " (remembered)" : "", gen);
        }
        const u32 others = ht.templates.size() > 1 ? 1u : 0u;
        a.skip = 0;
        if (gen == 0) hipLaunchKernelGGL(row_leaves_generated_kernel<rowgen::Layout0>, grid, block, 0, stream, a, ht.generated_variant, others);
        else if (gen == 1) hipLaunchKernelGGL(row_leaves_generated_kernel<rowgen::Layout1>, grid, block, 0, stream, a, ht.generated_variant, others);
        static_assert(rowgen::NUM_LAYOUTS == 2, "one launch per generated layout");
        if (gen >= 0) {
            BFS_HIP(hipGetLastError());
            g_generated_launches.fetch_add(1, std::memory_order_relaxed);
            a.skip = 1;
            a.skip_code = ht.generated_code;
        }
        if (gen < 0 || others) {
            hipLaunchKernelGGL(row_leaves_kernel, grid, block, 0, stream, a);
            BFS_HIP(hipGetLastError());
        }
        BFS_TRY(merkle_inner_launch((u64*)d_nodes, depth, n, stream, nullptr, 0));
        // error word and root in one copy: nodes[1] is the root (heap order), digests of 8 words
        BFS_HIP(hipMemcpyAsync(back, d_err, 16, hipMemcpyDeviceToHost, stream));
        if (h_root != nullptr) BFS_HIP(hipMemcpyAsync(back + 16, (const u64*)d_nodes + 8, 64, hipMemcpyDeviceToHost, stream));
        BFS_HIP(hipStreamSynchronize(stream));       // (tstage is released after this: the template copy has been consumed)
        const u32 err = *(const u32*)back;
        if (!err) {
            if (h_root != nullptr) memcpy(h_root, back + 16, 64);
            return BFS_OK;
        }
        if (!guessed) break;       // the patterns were collected from these very rows: a missing template is a bug
        // a row with a pattern the remembered set does not have: collect and hash again (the head block -- column description, zeroed
        // error words, empty set -- is still in the pinned stage; it goes up again in front of the pattern kernel)
        memset((char*)stage.host + cols_bytes + ext_bytes, 0, 16);
        memset((char*)stage.host + cols_bytes + ext_bytes + 16, 0xFF, set_bytes);
    }
    set_error("bfs_merkle_build_rows: a row pattern without a template (internal)");
    return BFS_ERR_BAD_ARG;
}
