// merkle.hip -- BLAKE2b-512 Merkle trees on gfx950.  Replaces Merkle.__init__ of the reference
// (/root/reference/code/merkle.py:8-41): leaf hashing `blake2b(pickle.dumps(leaf))` (:29-32) and the level-by-level
// parent hashing (:35-41).  One thread hashes one leaf (preimage synthesised in LDS, <= 4 compressions) or one
// parent (one compression); the top 9 levels run in a single workgroup.
#include "blake2b_quad.hpp"
#include "merkle_core.hpp"
#include <vector>

#include "runtime.hpp"

namespace bfs {

constexpr int LEAF_THREADS = 64;  // one wavefront per workgroup

// One wave hashes 64 leaves whose limbs it already holds (c0, c1, c2; `active` = the lane has a leaf) in the streaming form of
// merkle_core.hpp: 20 words of LDS per lane = 10 KiB per wave instead of 18 (the staged form of rounds 1-2 -- whole tail in LDS, then
// hashed -- was the slower side of profiles/r03/ab_leaf_streaming.txt and is gone).
constexpr int LEAF_STAGE_WORDS = XFE_STREAM_WORDS * LEAF_THREADS;

__device__ __forceinline__ void xfe_leaves_wave(u64 c0, u64 c1, u64 c2, bool active, u64* stage /* LEAF_STAGE_WORDS */, u64 h[8], const u64* midstates) {
    const u32 lane = threadIdx.x;
    const u32 k = active ? xfe_leaf_k(c0, c1, c2) : 0u;
    if (active && k == 0) merkle_leaf_xfe_zero(h, midstates);
    // class by class (wave-uniform branches): a wave of random extension elements, or of lifted base-field elements, takes exactly one
    // of the three; a mixed wave takes its classes in turn with the other lanes idle
    if (__ballot(k == 3) != 0) { if (k == 3) merkle_leaf_xfe_stream<3>(c0, c1, c2, stage + lane, LEAF_THREADS, h, midstates); }
    if (__ballot(k == 2) != 0) { if (k == 2) merkle_leaf_xfe_stream<2>(c0, c1, c2, stage + lane, LEAF_THREADS, h, midstates); }
    if (__ballot(k == 1) != 0) { if (k == 1) merkle_leaf_xfe_stream<1>(c0, c1, c2, stage + lane, LEAF_THREADS, h, midstates); }
}

__global__ void __launch_bounds__(LEAF_THREADS) merkle_leaves_xfe_kernel(const u64* limbs, u64 limb_stride, u64 n, u64* leaf_digests, const u64* midstates) {
    __shared__ u64 stage[LEAF_STAGE_WORDS];
    const u64 i = (u64)blockIdx.x * LEAF_THREADS + threadIdx.x;
    const bool active = i < n;
    u64 c0 = 0, c1 = 0, c2 = 0;
    if (active) { c0 = limbs[i]; c1 = limbs[limb_stride + i]; c2 = limbs[2 * limb_stride + i]; }
    u64 d[8];
    xfe_leaves_wave(c0, c1, c2, active, stage, d, midstates);
    if (!active) return;
    u64* out = leaf_digests + i * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = d[j];
}

// the same for a FRI round whose codeword does not exist yet: every thread first PRODUCES its element (the split-and-fold step of the
// previous round, fri.py:127-128), stores it for the later openings and hashes it -- one launch and one pass over the codeword less
__global__ void __launch_bounds__(LEAF_THREADS) merkle_leaves_xfe_fold_kernel(FriFoldArgs f, u64* cw, u64 cw_stride, u64 n, u64* leaf_digests, const u64* midstates) {
    __shared__ u64 stage[LEAF_STAGE_WORDS];
    const u64 i = (u64)blockIdx.x * LEAF_THREADS + threadIdx.x;
    const bool active = i < n;
    u64 c0 = 0, c1 = 0, c2 = 0;
    if (active) {
        const Xfe a{{f.in[i], f.in[f.in_stride + i], f.in[2 * f.in_stride + i]}};
        const Xfe b{{f.in[f.half + i], f.in[f.in_stride + f.half + i], f.in[2 * f.in_stride + f.half + i]}};
        const u64 sc = gl_mul(f.scal, tw_pow(f.winv_lo, f.winv_hi, f.lo_bits, i << f.round_shift));
        const Xfe beta = xfe_scale(f.alpha, sc);
        const Xfe sum = xfe_add(a, b), diff = xfe_sub(a, b);
        const Xfe prod = xfe_mul(beta, diff);
        const u64 x = (sum.c[0] >> 1) + ((sum.c[0] & 1) ? 0x7FFFFFFF80000001ULL : 0);      // / 2 mod p
        const u64 y = (sum.c[1] >> 1) + ((sum.c[1] & 1) ? 0x7FFFFFFF80000001ULL : 0);
        const u64 z = (sum.c[2] >> 1) + ((sum.c[2] & 1) ? 0x7FFFFFFF80000001ULL : 0);
        c0 = gl_add(x, prod.c[0]);
        c1 = gl_add(y, prod.c[1]);
        c2 = gl_add(z, prod.c[2]);
        cw[i] = c0;
        cw[cw_stride + i] = c1;
        cw[2 * cw_stride + i] = c2;
    }
    u64 d[8];
    xfe_leaves_wave(c0, c1, c2, active, stage, d, midstates);
    if (!active) return;
    u64* out = leaf_digests + i * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = d[j];
}

__global__ void __launch_bounds__(LEAF_THREADS) merkle_leaves_bfe_kernel(const u64* values, u64 n, u64* leaf_digests) {
    __shared__ u64 stage[BFE_LEAF_MAX_WORDS * LEAF_THREADS];
    const u64 i = (u64)blockIdx.x * LEAF_THREADS + threadIdx.x;
    if (i >= n) return;
    u64 d[8];
    merkle_leaf_bfe_body(values, i, stage + threadIdx.x, LEAF_THREADS, d);
    u64* out = leaf_digests + i * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = d[j];
}

// generic byte strings: message i occupies words [offsets[i], offsets[i] + ceil(len/8)) of `data`, lengths[i] bytes
__global__ void blake2b_batch_kernel(const u64* data, const u64* offsets, const u32* lengths, u64 n, u64* digests) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 h[8];
    blake2b_staged(data + offsets[i], 1, lengths[i], h);
    u64* out = digests + i * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = h[j];
}

// one level: parents [first, first+count) from children [2*first, ...).  present_children counts the child slots
// (from the start of the child level) that hold digests; only smaller than 2*count directly above a ragged leaf level.
__global__ void merkle_parents_kernel(u64* nodes, u64 first, u64 count, u64 present_children) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const u64 k = first + t;
    const u64 c = 2 * t;
    int present = c + 1 < present_children ? 2 : (c < present_children ? 1 : 0);
    u64 out[8];
    merkle_parent_body(nodes + (2 * k) * 8, nodes + (2 * k + 1) * 8, present, out);
    u64* dst = nodes + k * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[j] = out[j];
}

// the top of the tree in one workgroup: levels with `width` <= 256 parents down to the root, through LDS
// root_out (optional): pinned, host-visible memory; the root is written there followed by a sequence flag (word 8), so the
// host can pick it up without a copy command (FRI needs the root of every round on the host, fri.py:108-120)
__global__ void __launch_bounds__(256) merkle_top_kernel(u64* nodes, u32 width, u64 present_children, u64* root_out, u64 seq) {
    __shared__ u64 lvl[512 * 8];
    const u32 t = threadIdx.x;
    for (u32 i = t; i < 2 * width * 8; i += 256) lvl[i] = nodes[(u64)2 * width * 8 + i];
    __syncthreads();
    u64 present = present_children;
    for (u32 w = width; w >= 1; w >>= 1) {
        u64 out[8];
        if (t < w) {
            const u64 c = 2 * (u64)t;
            int pr = c + 1 < present ? 2 : (c < present ? 1 : 0);
            merkle_parent_body(lvl + (2 * t) * 8, lvl + (2 * t + 1) * 8, pr, out);
        }
        __syncthreads();
        if (t < w) {
            u64* dst = nodes + ((u64)w + t) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) { lvl[t * 8 + j] = out[j]; dst[j] = out[j]; }
            if (w == 1 && root_out != nullptr) {
#pragma unroll
                for (int j = 0; j < 8; ++j) __hip_atomic_store(root_out + j, out[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(root_out + 8, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __syncthreads();
        present = 2 * (u64)w;  // every computed level is complete
    }
}

// ---- latency-oriented variants: four lanes per hash (blake2b_quad.hpp), used when a level has few hashes ----

// parents [first, first+count) with one QUAD per parent; 64 parents per 256-thread workgroup, messages staged in LDS
__global__ void __launch_bounds__(256) merkle_parents_quad_kernel(u64* nodes, u64 first, u64 count, u64 present_children) {
#if defined(__HIP_DEVICE_COMPILE__)   // DPP builtins exist only in the device pass
    __shared__ u64 msg[64 * 16];
    const u32 q = threadIdx.x >> 2, j = threadIdx.x & 3;
    const u64 t = (u64)blockIdx.x * 64 + q;
    const QuadLane ql = quad_lane(threadIdx.x);
    if (t < count) {
        const u64 k = first + t, c = 2 * t;
        const int present = c + 1 < present_children ? 2 : (c < present_children ? 1 : 0);
        const u64* child = nodes + (2 * k) * 8;      // the two children are adjacent: 16 words
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int idx = 4 * j + w;
            msg[q * 16 + idx] = (idx < 8 * present) ? child[idx] : 0;
        }
        u64 hl, hh;
        blake2b_init_quad(ql, hl, hh);
        blake2b_compress_quad(ql, hl, hh, msg + q * 16, (u64)(64 * present + 32 * (2 - present)), true);
        nodes[k * 8 + j] = hl;
        nodes[k * 8 + 4 + j] = hh;
    }
#endif
}

// top of the tree (levels of width <= 256 down to the root) in one 1024-thread workgroup, one quad per parent,
// ping-pong level buffers in LDS
__global__ void __launch_bounds__(1024) merkle_top_quad_kernel(u64* nodes, u32 width, u64 present_children, u64* root_out, u64 seq) {
#if defined(__HIP_DEVICE_COMPILE__)   // DPP builtins exist only in the device pass
    __shared__ u64 bufA[512 * 8];
    __shared__ u64 bufB[256 * 8];
    const u32 tid = threadIdx.x, q = tid >> 2, j = tid & 3;
    const QuadLane ql = quad_lane(tid);
    for (u32 i = tid; i < 2 * width * 8; i += 1024) bufA[i] = nodes[(u64)2 * width * 8 + i];
    __syncthreads();
    u64* src = bufA;
    u64* dst = bufB;
    u64 present = present_children;
    for (u32 w = width; w >= 1; w >>= 1) {
        if (q < w) {
            const u64 c = 2 * (u64)q;
            const int pr = c + 1 < present ? 2 : (c < present ? 1 : 0);
            u64* m = src + q * 16;
            if (pr < 2) {                            // ragged leaf level only: absent children are zero bytes
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) { const int idx = 4 * j + wd; if (idx >= 8 * pr) m[idx] = 0; }
            }
            u64 hl, hh;
            blake2b_init_quad(ql, hl, hh);
            blake2b_compress_quad(ql, hl, hh, m, (u64)(64 * pr + 32 * (2 - pr)), true);
            dst[q * 8 + j] = hl;
            dst[q * 8 + 4 + j] = hh;
            u64* g = nodes + ((u64)w + q) * 8;
            g[j] = hl;
            g[4 + j] = hh;
            if (w == 1 && root_out != nullptr) {
                __hip_atomic_store(root_out + j, hl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(root_out + 4 + j, hh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __syncthreads();
        u64* tmp = src; src = dst; dst = tmp;
        present = 2 * (u64)w;
    }
    if (tid == 0 && root_out != nullptr) {
        __threadfence_system();
        __hip_atomic_store(root_out + 8, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#endif
}

// nine levels in one launch: every 1024-thread workgroup takes 512 adjacent digests of a complete level of `children` nodes
// (heap indices children .. 2 children - 1) and builds the subtree above them in LDS, down to its single root at the level of
// children / 512 nodes.  Same dependent chain of hashes as one launch per level (~2.3 us per level with a quad per hash), but
// without a launch boundary and a ramp-up per level: levels of <= 65536 parents are latency-bound, not throughput-bound.
__global__ void __launch_bounds__(1024) merkle_subtree_quad_kernel(u64* nodes, u64 children) {
#if defined(__HIP_DEVICE_COMPILE__)   // DPP builtins exist only in the device pass
    __shared__ u64 bufA[512 * 8];
    __shared__ u64 bufB[256 * 8];
    const u32 tid = threadIdx.x, q = tid >> 2, j = tid & 3;
    const QuadLane ql = quad_lane(tid);
    const u64* in = nodes + (children + (u64)blockIdx.x * 512) * 8;
    for (u32 i = tid; i < 512 * 8; i += 1024) bufA[i] = in[i];
    __syncthreads();
    u64* src = bufA;
    u64* dst = bufB;
    u64 level = children >> 1;                       // nodes of the level being written = heap index of its first node
    for (u32 w = 256; w >= 1; w >>= 1, level >>= 1) {
        if (q < w) {
            u64 hl, hh;
            blake2b_init_quad(ql, hl, hh);
            blake2b_compress_quad(ql, hl, hh, src + q * 16, 128, true);
            dst[q * 8 + j] = hl;
            dst[q * 8 + 4 + j] = hh;
            u64* g = nodes + (level + (u64)blockIdx.x * w + q) * 8;
            g[j] = hl;
            g[4 + j] = hh;
        }
        __syncthreads();
        u64* tmp = src; src = dst; dst = tmp;
    }
#endif
}

// leaves with one quad per leaf (small codewords: late FRI rounds): lane 0 of the quad assembles the preimage in LDS,
// the four lanes hash it
__global__ void __launch_bounds__(256) merkle_leaves_xfe_quad_kernel(const u64* limbs, u64 limb_stride, u64 n, u64* leaf_digests, const u64* midstates) {
#if defined(__HIP_DEVICE_COMPILE__)   // DPP builtins exist only in the device pass
    constexpr int WORDS = 48;                        // 3 blocks of 16 words per leaf (bytes 128..409)
    __shared__ u64 stage[64 * WORDS];
    const u32 q = threadIdx.x >> 2, j = threadIdx.x & 3;
    const u64 i = (u64)blockIdx.x * 64 + q;
    const QuadLane ql = quad_lane(threadIdx.x);
    if (i >= n) return;
    u64* m = stage + q * WORDS;
    const u64 c0 = limbs[i], c1 = limbs[limb_stride + i], c2 = limbs[2 * limb_stride + i];
    const u32 k = xfe_leaf_k(c0, c1, c2);
    u64 hl, hh;
    if (k == 0) {
        hl = midstates[(size_t)2 * LEAF_MS_LEN * 8 + j];
        hh = midstates[(size_t)2 * LEAF_MS_LEN * 8 + 4 + j];
    } else {
        const u32 body = xfe_leaf_body_len(k, c0, c1, c2), total = body + 11;
        const u32 nblk = (total + 127) / 128;
        const u64* ms = midstates + ((size_t)(k == 1 ? 0 : 1) * LEAF_MS_LEN + body) * 8;
        hl = ms[j];
        hh = ms[4 + j];
        if (j == 0) {
            LeafWriter w;
            w.init(m, 1);
            encode_xfe_leaf_tail(w, k, c0, c1, c2);
            for (u32 x = w.wpos; x < (nblk - 1) * 16; ++x) m[x] = 0;   // zero padding of the final block
        }
        for (u32 b = 1; b < nblk; ++b) {
            const bool last = b + 1 == nblk;
            blake2b_compress_quad(ql, hl, hh, m + 16 * (b - 1), last ? (u64)total : (u64)(b + 1) * 128, last);
        }
    }
    leaf_digests[i * 8 + j] = hl;
    leaf_digests[i * 8 + 4 + j] = hh;
#endif
}

// ---- one FRI round on a small codeword in one launch (fri.py:108 + 127-128) ----
// A late FRI round is a chain of dependent hashes -- 3 compressions per leaf, one per tree level, ~2 us each with a quad per
// hash -- and used to be 3-6 launches (fold, leaves, one per level down to 256 parents, top): launch gaps and ramp-up were as
// long as the work.  Here one workgroup takes FRI_WG_LEAVES leaves: it first PRODUCES them (the split-and-fold step of the
// previous round, when `f.in` is set; the folded codeword also goes to HBM for the openings), hashes them with one quad per
// leaf and builds the levels above them in LDS.  A codeword of <= FRI_WG_LEAVES elements is finished by a single workgroup, which
// also drops the root into the host mailbox; larger ones leave one digest per workgroup to merkle_top_quad_kernel.
// n must be a power of two (FRI codewords are).
BFS_HD u64 gl_half_m(u64 x) { return (x >> 1) + ((x & 1) ? 0x7FFFFFFF80000001ULL : 0); }  // x / 2 mod p

// 64 leaves = 64 quads = 4 waves per workgroup: ONE wave per SIMD, so that every compression of the chain runs at the speed of a
// lone wave (~2 us).  With 256 leaves in a 1024-thread workgroup the leaf phase and the two widest levels had 4 and 2 waves per SIMD
// taking turns: a 256-element round took 36 us against 11 compressions x 2.2 us.
constexpr u32 FRI_WG_LEAVES = 64;

__global__ void __launch_bounds__(4 * FRI_WG_LEAVES) fri_round_quad_kernel(FriFoldArgs f, u64* cw, u64 cw_stride, u64 n, u64* nodes, const u64* midstates,
                                                              u64* root_out, u64 seq) {
#if defined(__HIP_DEVICE_COMPILE__)   // DPP builtins exist only in the device pass
    constexpr int WORDS = 48;                        // 3 blocks of 16 words per leaf (bytes 128..409)
    __shared__ u64 stage[FRI_WG_LEAVES * WORDS];
    __shared__ u64 bufA[FRI_WG_LEAVES * 8];
    __shared__ u64 bufB[FRI_WG_LEAVES * 4];
    __shared__ u64 limbs[3 * FRI_WG_LEAVES];
    const u32 tid = threadIdx.x, q = tid >> 2, j = tid & 3;
    const QuadLane ql = quad_lane(tid);
    const u32 local = n < FRI_WG_LEAVES ? (u32)n : FRI_WG_LEAVES;     // leaves of this workgroup
    const u64 first = (u64)blockIdx.x * FRI_WG_LEAVES;
    if (tid < local) {
        const u64 i = first + tid;
        u64 c0, c1, c2;
        if (f.in != nullptr) {
            const Xfe a{{f.in[i], f.in[f.in_stride + i], f.in[2 * f.in_stride + i]}};
            const Xfe b{{f.in[f.half + i], f.in[f.in_stride + f.half + i], f.in[2 * f.in_stride + f.half + i]}};
            const u64 sc = gl_mul(f.scal, tw_pow(f.winv_lo, f.winv_hi, f.lo_bits, i << f.round_shift));
            const Xfe beta = xfe_scale(f.alpha, sc);
            const Xfe sum = xfe_add(a, b), diff = xfe_sub(a, b);
            const Xfe prod = xfe_mul(beta, diff);
            c0 = gl_add(gl_half_m(sum.c[0]), prod.c[0]);
            c1 = gl_add(gl_half_m(sum.c[1]), prod.c[1]);
            c2 = gl_add(gl_half_m(sum.c[2]), prod.c[2]);
            cw[i] = c0; cw[cw_stride + i] = c1; cw[2 * cw_stride + i] = c2;
        } else {
            c0 = cw[i]; c1 = cw[cw_stride + i]; c2 = cw[2 * cw_stride + i];
        }
        limbs[tid] = c0; limbs[FRI_WG_LEAVES + tid] = c1; limbs[2 * FRI_WG_LEAVES + tid] = c2;
    }
    __syncthreads();
    // leaf digests: one quad per leaf (as merkle_leaves_xfe_quad_kernel)
    if (q < local) {
        u64* m = stage + q * WORDS;
        const u64 c0 = limbs[q], c1 = limbs[FRI_WG_LEAVES + q], c2 = limbs[2 * FRI_WG_LEAVES + q];
        const u32 k = xfe_leaf_k(c0, c1, c2);
        u64 hl, hh;
        if (k == 0) {
            hl = midstates[(size_t)2 * LEAF_MS_LEN * 8 + j];
            hh = midstates[(size_t)2 * LEAF_MS_LEN * 8 + 4 + j];
        } else {
            const u32 body = xfe_leaf_body_len(k, c0, c1, c2), total = body + 11;
            const u32 nblk = (total + 127) / 128;
            const u64* ms = midstates + ((size_t)(k == 1 ? 0 : 1) * LEAF_MS_LEN + body) * 8;
            hl = ms[j];
            hh = ms[4 + j];
            if (j == 0) {
                LeafWriter w;
                w.init(m, 1);
                encode_xfe_leaf_tail(w, k, c0, c1, c2);
                for (u32 x = w.wpos; x < (nblk - 1) * 16; ++x) m[x] = 0;   // zero padding of the final block
            }
            for (u32 b = 1; b < nblk; ++b) {
                const bool last = b + 1 == nblk;
                blake2b_compress_quad(ql, hl, hh, m + 16 * (b - 1), last ? (u64)total : (u64)(b + 1) * 128, last);
            }
        }
        u64* g = nodes + (n + first + q) * 8;        // leaf i sits at heap index npo2 + i, npo2 = n
        g[j] = hl; g[4 + j] = hh;
        bufA[q * 8 + j] = hl; bufA[q * 8 + 4 + j] = hh;
    }
    __syncthreads();
    // the levels above this workgroup's leaves, ping-pong in LDS (as merkle_top_quad_kernel)
    u64* src = bufA;
    u64* dst = bufB;
    const u64 groups = n / local;                    // workgroups = subtrees
    const bool whole_tree = groups == 1;
    for (u32 w = local >> 1; w >= 1; w >>= 1) {
        if (q < w) {
            u64 hl, hh;
            blake2b_init_quad(ql, hl, hh);
            blake2b_compress_quad(ql, hl, hh, src + q * 16, 128, true);
            dst[q * 8 + j] = hl; dst[q * 8 + 4 + j] = hh;
            u64* g = nodes + (groups * w + (u64)blockIdx.x * w + q) * 8;   // this level has groups * w nodes, first at that heap index
            g[j] = hl; g[4 + j] = hh;
            if (w == 1 && whole_tree && root_out != nullptr) {
                __hip_atomic_store(root_out + j, hl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(root_out + 4 + j, hh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __syncthreads();
        u64* tmp = src; src = dst; dst = tmp;
    }
    if (tid == 0 && whole_tree && root_out != nullptr) {
        __threadfence_system();
        __hip_atomic_store(root_out + 8, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#endif
}

constexpr u64 QUAD_PARENTS_MAX = 8192;   // levels with at most this many parents use one quad per hash
constexpr u64 SUBTREE_PARENTS_MAX = 65536;    // levels of at most this many parents start a nine-level subtree launch
constexpr u64 QUAD_LEAVES_MAX = 8192;

// build all inner nodes above a leaf level of npo2 = 2^depth slots of which n_leaves hold digests
int merkle_inner_launch(u64* d_nodes, u32 depth, u64 n_leaves, hipStream_t stream, u64* root_out = nullptr, u64 seq = 0) {
    if (depth == 0) return BFS_OK;  // single leaf: root = leaf digest (merkle.py:43 nodes[1])
    u64 present = n_leaves;
    for (u32 lvl = depth; lvl-- > 0;) {
        const u64 count = 1ull << lvl;
        if (present >= 2 * count && count >= 512 && count <= SUBTREE_PARENTS_MAX) {      // (`present` is >= 2 * count for every complete level)
            // a complete level of 2 * count digests: nine levels per launch (merkle_subtree_quad_kernel)
            hipLaunchKernelGGL(merkle_subtree_quad_kernel, dim3((u32)(count / 256)), dim3(1024), 0, stream, d_nodes, 2 * count);
            BFS_HIP(hipGetLastError());
            lvl -= 8;                                // levels lvl .. lvl - 8 are done; the loop goes on below them
            present = 2 * count;
            continue;
        }
        if (count <= 256) {
            hipLaunchKernelGGL(merkle_top_quad_kernel, dim3(1), dim3(1024), 0, stream, d_nodes, (u32)count, present, root_out, seq);
            BFS_HIP(hipGetLastError());
            return BFS_OK;
        }
        if (count <= QUAD_PARENTS_MAX)
            hipLaunchKernelGGL(merkle_parents_quad_kernel, dim3((u32)((count + 63) / 64)), dim3(256), 0, stream, d_nodes, count, count, present);
        else
            hipLaunchKernelGGL(merkle_parents_kernel, dim3((u32)((count + 255) / 256)), dim3(256), 0, stream, d_nodes, count, count, present);
        BFS_HIP(hipGetLastError());
        present = 2 * count;
    }
    return BFS_OK;
}

static int get_leaf_midstates(const u64** d_table) {
    if (cached_table_lookup(0x6D696473ULL, 1, 0, d_table)) return BFS_OK;
    std::vector<u64> host(LEAF_MS_WORDS);
    leaf_midstates(host.data());
    return cached_table(0x6D696473ULL, 1, 0, host.data(), host.size(), d_table);
}

int merkle_build_xfe_launch(const u64* d_limbs, u64 limb_stride, u64 n, u64* d_nodes, hipStream_t stream, u64* root_out, u64 seq) {
    if (n == 0) return BFS_OK;
    const u64* d_ms = nullptr;
    BFS_TRY(get_leaf_midstates(&d_ms));
    u32 depth = 0;
    while ((1ull << depth) < n) ++depth;
    const u64 npo2 = 1ull << depth;
    if (n <= QUAD_LEAVES_MAX)
        hipLaunchKernelGGL(merkle_leaves_xfe_quad_kernel, dim3((u32)((n + 63) / 64)), dim3(256), 0, stream, d_limbs, limb_stride, n, d_nodes + npo2 * 8, d_ms);
    else
        hipLaunchKernelGGL(merkle_leaves_xfe_kernel, dim3((u32)((n + LEAF_THREADS - 1) / LEAF_THREADS)), dim3(LEAF_THREADS), 0, stream,
                           d_limbs, limb_stride, n, d_nodes + npo2 * 8, d_ms);
    BFS_HIP(hipGetLastError());
    return merkle_inner_launch(d_nodes, depth, n, stream, root_out, seq);
}

// Fri.commit's tree over a round's codeword, fused with the fold that produces the codeword (see fri_round_quad_kernel).
// fold.in == nullptr: the codeword is already at d_cw.  Returns BFS_ERR_BAD_ARG for sizes the fused path does not take.
int fri_round_fused_launch(const FriFoldArgs& fold, u64* d_cw, u64 cw_stride, u64 n, u64* d_nodes, hipStream_t stream, u64* root_out, u64 seq) {
    if (n < 2 || n > FRI_FUSED_MAX || (n & (n - 1))) { set_error("internal: fused FRI round on %llu elements", (unsigned long long)n); return BFS_ERR_BAD_ARG; }
    const u64* d_ms = nullptr;
    BFS_TRY(get_leaf_midstates(&d_ms));
    const u32 groups = (u32)(n <= FRI_WG_LEAVES ? 1 : n / FRI_WG_LEAVES);
    hipLaunchKernelGGL(fri_round_quad_kernel, dim3(groups), dim3(4 * FRI_WG_LEAVES), 0, stream, fold, d_cw, cw_stride, n, d_nodes, d_ms, root_out, seq);
    BFS_HIP(hipGetLastError());
    if (groups == 1) return BFS_OK;
    u32 depth = 0;
    while ((1ull << depth) < groups) ++depth;
    return merkle_inner_launch(d_nodes, depth, groups, stream, root_out, seq);     // the level of the subtree roots plays the leaf level
}

// Merkle(codeword) of a FRI round with more than FRI_FUSED_MAX elements, the fold that produces the codeword done by the leaf kernel
int merkle_build_xfe_fold_launch(const FriFoldArgs& fold, u64* d_cw, u64 cw_stride, u64 n, u64* d_nodes, hipStream_t stream, u64* root_out, u64 seq) {
    if (fold.in == nullptr || n <= QUAD_LEAVES_MAX) { set_error("internal: fused fold + leaves on %llu elements", (unsigned long long)n); return BFS_ERR_BAD_ARG; }
    const u64* d_ms = nullptr;
    BFS_TRY(get_leaf_midstates(&d_ms));
    u32 depth = 0;
    while ((1ull << depth) < n) ++depth;
    const u64 npo2 = 1ull << depth;
    hipLaunchKernelGGL(merkle_leaves_xfe_fold_kernel, dim3((u32)((n + LEAF_THREADS - 1) / LEAF_THREADS)), dim3(LEAF_THREADS), 0, stream,
                       fold, d_cw, cw_stride, n, d_nodes + npo2 * 8, d_ms);
    BFS_HIP(hipGetLastError());
    return merkle_inner_launch(d_nodes, depth, n, stream, root_out, seq);
}

int merkle_build_bfe_launch(const u64* d_values, u64 n, u64* d_nodes, hipStream_t stream) {
    if (n == 0) return BFS_OK;
    u32 depth = 0;
    while ((1ull << depth) < n) ++depth;
    const u64 npo2 = 1ull << depth;
    hipLaunchKernelGGL(merkle_leaves_bfe_kernel, dim3((u32)((n + LEAF_THREADS - 1) / LEAF_THREADS)), dim3(LEAF_THREADS), 0, stream,
                       d_values, n, d_nodes + npo2 * 8);
    BFS_HIP(hipGetLastError());
    return merkle_inner_launch(d_nodes, depth, n, stream);
}

int merkle_build_bytes_launch(const u64* d_data, const u64* d_offsets, const u32* d_lengths, u64 n, u64* d_nodes, hipStream_t stream) {
    if (n == 0) return BFS_OK;
    u32 depth = 0;
    while ((1ull << depth) < n) ++depth;
    const u64 npo2 = 1ull << depth;
    hipLaunchKernelGGL(blake2b_batch_kernel, dim3((u32)((n + 63) / 64)), dim3(64), 0, stream, d_data, d_offsets, d_lengths, n, d_nodes + npo2 * 8);
    BFS_HIP(hipGetLastError());
    return merkle_inner_launch(d_nodes, depth, n, stream);
}

}  // namespace bfs
