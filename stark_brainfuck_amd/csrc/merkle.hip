// merkle.hip -- BLAKE2b-512 Merkle trees on gfx950.  Replaces Merkle.__init__ of the reference
// (/root/reference/code/merkle.py:8-41): leaf hashing `blake2b(pickle.dumps(leaf))` (:29-32) and the level-by-level
// parent hashing (:35-41).  One thread hashes one leaf (preimage synthesised in LDS, <= 4 compressions) or one
// parent (one compression); the top 9 levels run in a single workgroup.
#include "blake2b_quad.hpp"
#include "merkle_core.hpp"
#include <vector>

#include "runtime.hpp"

namespace bfs {

constexpr int LEAF_THREADS = 64;  // one wavefront per workgroup: the staging area is 52 words x 64 lanes = 26 KiB

__global__ void __launch_bounds__(LEAF_THREADS) merkle_leaves_xfe_kernel(const u64* limbs, u64 limb_stride, u64 n, u64* leaf_digests, const u64* midstates) {
#ifdef BFS_LEAF_LDS_PAD
    __shared__ u64 stage[(XFE_TAIL_MAX_WORDS + BFS_LEAF_LDS_PAD) * LEAF_THREADS];
#else
    __shared__ u64 stage[XFE_TAIL_MAX_WORDS * LEAF_THREADS];
#endif
    const u64 i = (u64)blockIdx.x * LEAF_THREADS + threadIdx.x;
    if (i >= n) return;
    u64 d[8];
    merkle_leaf_xfe_body(limbs, limb_stride, i, stage + threadIdx.x, LEAF_THREADS, d, midstates);
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

constexpr u64 QUAD_PARENTS_MAX = 8192;   // levels with at most this many parents use one quad per hash
constexpr u64 QUAD_LEAVES_MAX = 8192;

// build all inner nodes above a leaf level of npo2 = 2^depth slots of which n_leaves hold digests
int merkle_inner_launch(u64* d_nodes, u32 depth, u64 n_leaves, hipStream_t stream, u64* root_out = nullptr, u64 seq = 0) {
    if (depth == 0) return BFS_OK;  // single leaf: root = leaf digest (merkle.py:43 nodes[1])
    u64 present = n_leaves;
    for (u32 lvl = depth; lvl-- > 0;) {
        const u64 count = 1ull << lvl;
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
