// merkle_core.hpp -- per-thread bodies of the Merkle kernels (host/device so that the host emulation test can
// run them).  Tree layout is the reference's: nodes[] has 2*npo2 entries of 64 bytes, leaf i at npo2 + i,
// parent k = H(nodes[2k] || nodes[2k+1]), root at index 1   (/root/reference/code/merkle.py:26-44).
#pragma once
#include "blake2b.hpp"
#include "leaf_encode.hpp"

namespace bfs {

// BLAKE2b of a message staged as word-major 64-bit words (word w at base[w*stride]); `total` bytes
BFS_HD void blake2b_staged(const u64* base, u32 stride, u32 total, u64 h[8]) {
    blake2b_init(h);
    const u32 nwords = (total + 7) / 8;
    const u32 nblk = total ? (total + 127) / 128 : 1;
    for (u32 b = 0; b < nblk; ++b) {
        u64 m[16];
        BFS_UNROLL
        for (int j = 0; j < 16; ++j) {
            u32 w = b * 16 + (u32)j;
            m[j] = w < nwords ? base[(size_t)w * stride] : 0;
        }
        const bool last = (b + 1 == nblk);
        blake2b_compress(h, m, last ? (u64)total : (u64)(b + 1) * 128, last);
    }
}

// leaf i of an extension-field codeword stored limb-major; digest -> nodes[(npo2 + i)]
BFS_HD void merkle_leaf_xfe_body(const u64* limbs, u64 limb_stride, u64 i, u64* stage, u32 stride, u64* digest_out) {
    LeafWriter w;
    w.init(stage, stride);
    u32 total = encode_xfe_leaf(w, limbs[i], limbs[limb_stride + i], limbs[2 * limb_stride + i]);
    u64 h[8];
    blake2b_staged(stage, stride, total, h);
    BFS_UNROLL
    for (int j = 0; j < 8; ++j) digest_out[j] = h[j];
}

BFS_HD void merkle_leaf_bfe_body(const u64* values, u64 i, u64* stage, u32 stride, u64* digest_out) {
    LeafWriter w;
    w.init(stage, stride);
    u32 total = encode_bfe_leaf(w, values[i]);
    u64 h[8];
    blake2b_staged(stage, stride, total, h);
    BFS_UNROLL
    for (int j = 0; j < 8; ++j) digest_out[j] = h[j];
}

// parent of two children given as 8-word digests; `present` = how many of the two child slots hold a digest
// (0, 1 or 2).  An absent leaf slot is the reference's 32 zero bytes (merkle.py:26), so the preimage is
// 64*present + 32*(2-present) bytes long.  Present children are always a prefix.
BFS_HD void merkle_parent_body(const u64* left, const u64* right, int present, u64* out) {
    u64 m[16];
    BFS_UNROLL
    for (int j = 0; j < 8; ++j) {
        m[j] = present >= 1 ? left[j] : 0;
        m[8 + j] = present >= 2 ? right[j] : 0;
    }
    u64 h[8];
    blake2b_init(h);
    blake2b_compress(h, m, (u64)(64 * present + 32 * (2 - present)), true);
    BFS_UNROLL
    for (int j = 0; j < 8; ++j) out[j] = h[j];
}

}  // namespace bfs
