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

// ---- leaf midstates: BLAKE2b state after the constant first block, per (class, body length) ----
// layout: [0, 512*8): class k = 1 indexed by body length; [512*8, 1024*8): class k >= 2; [1024*8, 1025*8): digest of the zero element
constexpr int LEAF_MS_LEN = 512;
constexpr int LEAF_MS_WORDS = (2 * LEAF_MS_LEN + 1) * 8;

inline void leaf_midstates(u64* table /* LEAF_MS_WORDS */) {
    for (int cls = 0; cls < 2; ++cls)
        for (int body = 0; body < LEAF_MS_LEN; ++body) {
            unsigned char block[128];
            const u64 hdr = 0x80ull | (0x04ull << 8) | (0x95ull << 16) | ((u64)body << 24);
            memcpy(block, &hdr, 8);
            memset(block + 8, 0, 3);
            memcpy(block + 11, tpl::XFE_PRE_A, tpl::XFE_PRE_A_LEN);
            if (cls == 0) memcpy(block + 124, tpl::XFE_PRE_B, 4);                        // k = 1: no MARK
            else { block[124] = 0x28; memcpy(block + 125, tpl::XFE_PRE_B, 3); }           // k >= 2
            u64 m[16], h[8];
            memcpy(m, block, 128);
            blake2b_init(h);
            blake2b_compress(h, m, 128, false);
            memcpy(table + ((size_t)cls * LEAF_MS_LEN + body) * 8, h, 64);
        }
    unsigned char d[64];
    blake2b_host(tpl::XFE_K0, tpl::XFE_K0_LEN, d);
    memcpy(table + (size_t)2 * LEAF_MS_LEN * 8, d, 64);
}

// BLAKE2b of a message whose first 128 bytes are already absorbed into h; tail words at base[w*stride], `total` = full length
BFS_HD void blake2b_staged_tail(const u64* base, u32 stride, u32 total, u64 h[8]) {
    const u32 nwords = (total - 128 + 7) / 8;
    const u32 nblk = (total + 127) / 128;          // >= 2 for every non-zero leaf
    for (u32 b = 1; b < nblk; ++b) {
        u64 m[16];
        BFS_UNROLL
        for (int j = 0; j < 16; ++j) {
            u32 w = (b - 1) * 16 + (u32)j;
            m[j] = w < nwords ? base[(size_t)w * stride] : 0;
        }
        const bool last = (b + 1 == nblk);
        blake2b_compress(h, m, last ? (u64)total : (u64)(b + 1) * 128, last);
    }
}

// leaf i of an extension-field codeword stored limb-major; digest -> nodes[(npo2 + i)]
BFS_HD void merkle_leaf_xfe_body(const u64* limbs, u64 limb_stride, u64 i, u64* stage, u32 stride, u64* digest_out, const u64* midstates) {
    const u64 c0 = limbs[i], c1 = limbs[limb_stride + i], c2 = limbs[2 * limb_stride + i];
    const u32 k = xfe_leaf_k(c0, c1, c2);
    u64 h[8];
    if (k == 0) {
        BFS_UNROLL
        for (int j = 0; j < 8; ++j) h[j] = midstates[(size_t)2 * LEAF_MS_LEN * 8 + j];
    } else {
        const u32 body = xfe_leaf_body_len(k, c0, c1, c2);
        const u64* ms = midstates + ((size_t)(k == 1 ? 0 : 1) * LEAF_MS_LEN + body) * 8;
        BFS_UNROLL
        for (int j = 0; j < 8; ++j) h[j] = ms[j];
        LeafWriter w;
        w.init(stage, stride);
        encode_xfe_leaf_tail(w, k, c0, c1, c2);
        blake2b_staged_tail(stage, stride, body + 11, h);
    }
    BFS_UNROLL
    for (int j = 0; j < 8; ++j) digest_out[j] = h[j];
}

// ---- the same leaf, streamed: the preimage is hashed block by block while it is being encoded ------------------------------------
// merkle_leaf_xfe_body stages the whole tail (36 words per lane: 18 KiB per wave, two waves per SIMD).  Here a lane owns
// XFE_STREAM_WORDS = 20 words: the first tail block is complete once a statically known prefix of the template has been written (the
// three integers move a lane's position by at most 27 bytes), so ALL lanes compress it at the same place in the code, shift their <= 4
// left-over words to the front and finish the encoding; the remaining one or two blocks follow as in the staged form.  The split
// point depends on the class K (how many coefficients the element stores), so the wave-level caller (merkle.hip: xfe_leaves_wave)
// runs the classes present in a wave one after the other -- exactly one for every codeword of random extension elements and for every
// lifted base-field codeword.
constexpr int XFE_STREAM_WORDS = 20;         // 10 KiB per wave: sixteen waves per CU
template <int K> struct XfeStreamSplit;      // bytes of the class' last constant segment written BEFORE the first compression
template <> struct XfeStreamSplit<1> { static constexpr int BYTES = 96; };   // 41 + int + 96 = 139..148 >= 128
template <> struct XfeStreamSplit<2> { static constexpr int BYTES = 24; };   // 42 + int + 58 + int + 24 = 128..146
template <> struct XfeStreamSplit<3> { static constexpr int BYTES = 8; };    // 42 + int + 58 + int + 16 + int + 8 = 130..157 (20 words); the rest of the tail is <= 20 words as well

template <int K>
BFS_HD void merkle_leaf_xfe_stream(u64 c0, u64 c1, u64 c2, u64* stage, u32 stride, u64 h[8], const u64* midstates) {
    static_assert(K >= 1 && K <= 3, "class");
    constexpr int SPLIT = XfeStreamSplit<K>::BYTES;
    static_assert(SPLIT % 8 == 0, "the constant segment is split between two words");
    const u32 body = xfe_leaf_body_len((u32)K, c0, c1, c2);
    const u32 total = body + 11;
    const u64* ms = midstates + ((size_t)(K == 1 ? 0 : 1) * LEAF_MS_LEN + body) * 8;
    BFS_UNROLL
    for (int j = 0; j < 8; ++j) h[j] = ms[j];
    LeafWriter w;
    w.init(stage, stride);
    const u64* post;
    if constexpr (K == 1) {
        w.put_const<tpl::XFE_PRE_B4_LEN>(tpl::XFE_PRE_B4);
        w.put_int(c0);
        post = tpl::XFE_POST1;
    } else {
        w.put_const<tpl::XFE_PRE_B3_LEN>(tpl::XFE_PRE_B3);
        w.put_int(c0);
        w.put_const<tpl::XFE_MID_A_LEN>(tpl::XFE_MID_A);
        w.put_int(c1);
        if constexpr (K == 2) {
            post = tpl::XFE_POST2;
        } else {
            w.put_const<tpl::XFE_MID_B_LEN>(tpl::XFE_MID_B);
            w.put_int(c2);
            post = tpl::XFE_POST3;
        }
    }
    constexpr int POST_LEN = K == 1 ? tpl::XFE_POST1_LEN : (K == 2 ? tpl::XFE_POST2_LEN : tpl::XFE_POST3_LEN);
    w.put_const<SPLIT>(post);
    // every lane holds at least 16 complete words now: bytes [128, 256) of the pickle, never the last block (totals are >= 345).
    // ONE compression site for all blocks (a second inlined copy of the round function cost 76 VGPRs): the first turn of the loop
    // takes those 16 words, shifts the <= 5 left-over words to the front and finishes the encoding; the later turns are the staged form.
    const u32 nblk = (total + 127) / 128;           // 3 or 4
    u32 first = 0, avail = 16;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
    for (u32 b = 1; b < nblk; ++b) {
        u64 m[16];
        BFS_UNROLL
        for (int j = 0; j < 16; ++j) {
            const u32 wi = first + (u32)j;
            m[j] = wi < avail ? stage[(size_t)wi * stride] : 0;
        }
        const bool last = (b + 1 == nblk);
        blake2b_compress(h, m, last ? (u64)total : (u64)(b + 1) * 128, last);
        if (b == 1) {
            BFS_UNROLL
            for (int j = 0; j < XFE_STREAM_WORDS - 16; ++j) stage[(size_t)j * stride] = stage[(size_t)(16 + j) * stride];
            w.wpos -= 16;
            w.put_const<POST_LEN - SPLIT>(post + SPLIT / 8);
            w.finish();
            avail = (total - 256 + 7) / 8;          // <= 20
        } else {
            first += 16;
        }
    }
}

// the zero element: a table lookup
BFS_HD void merkle_leaf_xfe_zero(u64 h[8], const u64* midstates) {
    BFS_UNROLL
    for (int j = 0; j < 8; ++j) h[j] = midstates[(size_t)2 * LEAF_MS_LEN * 8 + j];
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
// eight digest words; on the device as four 16-byte accesses (a thread-per-hash kernel reads at a 128-byte lane stride,
// and halving the number of load instructions took the inner-level kernel from 0.151 to 0.128 ms per 2^21 parents,
// tools/microbench/merkle_mb.hip).  Node arrays are 16-byte aligned (checked at the C ABI).
BFS_HD void load_digest(const u64* p, u64* m) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
    const u64x2* q = (const u64x2*)p;
    BFS_UNROLL
    for (int j = 0; j < 4; ++j) {
        const u64x2 v = q[j];
        m[2 * j] = v.x;
        m[2 * j + 1] = v.y;
    }
#else
    for (int j = 0; j < 8; ++j) m[j] = p[j];
#endif
}

BFS_HD void merkle_parent_body(const u64* left, const u64* right, int present, u64* out) {
    u64 m[16];
    BFS_UNROLL
    for (int j = 0; j < 16; ++j) m[j] = 0;
    if (present >= 1) load_digest(left, m);
    if (present >= 2) load_digest(right, m + 8);
    u64 h[8];
    blake2b_init(h);
    blake2b_compress(h, m, (u64)(64 * present + 32 * (2 - present)), true);
    BFS_UNROLL
    for (int j = 0; j < 8; ++j) out[j] = h[j];
}

}  // namespace bfs
