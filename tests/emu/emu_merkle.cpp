// Host emulation of the leaf encoder / BLAKE2b / Merkle bodies (TEST INFRASTRUCTURE, see emu_ntt.cpp).
#include <cstring>
#include <vector>

#include "../../stark_brainfuck_amd/csrc/keccak.hpp"
#include "../../stark_brainfuck_amd/csrc/merkle_core.hpp"

using namespace bfs;

extern "C" int emu_xfe_leaf_pickle(const u64 limbs[3], unsigned char* out /* >= 416 */) {
    u64 stage[XFE_LEAF_MAX_WORDS];
    LeafWriter w;
    w.init(stage, 1);
    u32 total = encode_xfe_leaf(w, limbs[0], limbs[1], limbs[2]);
    memcpy(out, stage, (total + 7) / 8 * 8);
    return (int)total;
}

extern "C" int emu_bfe_leaf_pickle(u64 v, unsigned char* out /* >= 128 */) {
    u64 stage[BFE_LEAF_MAX_WORDS];
    LeafWriter w;
    w.init(stage, 1);
    u32 total = encode_bfe_leaf(w, v);
    memcpy(out, stage, (total + 7) / 8 * 8);
    return (int)total;
}

extern "C" void emu_blake2b(const unsigned char* data, size_t len, unsigned char out[64]) { blake2b_host(data, len, out); }

extern "C" void emu_shake256(const unsigned char* data, size_t len, unsigned char* out, size_t outlen) { shake256(data, len, out, outlen); }

// full tree over an SoA extension codeword, emulating leaf kernel (64-lane word-major staging) + parent levels
extern "C" void emu_merkle_xfe(const u64* limbs, u64 stride, u64 n, u64* nodes /* 2*npo2*8 words, zeroed by caller */) {
    u32 depth = 0;
    while ((1ull << depth) < n) ++depth;
    const u64 npo2 = 1ull << depth;
    std::vector<u64> stage(XFE_TAIL_MAX_WORDS * 64), ms(LEAF_MS_WORDS);
    leaf_midstates(ms.data());
    for (u64 i = 0; i < n; ++i) merkle_leaf_xfe_body(limbs, stride, i, stage.data() + (i % 64), 64, nodes + (npo2 + i) * 8, ms.data());
    u64 present = n;
    for (u32 lvl = depth; lvl-- > 0;) {
        const u64 count = 1ull << lvl;
        for (u64 t = 0; t < count; ++t) {
            u64 c = 2 * t;
            int pr = c + 1 < present ? 2 : (c < present ? 1 : 0);
            merkle_parent_body(nodes + 2 * (count + t) * 8, nodes + (2 * (count + t) + 1) * 8, pr, nodes + (count + t) * 8);
        }
        present = 2 * count;
    }
}

// the streamed leaf (merkle_leaf_xfe_stream<K>) for every element of a codeword whose elements all share class K, and the staged form in
// the half-wave layout (stride 32) the kernel falls back to for mixed waves: 64-byte digests out, -1 on an element of another class
extern "C" int emu_xfe_leaf_stream(const u64* limbs, u64 stride, u64 n, int klass, u64* digests) {
    std::vector<u64> stage(XFE_STREAM_WORDS * 64, 0xA5A5A5A5A5A5A5A5ULL), ms(LEAF_MS_WORDS);
    leaf_midstates(ms.data());
    for (u64 i = 0; i < n; ++i) {
        const u64 c0 = limbs[i], c1 = limbs[stride + i], c2 = limbs[2 * stride + i];
        if ((int)xfe_leaf_k(c0, c1, c2) != klass) return -1;
        u64* st = stage.data() + (i % 64);
        u64* h = digests + 8 * i;
        if (klass == 1) merkle_leaf_xfe_stream<1>(c0, c1, c2, st, 64, h, ms.data());
        else if (klass == 2) merkle_leaf_xfe_stream<2>(c0, c1, c2, st, 64, h, ms.data());
        else merkle_leaf_xfe_stream<3>(c0, c1, c2, st, 64, h, ms.data());
    }
    return 0;
}
