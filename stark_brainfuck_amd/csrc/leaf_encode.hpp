// leaf_encode.hpp -- byte-exact re-creation of pickle.dumps(leaf) for the reference's element classes, written
// as host/device code.  Replaces the `pickle.dumps(self.leafs[i])` of /root/reference/code/merkle.py:30 (and
// salted_merkle.py:32) for leaves that are ExtensionFieldElement / BaseFieldElement objects: the GPU leaf kernel
// synthesises the preimage from the limbs and feeds it to BLAKE2b without the bytes ever touching HBM.
//
// The byte stream is a fixed template (pickle_templates.hpp) with variable-length integer opcodes spliced in,
// so every thread appends the same sequence of segments at thread-specific byte offsets.  LeafWriter is a
// funnel: it keeps < 8 pending bytes in a register and emits whole 64-bit words to a word-major staging area
// (LDS on the GPU: word w of lane l lives at base[w * stride], conflict-free for any per-lane w).
#pragma once
#include "gl.hpp"
#include "pickle_templates.hpp"

namespace bfs {

constexpr int XFE_LEAF_MAX_BYTES = 11 + tpl::XFE_PRE_A_LEN + 1 + tpl::XFE_PRE_B_LEN + 11 + tpl::XFE_MID_A_LEN + 11 + tpl::XFE_MID_B_LEN + 11 + tpl::XFE_POST3_LEN;
constexpr int XFE_LEAF_MAX_WORDS = (XFE_LEAF_MAX_BYTES + 7) / 8;  // 52
constexpr int BFE_LEAF_MAX_BYTES = 11 + tpl::BFE_PRE_LEN + 11 + tpl::BFE_POST_LEN;
constexpr int BFE_LEAF_MAX_WORDS = (BFE_LEAF_MAX_BYTES + 7) / 8;  // 15

struct LeafWriter {
    u64* base;
    u32 stride;
    u64 acc;
    u32 fill;  // pending bytes in acc (0..7)
    u32 wpos;  // words emitted

    BFS_HD void init(u64* b, u32 s) { base = b; stride = s; acc = 0; fill = 0; wpos = 0; }

    // append the low `nbytes` (0..8) bytes of data; bytes of `data` above nbytes must be zero
    BFS_HD void put(u64 data, u32 nbytes) {
        acc |= data << (8 * fill);
        u32 nf = fill + nbytes;
        if (nf >= 8) {
            base[(size_t)wpos * stride] = acc;
            ++wpos;
            acc = fill ? (data >> (8 * (8 - fill))) : 0;
            nf -= 8;
        }
        fill = nf;
    }

    template <int LEN>
    BFS_HD void put_const(const u64* words) {
        BFS_UNROLL
        for (int i = 0; i < LEN / 8; ++i) put(words[i], 8);
        if constexpr (LEN % 8 != 0) put(words[LEN / 8], LEN % 8);
    }

    // pickle's integer opcodes (save_long): BININT1 / BININT2 / BININT / LONG1
    BFS_HD void put_int(u64 v) {
        u64 lo, hi = 0;
        u32 len;
        if (v < (1ull << 8)) { lo = 0x4b | (v << 8); len = 2; }
        else if (v < (1ull << 16)) { lo = 0x4d | (v << 8); len = 3; }
        else if (v < (1ull << 31)) { lo = 0x4a | (v << 8); len = 5; }
        else {
            u32 bits = 64 - (u32)__builtin_clzll(v);
            u32 nn = bits / 8 + 1;
            lo = 0x8a | ((u64)nn << 8) | (v << 16);
            hi = v >> 48;
            len = 2 + nn;
        }
        put(lo, len < 8 ? len : 8);
        put(hi, len > 8 ? len - 8 : 0);
    }

    BFS_HD u32 finish() {
        u32 total = wpos * 8 + fill;
        if (fill) { base[(size_t)wpos * stride] = acc; ++wpos; acc = 0; fill = 0; }
        return total;
    }
};

BFS_HD u32 pickle_int_len(u64 v) {
    if (v < (1ull << 8)) return 2;
    if (v < (1ull << 16)) return 3;
    if (v < (1ull << 31)) return 5;
    return 2 + (64 - (u32)__builtin_clzll(v)) / 8 + 1;
}

BFS_HD void put_frame_header(LeafWriter& w, u32 body_len) {
    w.put(0x80ull | (0x04ull << 8) | (0x95ull << 16) | ((u64)body_len << 24), 8);  // 80 04 95 + low 5 bytes of the length
    w.put(0, 3);                                                                   // upper 3 bytes of the u64 length
}

// pickle.dumps(ExtensionFieldElement) for limbs (c0, c1, c2); the stored polynomial drops trailing zero
// coefficients (extension_field.py:6-9).  Returns the number of bytes written.
BFS_HD u32 encode_xfe_leaf(LeafWriter& w, u64 c0, u64 c1, u64 c2) {
    const u32 k = c2 ? 3 : (c1 ? 2 : (c0 ? 1 : 0));
    if (k == 0) {
        w.put_const<tpl::XFE_K0_LEN>(tpl::XFE_K0);
        return w.finish();
    }
    u32 body = tpl::XFE_PRE_A_LEN + tpl::XFE_PRE_B_LEN + pickle_int_len(c0);
    if (k == 1) body += tpl::XFE_POST1_LEN;
    else {
        body += 1 + tpl::XFE_MID_A_LEN + pickle_int_len(c1);
        body += (k == 2) ? tpl::XFE_POST2_LEN : tpl::XFE_MID_B_LEN + pickle_int_len(c2) + tpl::XFE_POST3_LEN;
    }
    put_frame_header(w, body);
    w.put_const<tpl::XFE_PRE_A_LEN>(tpl::XFE_PRE_A);
    w.put(k >= 2 ? 0x28 : 0, k >= 2 ? 1 : 0);  // MARK in front of a multi-item coefficient list
    w.put_const<tpl::XFE_PRE_B_LEN>(tpl::XFE_PRE_B);
    w.put_int(c0);
    if (k == 1) {
        w.put_const<tpl::XFE_POST1_LEN>(tpl::XFE_POST1);
    } else {
        w.put_const<tpl::XFE_MID_A_LEN>(tpl::XFE_MID_A);
        w.put_int(c1);
        if (k == 2) {
            w.put_const<tpl::XFE_POST2_LEN>(tpl::XFE_POST2);
        } else {
            w.put_const<tpl::XFE_MID_B_LEN>(tpl::XFE_MID_B);
            w.put_int(c2);
            w.put_const<tpl::XFE_POST3_LEN>(tpl::XFE_POST3);
        }
    }
    return w.finish();
}

// ---- midstate form ---------------------------------------------------------------------------------------------
// The first 128 bytes of a non-zero leaf are constant except for the length field, so BLAKE2b's state after the first
// block is tabulated per (class, body length) on the host (merkle_core.hpp: leaf_midstates).  The kernels then only
// assemble and hash the bytes from offset 128 on: 3 compressions instead of 4, 36 staging words instead of 52.
constexpr int XFE_TAIL_MAX_WORDS = (XFE_LEAF_MAX_BYTES - 128 + 7) / 8;   // 36

BFS_HD u32 xfe_leaf_k(u64 c0, u64 c1, u64 c2) { return c2 ? 3u : (c1 ? 2u : (c0 ? 1u : 0u)); }

// body length (total - 11) of the pickle of a non-zero element
BFS_HD u32 xfe_leaf_body_len(u32 k, u64 c0, u64 c1, u64 c2) {
    u32 body = tpl::XFE_PRE_A_LEN + tpl::XFE_PRE_B_LEN + pickle_int_len(c0);
    if (k == 1) return body + tpl::XFE_POST1_LEN;
    body += 1 + tpl::XFE_MID_A_LEN + pickle_int_len(c1);
    return body + ((k == 2) ? tpl::XFE_POST2_LEN : tpl::XFE_MID_B_LEN + pickle_int_len(c2) + tpl::XFE_POST3_LEN);
}

// bytes [128, total) of the leaf pickle for k >= 1; returns how many were written
BFS_HD u32 encode_xfe_leaf_tail(LeafWriter& w, u32 k, u64 c0, u64 c1, u64 c2) {
    if (k == 1) {
        w.put_const<tpl::XFE_PRE_B4_LEN>(tpl::XFE_PRE_B4);
        w.put_int(c0);
        w.put_const<tpl::XFE_POST1_LEN>(tpl::XFE_POST1);
    } else {
        w.put_const<tpl::XFE_PRE_B3_LEN>(tpl::XFE_PRE_B3);
        w.put_int(c0);
        w.put_const<tpl::XFE_MID_A_LEN>(tpl::XFE_MID_A);
        w.put_int(c1);
        if (k == 2) {
            w.put_const<tpl::XFE_POST2_LEN>(tpl::XFE_POST2);
        } else {
            w.put_const<tpl::XFE_MID_B_LEN>(tpl::XFE_MID_B);
            w.put_int(c2);
            w.put_const<tpl::XFE_POST3_LEN>(tpl::XFE_POST3);
        }
    }
    return w.finish();
}

// pickle.dumps(BaseFieldElement) with a stand-alone BaseField instance (algebra.py:110-115)
BFS_HD u32 encode_bfe_leaf(LeafWriter& w, u64 v) {
    put_frame_header(w, tpl::BFE_PRE_LEN + pickle_int_len(v) + tpl::BFE_POST_LEN);
    w.put_const<tpl::BFE_PRE_LEN>(tpl::BFE_PRE);
    w.put_int(v);
    w.put_const<tpl::BFE_POST_LEN>(tpl::BFE_POST);
    return w.finish();
}

}  // namespace bfs
