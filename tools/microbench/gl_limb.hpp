// gl_limb.hpp -- the radix-16 butterfly block of the NTT in a carry-free limb form (round 6).
//
// What it replaces: dif<16> of ntt_core.hpp -- the four butterfly levels of /root/reference/code/ntt.py:18-23 on 16 values whose
// in-block twiddles are powers of two -- followed by the twiddle product of every output.
//
// Why.  In the two-word form a modular sum is 4-6 carry instructions and a shift twiddle 7-11, every one of them of the 4-cycle
// class (v_add_co / v_addc_co / v_mad_u64_u32: profiles/r05/valu_opcode_rates.txt); plain 32-bit VOP2 adds and shifts, which produce
// and consume no carry, retire at 2 cycles per wave64.  Here an element is four signed 32-bit limbs of weight 2^24,
//     x = l0 + l1 2^24 + l2 2^48 + l3 2^72   (mod p),            2^96 = -1 (mod p)  because  p | 2^96 + 1,
// so that
//   * a sum / difference is four v_add_u32 / v_sub_u32 and nothing else -- no carry, no reduction: limbs grow by one bit per level,
//     24 bits in, 28 bits after the block's four levels;
//   * a twiddle 2^(24 k) is a rotation of the limbs with a sign change at the wrap (2^96 = -1): register renaming plus the choice
//     between a - b and b - a, i.e. FREE.  Of the shift twiddles of a radix-16 block (2^(12 j) after level one, 2^(24 j) after level
//     two, 2^(48 j) after level three) only the four with odd j at level one need work: a 12-bit move inside the limbs, three
//     instructions per limb (shift12);
//   * the twiddle product after the block takes the limbs as they are:  x w = sum_i l_i (w 2^(24 i) mod p), eight v_mad_u64_u32 that
//     accumulate in their addend -- against four for a two-word operand -- and then ONE fold (gl_fold_word, four instructions)
//     instead of the 128-bit reduction (eleven): fourteen instructions against fifteen, canonical result.  The limbs are made
//     non-negative for the unsigned multiply-add by an offset O_i = 2^29 + d_i added at the block's last level with
//     sum_i O_i 2^(24 i) = k p = 0 (mod p)  (limb_offset below): one more v_add_u32 per limb.
//   The price is a table of FOUR words-pairs per twiddle (w 2^(24 i) mod p, i = 0..3) and the split of a loaded value into limbs
//   (four instructions).
// Per element of a block and its product: 4 + 16 + 4 + 3 two-cycle-class instructions + 14 four-cycle-class, against 28.5 + 15
// four-cycle-class ones in the two-word form.
#pragma once
#include "gl.hpp"

namespace bfs {

typedef int32_t i32;

struct L4 {
    u32 l[4];      // two's complement; value = sum (i32)l[i] 2^(24 i) mod p
};

// offsets that make every limb of a block output non-negative without changing the value mod p:
// T = 2^29 (1 + 2^24 + 2^48 + 2^72), k = round(T / p), k p = T + delta, delta = d0 + d1 2^24 + d2 2^48 in balanced digits
struct LimbOffsets { u32 o[4]; };
constexpr LimbOffsets limb_offsets() {
    const u128 T = ((u128)1 << 29) + ((u128)1 << 53) + ((u128)1 << 77) + ((u128)1 << 101);
    const u128 k = (T + GL_P / 2) / GL_P;
    const u128 kp = k * GL_P;
    // delta = kp - T, |delta| <= p/2 < 2^63
    const bool neg = kp < T;
    const u64 mag = (u64)(neg ? T - kp : kp - T);
    // balanced base-2^24 digits of the signed delta
    long long d = neg ? -(long long)mag : (long long)mag;
    long long dig[3] = {0, 0, 0};
    for (int i = 0; i < 2; ++i) {
        long long r = d % (1ll << 24);
        if (r >= (1ll << 23)) r -= (1ll << 24);
        if (r < -(1ll << 23)) r += (1ll << 24);
        dig[i] = r;
        d = (d - r) / (1ll << 24);
    }
    dig[2] = d;
    LimbOffsets o{};
    o.o[0] = (u32)((1ll << 29) + dig[0]);
    o.o[1] = (u32)((1ll << 29) + dig[1]);
    o.o[2] = (u32)((1ll << 29) + dig[2]);
    o.o[3] = (u32)(1ll << 29);
    return o;
}
constexpr LimbOffsets LIMB_OFF = limb_offsets();
constexpr bool limb_offsets_ok() {
    u128 v = 0;
    for (int i = 3; i >= 0; --i) v = (v << 24) + LIMB_OFF.o[i];
    if (v % GL_P != 0) return false;
    for (int i = 0; i < 4; ++i)
        if (LIMB_OFF.o[i] < (1u << 29) - (1u << 24) || LIMB_OFF.o[i] > (1u << 29) + (1u << 24)) return false;
    return true;
}
static_assert(limb_offsets_ok(), "limb offsets must sum to a multiple of p with every digit next to 2^29");

// a canonical (or any 64-bit) value as limbs 24 + 24 + 16 + 0 bits
BFS_HD L4 limb_split(u64 x) {
    L4 r;
    r.l[0] = (u32)x & 0xFFFFFFu;
    r.l[1] = (u32)(x >> 24) & 0xFFFFFFu;
    r.l[2] = (u32)(x >> 48);
    r.l[3] = 0;
    return r;
}

BFS_HD L4 limb_add(const L4& a, const L4& b) {
    L4 r;
    BFS_UNROLL
    for (int i = 0; i < 4; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}

// (a - b) 2^(24 K): limb i of the result is limb i - K of the difference, negated when it wrapped past 2^96 = -1
template <int K>
BFS_HD L4 limb_sub_rot(const L4& a, const L4& b) {
    L4 r;
    BFS_UNROLL
    for (int i = 0; i < 4; ++i) {
        const int s = (i - K) & 3;
        const bool neg = i < K;
        r.l[i] = neg ? b.l[s] - a.l[s] : a.l[s] - b.l[s];
    }
    return r;
}

// x 2^12: l_i = h_i 2^12 + m_i (h_i = l_i >> 12 arithmetic, 0 <= m_i < 2^12)  ->  new l_i = m_i 2^12 + h_(i-1), h_(-1) = -h_3.
// Also renormalises: |new l_i| < 2^24 + 2^(bits - 12).
BFS_HD L4 limb_shift12(const L4& a) {
    u32 h[4], m[4];
    BFS_UNROLL
    for (int i = 0; i < 4; ++i) {
        h[i] = (u32)((i32)a.l[i] >> 12);
        m[i] = (a.l[i] & 0xFFFu) << 12;
    }
    L4 r;
    r.l[0] = m[0] - h[3];
    r.l[1] = m[1] + h[0];
    r.l[2] = m[2] + h[1];
    r.l[3] = m[3] + h[2];
    return r;
}

// one level of the radix-Q network on limbs; twiddle of index I is 2^(12 I (16 / Q)): a rotation by K = that / 24 limbs and, for an
// odd multiple of 12, the 12-bit move.  LAST (Q == 2): the offsets go in, every limb of both outputs is in (0, 2^30).
template <int Q, int I>
BFS_HD void limb_dif_level(L4* x) {
    if constexpr (I < Q / 2) {
        const L4 a = x[I], b = x[I + Q / 2];
        constexpr int BITS = (192 / Q) * I;              // multiple of 12, < 96
        if constexpr (Q == 2) {
            L4 t;
            BFS_UNROLL
            for (int i = 0; i < 4; ++i) t.l[i] = a.l[i] + LIMB_OFF.o[i];
            BFS_UNROLL
            for (int i = 0; i < 4; ++i) {
                x[I].l[i] = t.l[i] + b.l[i];
                x[I + 1].l[i] = t.l[i] - b.l[i];
            }
        } else {
            x[I] = limb_add(a, b);
            const L4 d = limb_sub_rot<BITS / 24>(a, b);
            if constexpr ((BITS % 24) != 0) x[I + Q / 2] = limb_shift12(d);
            else x[I + Q / 2] = d;
        }
        limb_dif_level<Q, I + 1>(x);
    }
}

// the same network as dif<Q> (ntt_core.hpp): result for output index k in x[bitrev(k)]; outputs carry the offsets (limbs in (0, 2^30))
template <int Q>
BFS_HD void limb_dif(L4* x) {
    if constexpr (Q >= 2) {
        limb_dif_level<Q, 0>(x);
        limb_dif<Q / 2>(x);
        limb_dif<Q / 2>(x + Q / 2);
    }
}

// the host emulation (tests/emu, -DBFS_CHECK_CANONICAL) counts a block output whose limbs left [0, 2^30) with the non-canonical operands
#if defined(BFS_CHECK_CANONICAL) && !defined(__HIP_DEVICE_COMPILE__)
#define BFS_LIMB_RANGE(x) do { for (int i_ = 0; i_ < 4; ++i_) if ((x).l[i_] >= (1u << 30)) ++gl_canonical_violations(); } while (0)
#else
#define BFS_LIMB_RANGE(x) ((void)0)
#endif

// A twiddle as the product wants it: (w 2^(24 i) mod p) for i = 0..3, low and high words.  32 bytes; tables of them are built on the host.
struct LimbTw {
    u32 lo[4], hi[4];
};
inline LimbTw limb_tw_of(u64 w) {
    LimbTw t;
    u64 v = w;
    for (int i = 0; i < 4; ++i) {
        t.lo[i] = (u32)v;
        t.hi[i] = (u32)(v >> 32);
        v = gl_mul(v, 1ull << 24);
    }
    return t;
}

// x w for a block output x (limbs in [0, 2^30)): canonical.  accL = sum l_i lo_i and accH = sum l_i hi_i stay below 2^64 (four
// terms below 2^62 each); x w = accL + 2^32 accH = (top : s : accL_lo) with s = accL_hi + accH_lo, top = accH_hi + carry < 2^32.
BFS_HD u64 limb_mul(const L4& x, const LimbTw& t) {
    BFS_LIMB_RANGE(x);
    u64 aL = (u64)x.l[0] * t.lo[0];
    u64 aH = (u64)x.l[0] * t.hi[0];
    BFS_UNROLL
    for (int i = 1; i < 4; ++i) {
        aL += (u64)x.l[i] * t.lo[i];
        aH += (u64)x.l[i] * t.hi[i];
    }
    const u64 s = (aL >> 32) + (u64)(u32)aH;
    const u32 top = (u32)(aH >> 32) + (u32)(s >> 32);
    return gl_reduce96(top, (s << 32) | (u32)aL);
}

// a block output back to a canonical value without a twiddle: the product with w = 1, whose table is (1, 2^24, 2^48, 2^40 - 2^8)
BFS_HD u64 limb_join(const L4& x) {
    BFS_LIMB_RANGE(x);
    // accL = l0 + l1 2^24 + l3 (2^32 - 2^8) ; accH = l2 2^16 + l3 (2^8 - 1) + carry...: written as the general product with constant words
    u64 aL = (u64)x.l[0] + ((u64)x.l[1] << 24) + (u64)x.l[3] * 0xFFFFFF00ull;
    u64 aH = ((u64)x.l[2] << 16) + (u64)x.l[3] * 0xFFull;
    const u64 s = (aL >> 32) + (u64)(u32)aH;
    const u32 top = (u32)(aH >> 32) + (u32)(s >> 32);
    return gl_reduce96(top, (s << 32) | (u32)aL);
}

}  // namespace bfs
