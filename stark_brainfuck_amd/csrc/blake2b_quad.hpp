// blake2b_quad.hpp -- BLAKE2b compression spread over FOUR lanes (device only).
//
// Why: one thread per hash is right when there are >= 10^5 hashes in flight (leaf level of a big tree), but the upper
// levels of every Merkle tree and all late FRI rounds have few hashes, and then what matters is the LATENCY of one
// compression: ~2900 dependent-ish VALU instructions = 4.5-5 us for a lone wave (profiles/r01: 45 us for the top nine levels
// of every tree).  BLAKE2b's state is a 4x4 matrix whose four columns (then four diagonals) are mixed independently, so
// lane j of a quad keeps column j (a=v[j], b=v[4+j], c=v[8+j], d=v[12+j]), runs G on it, rotates rows with DPP quad_perm
// moves (no LDS) to form the diagonals, runs G again and rotates back.  ~1000 instructions per lane per compression.
// Message words are read from LDS with per-lane indices taken from the (compile-time) sigma schedule.
// Same function as blake2b_compress() in blake2b.hpp (RFC 7693); replaces hashlib.blake2b of merkle.py:31,39.
#pragma once
#include "blake2b.hpp"

namespace bfs {

#if defined(__HIP_DEVICE_COMPILE__)

template <int CTRL>
__device__ __forceinline__ u64 quad_perm64(u64 v) {
    int lo = (int)(u32)v, hi = (int)(u32)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, true);
    return ((u64)(u32)hi << 32) | (u32)lo;
}

constexpr int B2_SIGMA[12][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};

// the four lanes' message indices for (round R, slot Q: 0 column-x, 1 column-y, 2 diagonal-x, 3 diagonal-y), 4 bits each
template <int R, int Q>
constexpr u32 sigma_pack() {
    u32 p = 0;
    for (int j = 0; j < 4; ++j) {
        int idx = (Q < 2) ? B2_SIGMA[R][2 * j + Q] : B2_SIGMA[R][8 + 2 * j + (Q - 2)];
        p |= (u32)idx << (4 * j);
    }
    return p;
}

struct QuadLane {
    u32 j4;        // 4 * (lane & 3)
    u64 iv_c, iv_d;  // IV[j], IV[4 + j]
};

__device__ __forceinline__ QuadLane quad_lane(u32 lane) {
    QuadLane q;
    const u32 j = lane & 3;
    q.j4 = 4 * j;
    q.iv_c = j == 0 ? B2_IV0 : (j == 1 ? B2_IV1 : (j == 2 ? B2_IV2 : B2_IV3));
    q.iv_d = j == 0 ? B2_IV4 : (j == 1 ? B2_IV5 : (j == 2 ? B2_IV6 : B2_IV7));
    return q;
}

// initial chaining value of this lane: (h[j], h[4+j])
__device__ __forceinline__ void blake2b_init_quad(const QuadLane& q, u64& hl, u64& hh) {
    hl = q.j4 == 0 ? (q.iv_c ^ B2_PARAM0) : q.iv_c;
    hh = q.iv_d;
}

// the four message words lane j needs in round R: column step (x, y), diagonal step (x, y)
struct QuadWords { u64 cx, cy, dx, dy; };
template <int R>
__device__ __forceinline__ QuadWords quad_words(const u64* m, u32 j4) {
    QuadWords w;
    w.cx = m[__builtin_amdgcn_ubfe(sigma_pack<R, 0>(), j4, 4)];
    w.cy = m[__builtin_amdgcn_ubfe(sigma_pack<R, 1>(), j4, 4)];
    w.dx = m[__builtin_amdgcn_ubfe(sigma_pack<R, 2>(), j4, 4)];
    w.dy = m[__builtin_amdgcn_ubfe(sigma_pack<R, 3>(), j4, 4)];
    __builtin_amdgcn_sched_barrier(0);       // (left to itself the scheduler sinks the reads back to where the words are used)
    return w;
}

__device__ __forceinline__ void blake2b_round_quad(u64& a, u64& b, u64& c, u64& d, const QuadWords& w) {
    BFS_B2_G(a, b, c, d, w.cx, w.cy);
    b = quad_perm64<0x39>(b);  // lane j takes b of lane j+1
    c = quad_perm64<0x4E>(c);  // lane j takes c of lane j+2
    d = quad_perm64<0x93>(d);  // lane j takes d of lane j+3
    BFS_B2_G(a, b, c, d, w.dx, w.dy);
    b = quad_perm64<0x93>(b);
    c = quad_perm64<0x4E>(c);
    d = quad_perm64<0x39>(d);
}

// (hl, hh) <- F((hl, hh), m, t, last) for the hash owned by this quad; m: 16 message words (LDS).  The words of round R + 1 are
// requested before round R is computed: a quad kernel runs ONE wave per SIMD, so nothing else hides the LDS latency of a read that
// is issued where its value is needed (~40 reads per compression).
__device__ __forceinline__ void blake2b_compress_quad(const QuadLane& q, u64& hl, u64& hh, const u64* m, u64 t, bool last) {
    u64 a = hl, b = hh, c = q.iv_c, d = q.iv_d;
    if (q.j4 == 0) d ^= t;               // v12 ^= t0
    if (q.j4 == 8 && last) d = ~d;       // v14 = ~v14 on the final block
    QuadWords w0 = quad_words<0>(m, q.j4), w1;
#define BFS_B2_QUAD_PAIR(R0, R1, R2)                                  \
    w1 = quad_words<R1>(m, q.j4); blake2b_round_quad(a, b, c, d, w0); \
    w0 = quad_words<R2>(m, q.j4); blake2b_round_quad(a, b, c, d, w1);
    BFS_B2_QUAD_PAIR(0, 1, 2)
    BFS_B2_QUAD_PAIR(2, 3, 4)
    BFS_B2_QUAD_PAIR(4, 5, 6)
    BFS_B2_QUAD_PAIR(6, 7, 8)
    BFS_B2_QUAD_PAIR(8, 9, 10)
    w1 = quad_words<11>(m, q.j4); blake2b_round_quad(a, b, c, d, w0);
    blake2b_round_quad(a, b, c, d, w1);
#undef BFS_B2_QUAD_PAIR
    hl ^= a ^ c;
    hh ^= b ^ d;
}

#endif  // __HIP_DEVICE_COMPILE__

}  // namespace bfs
