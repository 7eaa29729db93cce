// blake2b.hpp -- BLAKE2b-512 (RFC 7693), unkeyed, no salt/personalisation: what hashlib.blake2b(data).digest()
// computes in the reference (/root/reference/code/merkle.py:31,39  salted_merkle.py:34,43  fri.py:79).
// The compression function is host/device code: device kernels keep the 16 message words and the 16-word
// working state in VGPRs (all indices are compile-time constants after unrolling); the host uses the same
// code for Fri.sample_indices.
#pragma once
#include <stddef.h>
#include <string.h>

#include "gl.hpp"

namespace bfs {

#if defined(__HIP_DEVICE_COMPILE__)
#define BFS_B2_CONST __constant__
#else
#define BFS_B2_CONST
#endif

constexpr u64 B2_IV0 = 0x6a09e667f3bcc908ULL, B2_IV1 = 0xbb67ae8584caa73bULL, B2_IV2 = 0x3c6ef372fe94f82bULL,
              B2_IV3 = 0xa54ff53a5f1d36f1ULL, B2_IV4 = 0x510e527fade682d1ULL, B2_IV5 = 0x9b05688c2b3e6c1fULL,
              B2_IV6 = 0x1f83d9abfb41bd6bULL, B2_IV7 = 0x5be0cd19137e2179ULL;
constexpr u64 B2_PARAM0 = 0x01010040ULL;  // digest_length 64, key_length 0, fanout 1, depth 1

// rotate right by a compile-time amount.  On gfx950 a 64-bit rotation is two v_alignbit_b32 (a funnel shift of the
// two halves); the generic shift/or form compiles to three or four instructions, and the rotations are ~1/3 of G.
template <int R>
BFS_HD u64 rotr64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    if constexpr (R == 32) {
        return ((u64)lo << 32) | hi;
    } else if constexpr (R < 32) {
        return ((u64)__builtin_amdgcn_alignbit(lo, hi, R) << 32) | __builtin_amdgcn_alignbit(hi, lo, R);
    } else {
        return ((u64)__builtin_amdgcn_alignbit(hi, lo, R - 32) << 32) | __builtin_amdgcn_alignbit(lo, hi, R - 32);
    }
#else
    return (x >> R) | (x << (64 - R));
#endif
}

#define BFS_B2_G(a, b, c, d, x, y)          \
    do {                                    \
        a = a + b + (x);                    \
        d = rotr64<32>(d ^ a);              \
        c = c + d;                          \
        b = rotr64<24>(b ^ c);              \
        a = a + b + (y);                    \
        d = rotr64<16>(d ^ a);              \
        c = c + d;                          \
        b = rotr64<63>(b ^ c);              \
    } while (0)

#define BFS_B2_ROUND(s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
    do {                                                                                  \
        BFS_B2_G(v0, v4, v8, v12, m[s0], m[s1]);                                          \
        BFS_B2_G(v1, v5, v9, v13, m[s2], m[s3]);                                          \
        BFS_B2_G(v2, v6, v10, v14, m[s4], m[s5]);                                         \
        BFS_B2_G(v3, v7, v11, v15, m[s6], m[s7]);                                         \
        BFS_B2_G(v0, v5, v10, v15, m[s8], m[s9]);                                         \
        BFS_B2_G(v1, v6, v11, v12, m[s10], m[s11]);                                       \
        BFS_B2_G(v2, v7, v8, v13, m[s12], m[s13]);                                        \
        BFS_B2_G(v3, v4, v9, v14, m[s14], m[s15]);                                        \
    } while (0)

// h <- F(h, m, t, last);  t = number of message bytes absorbed so far including this block (< 2^64)
BFS_HD void blake2b_compress(u64 h[8], const u64 m[16], u64 t, bool last) {
    u64 v0 = h[0], v1 = h[1], v2 = h[2], v3 = h[3], v4 = h[4], v5 = h[5], v6 = h[6], v7 = h[7];
    u64 v8 = B2_IV0, v9 = B2_IV1, v10 = B2_IV2, v11 = B2_IV3;
    u64 v12 = B2_IV4 ^ t, v13 = B2_IV5, v14 = last ? ~B2_IV6 : B2_IV6, v15 = B2_IV7;
    BFS_B2_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    BFS_B2_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3);
    BFS_B2_ROUND(11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4);
    BFS_B2_ROUND(7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8);
    BFS_B2_ROUND(9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13);
    BFS_B2_ROUND(2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9);
    BFS_B2_ROUND(12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11);
    BFS_B2_ROUND(13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10);
    BFS_B2_ROUND(6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5);
    BFS_B2_ROUND(10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0);
    BFS_B2_ROUND(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    BFS_B2_ROUND(14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3);
    h[0] ^= v0 ^ v8;  h[1] ^= v1 ^ v9;  h[2] ^= v2 ^ v10; h[3] ^= v3 ^ v11;
    h[4] ^= v4 ^ v12; h[5] ^= v5 ^ v13; h[6] ^= v6 ^ v14; h[7] ^= v7 ^ v15;
}

BFS_HD void blake2b_init(u64 h[8]) {
    h[0] = B2_IV0 ^ B2_PARAM0; h[1] = B2_IV1; h[2] = B2_IV2; h[3] = B2_IV3;
    h[4] = B2_IV4; h[5] = B2_IV5; h[6] = B2_IV6; h[7] = B2_IV7;
}

// host convenience: one-shot digest of a byte string (little-endian host)
inline void blake2b_host(const void* data, size_t len, unsigned char out[64]) {
    u64 h[8];
    blake2b_init(h);
    const unsigned char* p = (const unsigned char*)data;
    size_t off = 0;
    u64 m[16];
    while (len - off > 128) {
        memcpy(m, p + off, 128);
        off += 128;
        blake2b_compress(h, m, off, false);
    }
    memset(m, 0, sizeof m);
    if (len - off) memcpy(m, p + off, len - off);
    blake2b_compress(h, m, len, true);
    memcpy(out, h, 64);
}

}  // namespace bfs
