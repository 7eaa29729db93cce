// keccak.hpp -- SHAKE256 (FIPS 202), host only.  The Fiat-Shamir challenge of the reference is
// shake_256(pickle.dumps(objects)).digest(32)  (/root/reference/code/ip.py:21-25); the transcript is a few KB
// per round and inherently sequential, so it stays on the host.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <vector>

namespace bfs {

inline void keccak_f1600(uint64_t s[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL, 0x000000000000808bULL,
        0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008aULL, 0x0000000000000088ULL,
        0x0000000080008009ULL, 0x000000008000000aULL, 0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL,
        0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    for (int round = 0; round < 24; ++round) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; ++x) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; ++x) {
            uint64_t r = c[(x + 1) % 5];
            d[x] = c[(x + 4) % 5] ^ ((r << 1) | (r >> 63));
        }
        for (int i = 0; i < 25; ++i) s[i] ^= d[i % 5];
        for (int x = 0; x < 5; ++x)
            for (int y = 0; y < 5; ++y) {
                int i = x + 5 * y, r = ROT[i];
                uint64_t v = s[i];
                v = r ? ((v << r) | (v >> (64 - r))) : v;
                b[y + 5 * ((2 * x + 3 * y) % 5)] = v;
            }
        for (int y = 0; y < 5; ++y)
            for (int x = 0; x < 5; ++x) s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        s[0] ^= RC[round];
    }
}

// SHAKE256: rate 136 bytes, domain suffix 0x1F
inline void shake256(const void* data, size_t len, unsigned char* out, size_t outlen) {
    uint64_t s[25];
    memset(s, 0, sizeof s);
    const size_t rate = 136;
    const unsigned char* p = (const unsigned char*)data;
    unsigned char block[136];
    while (len >= rate) {
        for (size_t i = 0; i < rate / 8; ++i) { uint64_t w; memcpy(&w, p + 8 * i, 8); s[i] ^= w; }
        keccak_f1600(s);
        p += rate;
        len -= rate;
    }
    memset(block, 0, rate);
    if (len) memcpy(block, p, len);
    block[len] ^= 0x1F;
    block[rate - 1] ^= 0x80;
    for (size_t i = 0; i < rate / 8; ++i) { uint64_t w; memcpy(&w, block + 8 * i, 8); s[i] ^= w; }
    keccak_f1600(s);
    size_t off = 0;
    while (off < outlen) {
        size_t take = outlen - off < rate ? outlen - off : rate;
        memcpy(out + off, s, take);
        off += take;
        if (off < outlen) keccak_f1600(s);
    }
}

}  // namespace bfs
