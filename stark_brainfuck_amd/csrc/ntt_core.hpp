// ntt_core.hpp -- tiled mixed-radix Goldilocks NTT, the device-side body of bfs_gl_ntt().
//
// Replaces the recursive object-list transform of the reference:
//   ntt()   /root/reference/code/ntt.py:4-23     out[k] = sum_j v[j] w^(jk), natural order in and out
//   intt()  ntt.py:26-42                          same with w^-1, then * n^-1 (folded into a twiddle table here)
//   Polynomial.scale()  univariate.py:168-169     c_j * s^j, fused into the first pass' load (coset evaluation)
//   zero padding of fast_coset_evaluate  ntt.py:164-168  fused: inputs j >= n_in read as 0, never materialised
//
// Algorithm (DESIGN.md "NTT"): n = n_1 * n_2 * ... * n_m (m <= 4 HBM passes, n_t = 2^S_t, S_t <= 12).
// Pass t transforms digit t of the input index (most significant first).  After pass t the slot that held input
// digit j_t holds output digit k_t, so after all passes slot (p_1|p_2|...|p_m) holds X[p_1 + n_1 p_2 + ...]; the
// last pass writes straight to that natural-order index.  Between passes the element is multiplied by
// w_{N_{t+1}}^(j_{t+1} * K_t) (N_t = n_1..n_t, K_t = k_1 + n_1 k_2 + ...), applied while pass t+1 loads.
// Inside a pass the 2^S-point column transform is again split into <= 3 register stages of radix 2^B <= 16:
// every thread holds 16 elements in VGPRs, runs a decimation-in-frequency network whose twiddles are powers
// of two (any primitive 16th root of unity in this field is 2^(12u), u odd), multiplies by the inner twiddle
// from a 4096-entry table and exchanges through LDS once per stage.
//
// The stage bodies are pure functions of (thread id, block id, LDS pointer) and compile for the host as well,
// which is how tests/test_emulation.py checks the index arithmetic without a GPU (test infrastructure only:
// the product never runs them on the CPU).
#pragma once
#include "gl.hpp"

namespace bfs {

// 2^k mod p for 0 <= k < 192, evaluated at compile time
constexpr u64 cx_mulmod(u64 a, u64 b) { return (u64)(((u128)a * b) % GL_P); }
constexpr u64 cx_pow2(int k) {
    u64 r = 1;
    for (int i = 0; i < k; ++i) r = cx_mulmod(r, 2);
    return r;
}

// x * 2^K mod p.  Written as a multiplication by the compile-time constant 2^K mod p: hipcc drops the partial products of
// the constant's zero limbs (1-2 v_mad_u64_u32 instead of 4), which measured fewer instructions than an explicit
// shift-and-fold formulation (tools/microbench, DESIGN.md "modular arithmetic").
template <int K>
BFS_HD u64 mul_pow2(u64 x) {
    if constexpr (K == 0) {
        return x;
    } else {
        constexpr u64 c = cx_pow2(K);
        return gl_mul(x, c);
    }
}

// one level of the radix-Q decimation-in-frequency network, twiddle w_Q = 2^(192/Q)
template <int Q, int I>
BFS_HD void dif_level(u64* x) {
    if constexpr (I < Q / 2) {
        u64 a = x[I], b = x[I + Q / 2];
        x[I] = gl_add(a, b);
        x[I + Q / 2] = mul_pow2<(192 / Q) * I>(gl_sub(a, b));
        dif_level<Q, I + 1>(x);
    }
}

// Q-point NTT with root 2^(192/Q); result for output index k is left in x[bitrev(k)]
template <int Q>
BFS_HD void dif(u64* x) {
    if constexpr (Q >= 2) {
        dif_level<Q, 0>(x);
        dif<Q / 2>(x);
        dif<Q / 2>(x + Q / 2);
    }
}

BFS_HD u32 bitrev(u32 v, int bits) {
    u32 r = 0;
    BFS_UNROLL
    for (int i = 0; i < 4; ++i)
        if (i < bits) r |= ((v >> i) & 1u) << (bits - 1 - i);
    return r;
}

struct NttTables {
    const u64* w_lo;       // w^i,              i < 2^lo_bits
    const u64* w_hi;       // w^(i * 2^lo_bits), i < 2^(log_n - lo_bits)
    u32 lo_bits;
    u32 t_in_log;          // inner table has 2^t_in_log entries: Omega^i, Omega = w^(n / 2^t_in_log)
    const u64* t_in;
    const u64* t_in_last;  // Omega^i * post_scale (used for the last inner twiddle of the final pass)
    const u64* s_lo;       // coset shift s: s^i, i < 2^lo_bits (null when no coset)
    const u64* s_hi;       // s^(i * 2^lo_bits)
};

enum { PASS_COLUMN = 0, PASS_FINAL = 1 };

struct PassArgs {
    const u64* in;
    u64* out;
    u64 in_batch_stride, out_batch_stride;
    u64 n_in;            // valid input elements per transform (pass 0 only); the rest reads as zero
    u32 log_n;
    u32 mode;            // PASS_COLUMN | PASS_FINAL
    u32 logC;            // columns per tile
    u32 pass_index;      // t, 0-based
    u32 npass;
    u32 pass_bits;       // S_v packed one byte per pass (no array: kernel-argument arrays indexed at run time go to scratch)
    u32 uinv;            // u^-1 mod 16 where w^(n/16) = 2^(12u)
    u32 has_coset;       // pass 0: multiply input j by s^j
    u64 coset_delta;     // s^(stride of the stage-1 register index)
    u64 post_scale;      // multiplied in at the final store when the final pass has a single stage
    u32 pad_shift;       // LDS padding: phys = lin + (lin >> pad_shift) * pad_amount
    u32 pad_amount;
    u32 lds_cmajor;      // LDS tile layout: 0 = [row][column], 1 = [column][row] (final pass of a multi-pass plan)
    u32 wide_load;       // 16-byte paired-lane loads allowed (pointer / stride alignment checked on the host)
    u32 wide_store;
    NttTables tb;
};

BFS_HD u64 tw_pow(const u64* lo, const u64* hi, u32 lo_bits, u64 e) {
    u64 a = lo[e & ((1ull << lo_bits) - 1)];
    u64 b = hi[e >> lo_bits];
    return gl_mul(a, b);
}

// exchange a 64-bit value with the neighbouring lane (lane ^ 1): two DPP moves, no LDS traffic.
// Used to turn two 8-byte accesses of adjacent lanes into one 16-byte access per lane (HBM efficiency of the tile
// passes: 4.4 TB/s with 16 B per lane against 3.2 TB/s with 8 B per lane, tools/microbench/mem.hip).
BFS_HD u64 lane_swap1(u64 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    lo = (u32)__builtin_amdgcn_update_dpp((int)lo, (int)lo, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    hi = (u32)__builtin_amdgcn_update_dpp((int)hi, (int)hi, 0xB1, 0xF, 0xF, true);
    return ((u64)hi << 32) | lo;
#else
    return v;
#endif
}

struct alignas(16) U64x2 {
    u64 x, y;
};

BFS_HD u32 lds_phys(const PassArgs& a, u32 lin) { return lin + (lin >> a.pad_shift) * a.pad_amount; }

// physical LDS word of tile element (row r, column c); S = log2(rows)
template <int S>
BFS_HD u32 lds_addr(const PassArgs& a, u32 r, u32 c) {
    return lds_phys(a, a.lds_cmajor ? ((c << S) + r) : ((r << a.logC) + c));
}

BFS_HD u32 perm_digit(u32 m, int bits, u32 uinv) { return (bitrev(m, bits) * uinv) & ((1u << bits) - 1); }

// geometry shared by the stages of one tile
struct TileGeom {
    u64 in_base, out_base;  // element offsets of this transform (batch)
    u64 base;               // COLUMN: h * n_t * L
    u64 L;                  // COLUMN: lower stride
    u64 c0;                 // COLUMN: first column; FINAL: first p1
    u64 kstep;              // COLUMN: exponent step of the inter-pass twiddle (w^(kstep * row))
    u64 mid;                // FINAL: fixed middle slot digits
    u64 kmid;               // FINAL: their digit-reversed value
    u32 mid_bits, n1_bits;
};

// digit-reverse the slot digits p_first..p_last (p_first most significant in h) into K = p_first + n_first*(...)
BFS_HD u32 pass_bits_of(u32 packed, int v) { return (packed >> (8 * v)) & 0xFFu; }

BFS_HD u64 digit_reverse(u64 h, u32 packed_bits, int first, int last) {
    u64 K = 0;
    bool any = false;
    for (int v = last; v >= first; --v) {
        const u32 b = pass_bits_of(packed_bits, v);
        u64 p = h & ((1ull << b) - 1);
        h >>= b;
        K = any ? p + (K << b) : p;
        any = true;
    }
    return K;
}

template <int S>
BFS_HD TileGeom tile_geom(const PassArgs& a, u32 bid_x, u32 bid_y) {
    TileGeom g{};
    g.in_base = (u64)bid_y * a.in_batch_stride;
    g.out_base = (u64)bid_y * a.out_batch_stride;
    const u64 n = 1ull << a.log_n;
    if (a.mode == PASS_COLUMN) {
        u32 done = 0;
        for (u32 v = 0; v <= a.pass_index; ++v) done += pass_bits_of(a.pass_bits, (int)v);
        g.L = n >> done;
        u64 nl = g.L >> a.logC;
        u64 h = bid_x / nl, lch = bid_x % nl;
        g.c0 = lch << a.logC;
        g.base = h * (g.L << S);
        u64 K = a.pass_index ? digit_reverse(h, a.pass_bits, 0, (int)a.pass_index - 1) : 0;
        g.kstep = (K * (n >> done)) & (n - 1);
    } else {
        g.n1_bits = pass_bits_of(a.pass_bits, 0);
        if (a.npass > 1) {
            u32 mb = 0;
            for (u32 v = 1; v + 1 < a.npass; ++v) mb += pass_bits_of(a.pass_bits, (int)v);
            g.mid_bits = mb;
            u64 chunks = (1ull << g.n1_bits) >> a.logC;
            g.c0 = (bid_x % chunks) << a.logC;
            g.mid = bid_x / chunks;
            g.kmid = mb ? digit_reverse(g.mid, a.pass_bits, 1, (int)a.npass - 2) : 0;
        }
    }
    return g;
}

// load address / inter-pass exponent for (row r, column c)
template <int S>
BFS_HD u64 in_index(const PassArgs& a, const TileGeom& g, u32 r, u32 c) {
    if (a.mode == PASS_COLUMN) return g.base + (u64)r * g.L + g.c0 + c;
    if (a.npass == 1) return r;
    u64 p1 = g.c0 + c;
    return (((p1 << g.mid_bits) + g.mid) << S) + r;
}

template <int S>
BFS_HD u64 out_index(const PassArgs& a, const TileGeom& g, u32 kpass, u32 c) {
    if (a.mode == PASS_COLUMN) return g.base + (u64)kpass * g.L + g.c0 + c;
    if (a.npass == 1) return kpass;
    u64 K = (g.c0 + c) + (g.kmid << g.n1_bits);
    return K + ((u64)kpass << (a.log_n - S));
}

template <int B1, int B2, int B3>
struct TileCfg {
    static constexpr int S = B1 + B2 + B3;
    static constexpr int SH1 = B2 + B3, SH2 = B3;
    static constexpr int U = (B2 == 0) ? 1 : (B3 == 0 ? 2 : 3);
};

// store the 16 registers of the last stage.  f_lo / f_mid are the already-final lower digits of k_pass.
template <int B1, int B2, int B3, int BQ>
BFS_HD void final_store(const PassArgs& a, const TileGeom& g, const u64* x, u32 sub, u32 klow, int kshift, u32 c, u32 G) {
    typedef TileCfg<B1, B2, B3> Cfg;
    u64* out = a.out + g.out_base;
    constexpr int Q = 1 << BQ;
    const bool scale = (Cfg::U == 1) && a.post_scale != 1;
    u64 v[Q];
    BFS_UNROLL
    for (int m = 0; m < Q; ++m) v[m] = scale ? gl_mul(x[sub * Q + m], a.post_scale) : x[sub * Q + m];
#if defined(__HIP_DEVICE_COMPILE__)
    if (Q >= 2 && a.wide_store && a.logC >= 1) {
        // lanes 2i / 2i+1 hold adjacent columns: the even lane stores both columns of outputs m < Q/2, the odd lane
        // those of m >= Q/2, as one 16-byte store each
        constexpr int H = Q / 2 ? Q / 2 : 1;
        const u32 par = G & 1;
        BFS_UNROLL
        for (int mm = 0; mm < H; ++mm) {
            u64 keep = par ? v[H + mm] : v[mm];
            u64 send = par ? v[mm] : v[H + mm];
            u64 recv = lane_swap1(send);
            u32 kd = perm_digit((u32)(par ? H + mm : mm), BQ, a.uinv);
            u32 kpass = klow + (kd << kshift);
            U64x2 pr;
            pr.x = par ? recv : keep;
            pr.y = par ? keep : recv;
            *reinterpret_cast<U64x2*>(out + out_index<Cfg::S>(a, g, kpass, c & ~1u)) = pr;
        }
        return;
    }
#endif
    BFS_UNROLL
    for (int m = 0; m < Q; ++m) {
        u32 kd = perm_digit((u32)m, BQ, a.uinv);
        u32 kpass = klow + (kd << kshift);
        out[out_index<Cfg::S>(a, g, kpass, c)] = v[m];
    }
}

// raw 16-byte pieces of one tile as fetched by a thread (paired-lane wide loads); lets the kernel issue the loads of
// the NEXT tile before it starts computing on the current one (software prefetch: HBM latency hides under the VALU work)
struct RawTile {
    U64x2 pr[8];
};

template <int B1, int B2, int B3>
BFS_HD bool wide_load_possible(const PassArgs& a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return B1 == 4 && a.wide_load && (a.mode == PASS_COLUMN ? a.logC >= 1 : TileCfg<B1, B2, B3>::SH1 >= 1);
#else
    return false;
#endif
}

// issue the global loads of tile (bid_x, bid_y) for this thread; only valid when wide_load_possible()
template <int B1, int B2, int B3>
BFS_HD void ntt_prefetch(const PassArgs& a, u32 tid, u32 bid_x, u32 bid_y, RawTile& raw) {
    typedef TileCfg<B1, B2, B3> Cfg;
    const TileGeom g = tile_geom<Cfg::S>(a, bid_x, bid_y);
    const u64* in = a.in + g.in_base;
    const u32 G = tid;
    u32 o, c;
    if (a.mode == PASS_COLUMN) { c = G & ((1u << a.logC) - 1); o = G >> a.logC; }
    else { o = G & ((1u << Cfg::SH1) - 1); c = G >> Cfg::SH1; }
    const u32 par = G & 1;
    const u32 oe = (a.mode == PASS_COLUMN) ? o : (o & ~1u);
    const u32 ce = (a.mode == PASS_COLUMN) ? (c & ~1u) : c;
    BFS_UNROLL
    for (int dd = 0; dd < 8; ++dd) {
        u32 d = par ? (u32)(8 + dd) : (u32)dd;
        u64 idx = in_index<Cfg::S>(a, g, (d << Cfg::SH1) | oe, ce);
        U64x2 pr;
        if (a.pass_index > 0 || idx + 1 < a.n_in) pr = *reinterpret_cast<const U64x2*>(in + idx);
        else { pr.x = idx < a.n_in ? in[idx] : 0; pr.y = 0; }
        raw.pr[dd] = pr;
    }
}

// same for plain 8-byte loads (one element per lane and load): raw.pr[d/2].{x,y} = element d of this thread
template <int B1, int B2, int B3>
BFS_HD void ntt_prefetch_narrow(const PassArgs& a, u32 tid, u32 bid_x, u32 bid_y, RawTile& raw) {
    typedef TileCfg<B1, B2, B3> Cfg;
    const TileGeom g = tile_geom<Cfg::S>(a, bid_x, bid_y);
    const u64* in = a.in + g.in_base;
    const u32 G = tid;
    u32 o, c;
    if (a.mode == PASS_COLUMN) { c = G & ((1u << a.logC) - 1); o = G >> a.logC; }
    else { o = G & ((1u << Cfg::SH1) - 1); c = G >> Cfg::SH1; }
    BFS_UNROLL
    for (int d = 0; d < 16; ++d) {
        u64 idx = in_index<Cfg::S>(a, g, ((u32)d << Cfg::SH1) | o, c);
        u64 v = (a.pass_index > 0 || idx < a.n_in) ? in[idx] : 0;
        if (d & 1) raw.pr[d / 2].y = v; else raw.pr[d / 2].x = v;
    }
}

// ---- stage 1: global load (+ coset / inter-pass twiddle), first radix, inner twiddle, LDS write (or final store)
// tw: dense inner-twiddle table for the stage 1 -> 2 exchange (2^(B1+B2) entries, in LDS on the GPU)
template <int B1, int B2, int B3, int PRE /* 0 load here, 1 wide prefetched, 2 narrow prefetched */>
BFS_HD void ntt_stage1(const PassArgs& a, u64* smem, const u64* tw, u32 tid, u32 bid_x, u32 bid_y, const RawTile& pre) {
    typedef TileCfg<B1, B2, B3> Cfg;
    constexpr int Q = 1 << B1, SG = 16 / Q;
    const u32 W = ((1u << Cfg::S) << a.logC) >> 4;
    const TileGeom g = tile_geom<Cfg::S>(a, bid_x, bid_y);
    const u64* in = a.in + g.in_base;
    const u64 n = 1ull << a.log_n;
    const bool twiddle = a.pass_index > 0;
    u64 x[16];
    BFS_UNROLL
    for (int s = 0; s < SG; ++s) {
        u32 G = (u32)s * W + tid;
        u32 o, c;
        if (a.mode == PASS_COLUMN) { c = G & ((1u << a.logC) - 1); o = G >> a.logC; }
        else { o = G & ((1u << Cfg::SH1) - 1); c = G >> Cfg::SH1; }
        if constexpr (PRE == 2) {
            BFS_UNROLL
            for (int d = 0; d < 16; ++d) x[d] = (d & 1) ? pre.pr[d / 2].y : pre.pr[d / 2].x;
        } else if constexpr (PRE == 1) {
            // paired lanes (adjacent columns in a column pass, adjacent rows in the final pass) fetched 16 bytes each:
            // the even lane the first half of the register index d, the odd lane the second half; now they swap
            const u32 par = G & 1;
            BFS_UNROLL
            for (int dd = 0; dd < 8; ++dd) {
                U64x2 pr = pre.pr[dd];
                u64 keep = par ? pr.y : pr.x;
                u64 recv = lane_swap1(par ? pr.x : pr.y);
                x[dd] = par ? recv : keep;
                x[8 + dd] = par ? keep : recv;
            }
        } else {
            BFS_UNROLL
            for (int d = 0; d < Q; ++d) {
                u32 r = ((u32)d << Cfg::SH1) | o;
                u64 idx = in_index<Cfg::S>(a, g, r, c);
                x[s * Q + d] = (a.pass_index > 0 || idx < a.n_in) ? in[idx] : 0;
            }
        }
#ifdef BFS_ABL_NO_CHAIN
        if (false) {
#else
        if (twiddle || a.has_coset) {
#endif
            u64 gam, del;
            if (twiddle) {
                u64 ks = (a.mode == PASS_COLUMN) ? g.kstep : ((g.c0 + c) + (g.kmid << g.n1_bits));
                gam = tw_pow(a.tb.w_lo, a.tb.w_hi, a.tb.lo_bits, ((u64)o * ks) & (n - 1));
                del = tw_pow(a.tb.w_lo, a.tb.w_hi, a.tb.lo_bits, (ks << Cfg::SH1) & (n - 1));
            } else {
                gam = tw_pow(a.tb.s_lo, a.tb.s_hi, a.tb.lo_bits, in_index<Cfg::S>(a, g, o, c));
                del = a.coset_delta;
            }
            u64 f = gam;
            BFS_UNROLL
            for (int d = 0; d < Q; ++d) {
                x[s * Q + d] = gl_mul(x[s * Q + d], f);
                f = gl_mul(f, del);
            }
        }
#ifndef BFS_ABL_NO_DIF
        dif<Q>(x + s * Q);
#endif
        if constexpr (Cfg::U == 1) {
            final_store<B1, B2, B3, B1>(a, g, x, (u32)s, 0, 0, c, G);
        } else {
            u32 i2 = o >> B3;
            BFS_UNROLL
            for (int m = 0; m < Q; ++m) {
                u32 k1 = perm_digit((u32)m, B1, a.uinv);
                u32 e = (i2 * k1) & ((1u << (B1 + B2)) - 1);
#ifdef BFS_ABL_NO_INNER
                u64 v = x[s * Q + m] + e;
#else
                u64 v = gl_mul(x[s * Q + m], tw[e]);
#endif
                u32 r = (k1 << Cfg::SH1) | o;
                smem[lds_addr<Cfg::S>(a, r, c)] = v;
            }
        }
    }
}

// ---- stage 2: LDS read, second radix, (inner twiddle + LDS write) or final store
template <int B1, int B2, int B3>
BFS_HD void ntt_stage2(const PassArgs& a, u64* smem, u32 tid, u32 bid_x, u32 bid_y) {
    typedef TileCfg<B1, B2, B3> Cfg;
    if constexpr (B2 > 0) {
        constexpr int Q = 1 << B2, SG = 16 / Q;
        const u32 W = ((1u << Cfg::S) << a.logC) >> 4;
        const TileGeom g = tile_geom<Cfg::S>(a, bid_x, bid_y);
        u64 x[16];
        BFS_UNROLL
    for (int s = 0; s < SG; ++s) {
            u32 G = (u32)s * W + tid;
            u32 c = G & ((1u << a.logC) - 1);
            u32 rest = G >> a.logC;
            u32 f1 = rest & ((1u << B1) - 1), f3 = rest >> B1;
            BFS_UNROLL
            for (int d = 0; d < Q; ++d) {
                u32 r = (f1 << Cfg::SH1) | ((u32)d << Cfg::SH2) | f3;
                x[s * Q + d] = smem[lds_addr<Cfg::S>(a, r, c)];
            }
#ifndef BFS_ABL_NO_DIF
            dif<Q>(x + s * Q);
#endif
            if constexpr (Cfg::U == 2) {
                final_store<B1, B2, B3, B2>(a, g, x, (u32)s, f1, B1, c, G);
            } else {
                const u64* tab = (a.mode == PASS_FINAL) ? a.tb.t_in_last : a.tb.t_in;
                BFS_UNROLL
                for (int m = 0; m < Q; ++m) {
                    u32 k2 = perm_digit((u32)m, B2, a.uinv);
                    u32 e = (f3 * (f1 + (k2 << B1))) & ((1u << Cfg::S) - 1);
                    u64 v = gl_mul(x[s * Q + m], tab[(u64)e << (a.tb.t_in_log - Cfg::S)]);
                    u32 r = (f1 << Cfg::SH1) | (k2 << Cfg::SH2) | f3;
                    smem[lds_addr<Cfg::S>(a, r, c)] = v;
                }
            }
        }
    }
}

// ---- stage 3: LDS read, third radix, final store
template <int B1, int B2, int B3>
BFS_HD void ntt_stage3(const PassArgs& a, u64* smem, u32 tid, u32 bid_x, u32 bid_y) {
    typedef TileCfg<B1, B2, B3> Cfg;
    if constexpr (B3 > 0) {
        constexpr int Q = 1 << B3, SG = 16 / Q;
        const u32 W = ((1u << Cfg::S) << a.logC) >> 4;
        const TileGeom g = tile_geom<Cfg::S>(a, bid_x, bid_y);
        u64 x[16];
        BFS_UNROLL
    for (int s = 0; s < SG; ++s) {
            u32 G = (u32)s * W + tid;
            u32 c = G & ((1u << a.logC) - 1);
            u32 rest = G >> a.logC;
            u32 f1 = rest & ((1u << B1) - 1), f2 = rest >> B1;
            BFS_UNROLL
            for (int d = 0; d < Q; ++d) {
                u32 r = (f1 << Cfg::SH1) | (f2 << Cfg::SH2) | (u32)d;
                x[s * Q + d] = smem[lds_addr<Cfg::S>(a, r, c)];
            }
            dif<Q>(x + s * Q);
            final_store<B1, B2, B3, B3>(a, g, x, (u32)s, f1 + (f2 << B1), B1 + B2, c, G);
        }
    }
}

// elements of LDS a tile needs (including padding)
inline u32 tile_lds_elems(u32 S, u32 logC, u32 pad_shift, u32 pad_amount) {
    u32 T = (1u << S) << logC;
    return T + ((T - 1) >> pad_shift) * pad_amount + pad_amount;
}

}  // namespace bfs

namespace bfs {

// direct O(n^2) transform for n <= 8 (ntt.py:4-23 evaluated literally); one thread per output element
struct SmallArgs {
    const u64* in;
    u64* out;
    u64 in_batch_stride, out_batch_stride;
    u64 n_in;
    u32 log_n;
    u64 root, shift, post_scale;
};

BFS_HD void ntt_small_body(const SmallArgs& a, u32 k, u32 bid_y) {
    const u32 n = 1u << a.log_n;
    if (k >= n) return;
    const u64* in = a.in + (u64)bid_y * a.in_batch_stride;
    u64 wk = gl_pow(a.root, k);
    u64 acc = 0, wjk = 1, sj = 1;
    for (u32 j = 0; j < n; ++j) {
        u64 v = j < a.n_in ? in[j] : 0;
        acc = gl_add(acc, gl_mul(gl_mul(v, sj), wjk));
        wjk = gl_mul(wjk, wk);
        sj = gl_mul(sj, a.shift);
    }
    a.out[(u64)bid_y * a.out_batch_stride + k] = gl_mul(acc, a.post_scale);
}

}  // namespace bfs
