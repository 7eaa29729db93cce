// fri.hip -- the FRI prover loop on gfx950.  Replaces Fri.commit / Fri.query / Fri.query_last / Fri.prove of the
// reference (/root/reference/code/fri.py:91-199): per round a Merkle tree over the codeword (GPU), the root to the
// host, the Fiat-Shamir challenge from the transcript (host: pickle + SHAKE256, ip.py:21-22), the split-and-fold
// step (GPU, fri.py:127-128); then index sampling (fri.py:62-86) and one batched gather of every revealed leaf and
// authentication-path node.
#include <algorithm>
#include <chrono>
#include <map>
#include <unordered_map>
#include <vector>

#include "../../include/bfstark.h"
#include "blake2b.hpp"
#include "refpickle.hpp"
#include "runtime.hpp"

namespace bfs {

int merkle_build_xfe_launch(const u64* d_limbs, u64 limb_stride, u64 n, u64* d_nodes, hipStream_t stream, u64* root_out = nullptr, u64 seq = 0);
int ntt_power_tables(u64 root, u32 log_n, const u64** lo, const u64** hi, u32* lo_bits);

BFS_HD u64 gl_half(u64 x) { return (x >> 1) + ((x & 1) ? 0x7FFFFFFF80000001ULL : 0); }  // x / 2 mod p

// fri.py:127-128:  out[i] = 2^-1 * ((1 + alpha/x_i) * a + (1 - alpha/x_i) * b),  x_i = offset * omega^i
//                         = (a + b)/2 + alpha * (2^-1 * offset^-1 * omega^-i) * (a - b)
// winv_*: two-level powers of the ROUND-0 omega^-1 (exponent i << round_shift); scal = 2^-1 * offset_r^-1
__global__ void fri_fold_kernel(const u64* in, u64 in_stride, u64* out, u64 out_stride, u64 half, Xfe alpha, u64 scal,
                                const u64* winv_lo, const u64* winv_hi, u32 lo_bits, u32 round_shift) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += (u64)gridDim.x * blockDim.x) {
        Xfe a{{in[i], in[in_stride + i], in[2 * in_stride + i]}};
        Xfe b{{in[half + i], in[in_stride + half + i], in[2 * in_stride + half + i]}};
        u64 s = gl_mul(scal, tw_pow(winv_lo, winv_hi, lo_bits, i << round_shift));
        Xfe beta = xfe_scale(alpha, s);
        Xfe sum = xfe_add(a, b), diff = xfe_sub(a, b);
        Xfe prod = xfe_mul(beta, diff);
        out[i] = gl_add(gl_half(sum.c[0]), prod.c[0]);
        out[out_stride + i] = gl_add(gl_half(sum.c[1]), prod.c[1]);
        out[2 * out_stride + i] = gl_add(gl_half(sum.c[2]), prod.c[2]);
    }
}

// one request = `nwords` 64-bit words at base, base + stride, ...; requests and results live in pinned host memory that the
// GPU reads / writes directly (no copy commands: the openings of a proof are a few thousand scattered words)
struct GatherReq {
    const u64* base;
    u32 nwords, stride;
    u64 out_offset;
};
__global__ void gather_requests_kernel(const GatherReq* req, u32 count, u64* out) {
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const GatherReq r = req[i];
        for (u32 w = 0; w < r.nwords; ++w) out[r.out_offset + w] = r.base[(u64)w * r.stride];
    }
}

// wall-clock breakdown of the last bfs_fri_commit / bfs_fri_query on this thread (ms): see bfs_fri_last_timing
static thread_local double g_fri_timing[8] = {0, 0, 0, 0, 0, 0, 0, 0};
static inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct FriRound {
    const u64* cw = nullptr;  // limb-major codeword
    u64 stride = 0, length = 0;
    u64* nodes = nullptr;     // 2*length digests of 8 words
    unsigned char root[64];
};

// pinned, host-visible mailbox the tree kernel drops each round's root into (9 words: digest + sequence flag)
struct RootMailbox {
    u64* host = nullptr;
    u64* dev = nullptr;
    u64 seq = 0;
    int init() {
        if (host) return BFS_OK;
        BFS_TRY(host_alloc(16 * sizeof(u64), (void**)&host));       // pooled: hipHostFree per session stalls later dispatches
        memset(host, 0, 16 * sizeof(u64));
        BFS_HIP(hipHostGetDevicePointer((void**)&dev, host, 0));
        return BFS_OK;
    }
    ~RootMailbox() { if (host) (void)host_release(host); }
};

struct FriSession {
    std::vector<FriRound> rounds;
    void* block = nullptr;  // device allocation owned by the session (null when it borrows the cached workspace)
    bool use_workspace = false;
    RootMailbox mailbox;
    u32 log_n = 0;
    // (round, index) -> the element / tree-node object.  One object per key: the reference pushes the same Python object
    // again when an index recurs, and pickle memoises by identity.
    struct Key {
        u64 v;
        Key(u32 round, u64 index) : v(((u64)round << 48) | index) {}
        bool operator==(const Key& o) const { return v == o.v; }
    };
    // open-addressing table (linear probing, power-of-two capacity): a proof makes ~4 000 look-ups and insertions here, and
    // std::unordered_map's node allocations were 0.2 ms of a 1.9 ms proof
    struct KeyMap {
        std::vector<u64> keys;          // key.v + 1; 0 = empty slot
        std::vector<u32> slots;         // index into vals
        std::vector<rp::Ref> vals;      // in insertion order
        void reserve(size_t n) { size_t cap = 64; while (cap < 2 * n) cap <<= 1; if (cap > keys.size()) rehash(cap); vals.reserve(n); }
        void rehash(size_t cap) {
            std::vector<u64> k2(cap, 0);
            std::vector<u32> s2(cap, 0);
            for (size_t i = 0; i < keys.size(); ++i)
                if (keys[i]) { size_t j = slot_of(keys[i], cap); while (k2[j]) j = (j + 1) & (cap - 1); k2[j] = keys[i]; s2[j] = slots[i]; }
            keys.swap(k2); slots.swap(s2);
        }
        static size_t slot_of(u64 stored, size_t cap) { return (size_t)((stored * 0x9E3779B97F4A7C15ULL) >> 20) & (cap - 1); }
        // the slot of `key`: existing, or the empty one where it goes
        size_t find(const Key& key) const {
            size_t j = slot_of(key.v + 1, keys.size());
            while (keys[j] && keys[j] != key.v + 1) j = (j + 1) & (keys.size() - 1);
            return j;
        }
        bool count(const Key& key) const { return !keys.empty() && keys[find(key)] != 0; }
        rp::Ref& operator[](const Key& key) {
            if (2 * (vals.size() + 1) > keys.size()) rehash(keys.empty() ? 64 : 2 * keys.size());
            const size_t j = find(key);
            if (!keys[j]) { keys[j] = key.v + 1; slots[j] = (u32)vals.size(); vals.emplace_back(); }
            return vals[slots[j]];
        }
    };
    KeyMap elements, nodes;
    std::vector<uint64_t> last_handles;
    hipStream_t block_stream = nullptr;
    const u64* round0_nodes = nullptr;     // a tree over the input codeword that the caller has already built (bfs_fri_session_round0_tree)
    unsigned char round0_root[64];
    ~FriSession() { if (block) (void)device_release(block, block_stream); }
};

static u32 fri_num_rounds(u64 length, u32 expansion) {  // fri.py:54-60
    u32 r = 0;
    while (length > expansion) { length /= 2; ++r; }
    return r;
}

int fri_commit(FriSession& S, rp::Transcript& ps, const u64* d_cw, u64 stride, u32 log_n, u64 offset, u64 omega,
               u32 expansion, hipStream_t stream) {
    const double t_begin = now_ms();
    const u64 N = 1ull << log_n;
    const u32 R = fri_num_rounds(N, expansion);
    if (R < 1) { set_error("cannot do FRI with less than one round"); return BFS_ERR_BAD_ARG; }
    if (gl_pow(omega, N) != 1 || (log_n && gl_pow(omega, N / 2) == 1)) {
        set_error("error in commit: omega does not have the right order!");
        return BFS_ERR_NOT_ROOT;
    }
    S.log_n = log_n;
    S.rounds.assign(R, FriRound());
    // one allocation: nodes of every round + codewords of rounds >= 1
    size_t words = 0;
    for (u32 r = 0; r < R; ++r) words += (size_t)16 * (N >> r) + (r ? (size_t)3 * (N >> r) : 0);
    u64* p = nullptr;
    if (S.use_workspace) {
        void* w = nullptr;
        BFS_TRY(workspace(4, words * sizeof(u64), stream, &w));   // bfs_fri_prove: nothing outlives the call
        p = (u64*)w;
    } else {
        if (S.block) { (void)device_release(S.block, S.block_stream); S.block = nullptr; }
        BFS_TRY(device_alloc(words * sizeof(u64), stream, &S.block));
        S.block_stream = stream;
        p = (u64*)S.block;
    }
    BFS_TRY(S.mailbox.init());
    for (u32 r = 0; r < R; ++r) {
        FriRound& fr = S.rounds[r];
        fr.length = N >> r;
        fr.nodes = p; p += 16 * fr.length;
        if (r == 0) { fr.cw = d_cw; fr.stride = stride; }
        else { fr.cw = p; fr.stride = fr.length; p += 3 * fr.length; }
    }
    const u64 *winv_lo = nullptr, *winv_hi = nullptr;
    u32 lo_bits = 0;
    BFS_TRY(ntt_power_tables(gl_inv(omega), log_n, &winv_lo, &winv_hi, &lo_bits));
    const u64 half_inv = gl_inv(2);
    u64 g = offset;
    constexpr u64 FRI_FOLD_IN_LEAVES_MIN = 16384;     // = FRI_FUSED_MAX: every round >= 1 folds inside its leaf kernel (must exceed QUAD_LEAVES_MAX)
    FriFoldArgs pending{};                         // the fold that produces round r's codeword, when round r runs fused
    pending.in = nullptr;
    // Fiat-Shamir look-ahead: with a long transcript in front (a STARK proof: tens of KB), helper threads absorb the SHAKE256 prefix of
    // every coming challenge now (Transcript::Lookahead); rounds 1 .. R-2 push a root and draw a challenge
    rp::Transcript::Lookahead look;
    static const bool lookahead_on = getenv("BFS_FRI_LOOKAHEAD") == nullptr || atoi(getenv("BFS_FRI_LOOKAHEAD")) != 0;
    static const bool trace = getenv("BFS_FRI_TRACE") != nullptr;      // development aid: host timeline of every round on stderr
    // (begun behind round 1's launch, below: laying out the R - 2 pickles takes ~50 us with a STARK proof's openings in front, and
    //  nothing is pushed before round 1's root -- so the GPU hashes round 1 meanwhile instead of waiting for it)
    bool looking = false, look_begun = false;
    auto begin_lookahead = [&]() {
        if (look_begun) return;
        look_begun = true;
        const double t_look = trace ? now_ms() : 0;
        looking = lookahead_on && R >= 3 && ps.lookahead_begin(look, R - 2);
        if (trace) fprintf(stderr, "fri look-ahead %s: %.1f us, %zu objects in front\n", looking ? "on" : "off", 1e3 * (now_ms() - t_look), ps.objects.size());
    };
    for (u32 r = 0; r < R; ++r) {
        FriRound& fr = S.rounds[r];
        unsigned char seed[32];
        bool have_seed = false, speculating = false;
        const double t_round = trace ? now_ms() : 0;
        double t_launched = 0, t_absorbed = 0, t_root = 0;
        rp::Transcript::Speculation speculation;
        const bool fused = fr.length >= 2 && fr.length <= FRI_FUSED_MAX && !(r == 0 && S.round0_nodes);
        if (fused) {
            // small codeword: fold (of the previous round) + leaves + subtrees in one launch, root through the mailbox
            const u64 seq = ++S.mailbox.seq;
            BFS_TRY(fri_round_fused_launch(pending, (u64*)fr.cw, fr.stride, fr.length, fr.nodes, stream, S.mailbox.dev, seq));
            pending.in = nullptr;
            if (trace) t_launched = now_ms();
            if (r >= 1) begin_lookahead();
            if (r + 1 < R) {
                if (r == 0) { ps.fiat_shamir(ps.objects.size(), seed, 32); have_seed = true; }
                else if (!looking) { ps.speculate(speculation); speculating = true; }
            }
            if (trace) t_absorbed = now_ms();
            volatile u64* flag = S.mailbox.host + 8;
            u64 spins = 0;
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
                if (++spins > (1ull << 22)) {
                    BFS_HIP(hipStreamSynchronize(stream));
                    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) { set_error("root mailbox was not written"); return BFS_ERR_HIP; }
                    break;
                }
            }
            memcpy(fr.root, S.mailbox.host, 64);
        } else if (r == 0 && S.round0_nodes) {
            fr.nodes = (u64*)S.round0_nodes;          // the STARK prover has just committed to this very codeword (brainfuck_stark.py:301 / fri.py:108)
            memcpy(fr.root, S.round0_root, 64);
        } else if (fr.length >= 2) {
            // the tree kernel writes the root straight into pinned host memory; poll the sequence flag instead of
            // paying a copy command + stream synchronisation per round
            const u64 seq = ++S.mailbox.seq;
            if (pending.in != nullptr) {               // the leaf kernel folds the previous round's codeword on the way
                BFS_TRY(merkle_build_xfe_fold_launch(pending, (u64*)fr.cw, fr.stride, fr.length, fr.nodes, stream, S.mailbox.dev, seq));
                pending.in = nullptr;
            } else {
                BFS_TRY(merkle_build_xfe_launch(fr.cw, fr.stride, fr.length, fr.nodes, stream, S.mailbox.dev, seq));  // fri.py:108
            }
            // While the GPU hashes: the next challenge is SHAKE256 of the WHOLE transcript including this root (fri.py:112-120), tens
            // of KB -- as long as the tree kernels of the late rounds.  Everything in front of the root's 64 bytes is known already,
            // so the sponge absorbs it now and only the last block or two wait for the root.
            if (trace) t_launched = now_ms();
            if (r >= 1) begin_lookahead();
            if (r + 1 < R) {
                if (r == 0) { ps.fiat_shamir(ps.objects.size(), seed, 32); have_seed = true; }
                else if (!looking) { ps.speculate(speculation); speculating = true; }
            }
            if (trace) t_absorbed = now_ms();
            volatile u64* flag = S.mailbox.host + 8;
            u64 spins = 0;
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
                if (++spins > (1ull << 22)) {   // ~ms: fall back to a real synchronisation (also surfaces kernel errors)
                    BFS_HIP(hipStreamSynchronize(stream));
                    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) { set_error("root mailbox was not written"); return BFS_ERR_HIP; }
                    break;
                }
            }
            memcpy(fr.root, S.mailbox.host, 64);
        } else {
            BFS_TRY(merkle_build_xfe_launch(fr.cw, fr.stride, fr.length, fr.nodes, stream));
            if (r >= 1) begin_lookahead();
            BFS_HIP(hipMemcpyAsync(fr.root, fr.nodes + 8, 64, hipMemcpyDeviceToHost, stream));
            BFS_HIP(hipStreamSynchronize(stream));
        }
        if (trace) t_root = now_ms();
        if (speculating) ps.resolve(speculation, fr.root, seed, 32);          // the placeholder pushed by speculate() becomes the root
        else if (r > 0 && looking) { ps.lookahead_next(look, fr.root, r + 1 < R ? seed : nullptr, 32); have_seed = r + 1 < R; }
        else if (r > 0) ps.objects.push_back(rp::mk_bytes(fr.root, 64));      // fri.py:112-113
        if (trace)
            fprintf(stderr, "fri round %2u  n %8llu  launch %6.1f us  absorb %6.1f us  wait %6.1f us  finish %6.1f us\n", r, (unsigned long long)fr.length,
                    1e3 * (t_launched - t_round), 1e3 * (t_absorbed - t_launched), 1e3 * (t_root - t_absorbed), 1e3 * (now_ms() - t_root));
        if (r == R - 1) break;                                                // fri.py:116-117
        if (!speculating && !have_seed) ps.fiat_shamir(ps.objects.size(), seed, 32);   // fri.py:120
        Xfe alpha = rp::sample_xfe(seed, 32);
        FriRound& nx = S.rounds[r + 1];
        const u64 half = fr.length / 2;
        if (half >= 2 && (half <= FRI_FUSED_MAX || half > FRI_FOLD_IN_LEAVES_MIN)) {
            // the next round folds while it builds its tree (fri_round_quad_kernel, or merkle_leaves_xfe_fold_kernel for large rounds)
            pending = FriFoldArgs{fr.cw, fr.stride, half, alpha, gl_mul(half_inv, gl_inv(g)), winv_lo, winv_hi, lo_bits, r};
        } else {
            u32 grid = (u32)((half + 255) / 256);
            if (grid > 4096) grid = 4096;
            hipLaunchKernelGGL(fri_fold_kernel, dim3(grid), dim3(256), 0, stream, fr.cw, fr.stride, (u64*)nx.cw, nx.stride, half, alpha,
                               gl_mul(half_inv, gl_inv(g)), winv_lo, winv_hi, lo_bits, r);
            BFS_HIP(hipGetLastError());
        }
        g = gl_sqr(g);  // fri.py:130-131 (omega is squared implicitly through round_shift)
    }
    g_fri_timing[0] = now_ms() - t_begin;   // rounds: trees, roots, challenges, folds
    // fri.py:134: the last codeword goes into the transcript as a list of element objects
    FriRound& last = S.rounds[R - 1];
    std::vector<u64> host(3 * last.length);
    if (last.stride == last.length) {
        BFS_HIP(hipMemcpyAsync(host.data(), last.cw, 3 * last.length * sizeof(u64), hipMemcpyDeviceToHost, stream));
    } else {
        for (int k = 0; k < 3; ++k)
            BFS_HIP(hipMemcpyAsync(host.data() + k * last.length, last.cw + k * last.stride, last.length * sizeof(u64), hipMemcpyDeviceToHost, stream));
    }
    BFS_HIP(hipStreamSynchronize(stream));
    std::vector<rp::Ref> items;
    for (u64 j = 0; j < last.length; ++j) {
        u64 l[3] = {host[j], host[last.length + j], host[2 * last.length + j]};
        rp::Ref e = ps.world.xfe_compact(l);
        S.elements[FriSession::Key(R - 1, j)] = e;
        items.push_back(e);
    }
    ps.objects.push_back(rp::mk_list(items));
    g_fri_timing[1] = now_ms() - t_begin - g_fri_timing[0];   // last codeword to the host
    return BFS_OK;
}

// fri.py:62-86
static int sample_indices(const unsigned char seed[32], u64 size, u64 reduced_size, u32 number, std::vector<u64>& out) {
    if (number > reduced_size) {
        set_error("cannot sample more indices than available in last codeword; requested: %u, available: %llu", number, (unsigned long long)reduced_size);
        return BFS_ERR_TOO_MANY_INDICES;
    }
    out.clear();
    std::vector<u64> reduced;
    std::vector<unsigned char> msg(seed, seed + 32);
    while (out.size() < number) {
        unsigned char digest[64];
        blake2b_host(msg.data(), msg.size(), digest);  // blake2b(seed + bytes(counter)): `counter` zero bytes
        msg.push_back(0);
        u128 acc = 0;
        for (int i = 0; i < 64; ++i) acc = ((acc << 8) | digest[i]) % size;
        u64 index = (u64)acc, red = index % reduced_size;
        bool seen = false;
        for (u64 x : reduced) seen |= (x == red);
        if (!seen) { out.push_back(index); reduced.push_back(red); }
    }
    return BFS_OK;
}

int fri_query(FriSession& S, rp::Transcript& ps, u32 t, u64* h_top, hipStream_t stream) {
    const u32 R = (u32)S.rounds.size();
    if (R < 2) { set_error("Fri.prove needs at least two rounds (fri.py:186 indexes codewords[1])"); return BFS_ERR_BAD_ARG; }
    const double t_begin = now_ms();
    unsigned char seed[32];
    ps.fiat_shamir(ps.objects.size(), seed, 32);
    std::vector<u64> top;
    BFS_TRY(sample_indices(seed, S.rounds[1].length, S.rounds[R - 1].length, t, top));  // fri.py:186-187
    for (u32 s = 0; s < t; ++s) h_top[s] = top[s];

    // plan every opening first (fri.py:191-197), then fetch everything with one gather
    std::vector<std::vector<u64>> layer_idx;  // c indices per layer
    std::vector<u64> idx = top;
    for (u32 i = 0; i + 2 < R; ++i) {  // len(trees) - 1 = R - 2 layers use query()
        for (auto& x : idx) x %= S.rounds[i].length / 2;
        layer_idx.push_back(idx);
    }
    for (auto& x : idx) x %= S.rounds[R - 1].length;
    layer_idx.push_back(idx);  // query_last

    g_fri_timing[2] = now_ms() - t_begin;   // Fiat-Shamir + index sampling
    typedef FriSession::Key Key;
    {   // what the openings can touch at most: 3 elements and 3 authentication paths per colinearity check and layer
        size_t depth_sum = 0;
        for (u32 r = 0; r < R; ++r) depth_sum += 64 - (size_t)__builtin_clzll(S.rounds[r].length);
        S.elements.reserve(S.elements.vals.size() + (size_t)3 * t * R + 8);
        S.nodes.reserve((size_t)3 * t * depth_sum / 2 + 64);
    }
    std::vector<GatherReq> reqs;                 // what to fetch
    std::vector<std::pair<int, Key>> order;      // what the fetched words are: (0 = element | 1 = tree node, key)
    reqs.reserve(4096); order.reserve(4096);
    u64 nwords = 0;
    auto need_element = [&](u32 r, u64 j) {
        Key key(r, j);
        if (S.elements.count(key)) return;       // same Python object in the reference -> same node here
        S.elements[key] = rp::Ref();
        order.push_back({0, key});
        const FriRound& fr = S.rounds[r];
        reqs.push_back(GatherReq{fr.cw + j, 3u, (u32)fr.stride, nwords});
        nwords += 3;
    };
    auto need_path = [&](u32 r, u64 leaf) {      // merkle.py:46-52
        const FriRound& fr = S.rounds[r];
        for (u64 k = fr.length | leaf; k > 1; k >>= 1) {
            Key key(r, k ^ 1);
            if (S.nodes.count(key)) continue;
            S.nodes[key] = rp::Ref();
            order.push_back({1, key});
            reqs.push_back(GatherReq{fr.nodes + (k ^ 1) * 8, 8u, 1u, nwords});
            nwords += 8;
        }
    };
    for (u32 i = 0; i < (u32)layer_idx.size(); ++i) {
        const bool lastq = (i + 1 == layer_idx.size());
        const u32 cur = lastq ? R - 2 : i;
        const u64 half = S.rounds[cur].length / 2;
        for (u32 s = 0; s < t; ++s) {
            u64 c = layer_idx[i][s];
            need_element(cur, c); need_element(cur, c + half); need_element(cur + 1, c);
            need_path(cur, c); need_path(cur, c + half);
            if (!lastq) need_path(cur + 1, c);
        }
    }
    g_fri_timing[3] = now_ms() - t_begin - g_fri_timing[2];   // planning the openings
    const double t_gather = now_ms();
    const u64* words = nullptr;
    PinnedLease req_area, res_area;
    if (!reqs.empty()) {
        BFS_TRY(req_area.get(reqs.size() * sizeof(GatherReq)));
        BFS_TRY(res_area.get(nwords * sizeof(u64)));
        memcpy(req_area.host, reqs.data(), reqs.size() * sizeof(GatherReq));
        u32 grid = (u32)((reqs.size() + 255) / 256);
        hipLaunchKernelGGL(gather_requests_kernel, dim3(grid), dim3(256), 0, stream, (const GatherReq*)req_area.dev, (u32)reqs.size(), (u64*)res_area.dev);
        BFS_HIP(hipGetLastError());
        BFS_HIP(hipStreamSynchronize(stream));
        words = (const u64*)res_area.host;
    }
    g_fri_timing[4] = now_ms() - t_gather;   // gather kernel + synchronisation
    const double t_build = now_ms();
    size_t pos = 0;
    for (auto& o : order) {
        if (o.first == 0) {
            u64 l[3] = {words[pos], words[pos + 1], words[pos + 2]};
            pos += 3;
            S.elements[o.second] = ps.world.xfe_compact(l);
        } else {
            S.nodes[o.second] = rp::mk_bytes(words + pos, 64);
            pos += 8;
        }
    }
    auto path_obj = [&](u32 r, u64 leaf) {
        std::vector<rp::Ref> items;
        items.reserve(64 - (size_t)__builtin_clzll(S.rounds[r].length));
        for (u64 k = S.rounds[r].length | leaf; k > 1; k >>= 1) items.push_back(S.nodes[Key(r, k ^ 1)]);
        return rp::mk_list(std::move(items));
    };
    // push in the reference's order: per layer, t leaf triples then the authentication paths (fri.py:147-156, 166-174)
    for (u32 i = 0; i < (u32)layer_idx.size(); ++i) {
        const bool lastq = (i + 1 == layer_idx.size());
        const u32 cur = lastq ? R - 2 : i;
        const u64 half = S.rounds[cur].length / 2;
        for (u32 s = 0; s < t; ++s) {
            u64 c = layer_idx[i][s];
            ps.objects.push_back(rp::mk_tuple({S.elements[Key(cur, c)], S.elements[Key(cur, c + half)], S.elements[Key(cur + 1, c)]}));
        }
        for (u32 s = 0; s < t; ++s) {
            u64 c = layer_idx[i][s];
            ps.objects.push_back(path_obj(cur, c));
            ps.objects.push_back(path_obj(cur, c + half));
            if (!lastq) ps.objects.push_back(path_obj(cur + 1, c));
        }
    }
    g_fri_timing[5] = now_ms() - t_build;   // building the transcript objects
    return BFS_OK;
}

}  // namespace bfs

using namespace bfs;

extern "C" {

void* bfs_fri_session_new(void) { return new FriSession(); }
void bfs_fri_session_free(void* s) { delete (FriSession*)s; }

int bfs_fri_commit(void* session, void* ps, const uint64_t* d_codeword, uint64_t limb_stride, uint32_t log_n, uint64_t offset,
                   uint64_t omega, uint32_t expansion_factor, void* stream) {
    return fri_commit(*(FriSession*)session, *(rp::Transcript*)ps, d_codeword, limb_stride, log_n, offset, omega, expansion_factor, (hipStream_t)stream);
}

int bfs_fri_query(void* session, void* ps, uint32_t num_colinearity_tests, uint64_t* h_top_level_indices, void* stream) {
    return fri_query(*(FriSession*)session, *(rp::Transcript*)ps, num_colinearity_tests, h_top_level_indices, (hipStream_t)stream);
}

int bfs_fri_prove(void* ps, const uint64_t* d_codeword, uint64_t limb_stride, uint32_t log_n, uint64_t offset, uint64_t omega,
                  uint32_t expansion_factor, uint32_t num_colinearity_tests, uint64_t* h_top_level_indices, void* stream) {
    FriSession S;
    S.use_workspace = true;
    BFS_TRY(fri_commit(S, *(rp::Transcript*)ps, d_codeword, limb_stride, log_n, offset, omega, expansion_factor, (hipStream_t)stream));
    return fri_query(S, *(rp::Transcript*)ps, num_colinearity_tests, h_top_level_indices, (hipStream_t)stream);
}

// The openings of BrainfuckStark.prove (brainfuck_stark.py:315-333) written into the transcript without a Python object in between:
// for every sampled index and every distance d in (0, unit distances...): the base row at index + d, its (salt, path), the extension
// row, its (salt, path); then for every index the combination leaf and its path.  Everything the GPU holds -- row words, salts made
// on the device, tree nodes -- comes back through ONE gather; the objects are built here with the identities the reference's objects
// have (pickle memoises by identity): a row opened twice is one tuple object, a tree node or a salt one bytes object however often it
// appears, every (salt, path) tuple and path list is new, extension elements of a column whose interpolant has all its non-zero
// coefficients at multiples of 2^v share their coefficient objects with the rows i' = i mod modulus (table.ext_sharing_moduli).
int bfs_stark_push_openings(void* ps_, const bfs_gather_request* base_row, uint32_t n_base_req, int32_t base_field_id,
                            const bfs_gather_request* ext_row, uint32_t n_ext_req, const uint64_t* ext_moduli, uint32_t n_ext_cols,
                            uint64_t n, const uint8_t* d_base_nodes, const uint8_t* base_salts, int base_salts_on_device,
                            const uint8_t* d_ext_nodes, const uint8_t* ext_salts, int ext_salts_on_device,
                            const uint64_t* d_combination, uint64_t combination_stride, const uint8_t* d_combination_nodes,
                            const uint64_t* indices, uint32_t n_indices, const uint64_t* distances, uint32_t n_distances,
                            uint64_t* out_leaf_handles, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    rp::Transcript& ps = *(rp::Transcript*)ps_;
    if (n == 0 || (n & (n - 1))) { set_error("bfs_stark_push_openings: n must be a power of two"); return BFS_ERR_BAD_ARG; }
    u32 base_words = 0, ext_words = 0;
    for (u32 k = 0; k < n_base_req; ++k) base_words += base_row[k].nwords;
    for (u32 k = 0; k < n_ext_req; ++k) ext_words += ext_row[k].nwords;
    if (base_words < 3 || ext_words != 3 * n_ext_cols) { set_error("bfs_stark_push_openings: row layout"); return BFS_ERR_BAD_ARG; }
    // unique rows in order of first use, unique indices
    std::vector<u64> rows, uniq_idx;
    for (u32 a = 0; a < n_indices; ++a) {
        for (u32 b = 0; b < n_distances; ++b) {
            const u64 r = (indices[a] + distances[b]) % n;
            if (std::find(rows.begin(), rows.end(), r) == rows.end()) rows.push_back(r);
        }
        if (std::find(uniq_idx.begin(), uniq_idx.end(), indices[a]) == uniq_idx.end()) uniq_idx.push_back(indices[a]);
    }
    // ---- one gather
    std::vector<GatherReq> reqs;
    u64 nwords = 0;
    auto want = [&](const u64* base, u32 words, u32 stride) { reqs.push_back(GatherReq{base, words, stride, nwords}); const u64 at = nwords; nwords += words; return at; };
    struct RowAt { u64 base, ext, base_salt, ext_salt; };
    std::vector<RowAt> row_at(rows.size());
    typedef FriSession::Key Key;
    FriSession::KeyMap node_at[3];                      // (tree, heap index) -> offset of the digest in the gathered words (stored as a K_INT node)
    std::vector<std::pair<int, u64>> node_list;        // order of first use
    std::vector<u64> node_off;
    const uint8_t* tree_nodes[3] = {d_base_nodes, d_ext_nodes, d_combination_nodes};
    auto want_path = [&](int tree, u64 leaf) {
        for (u64 k = n | leaf; k > 1; k >>= 1) {
            const Key key(0, k ^ 1);
            if (node_at[tree].count(key)) continue;
            node_at[tree][key] = rp::mk_int(node_off.size());
            node_list.push_back({tree, k ^ 1});
            node_off.push_back(want((const u64*)(tree_nodes[tree] + 64 * (k ^ 1)), 8, 1));
        }
    };
    for (size_t r = 0; r < rows.size(); ++r) {
        const u64 i = rows[r];
        row_at[r].base = nwords;
        for (u32 k = 0; k < n_base_req; ++k) want(base_row[k].d_base + i, base_row[k].nwords, base_row[k].stride);
        row_at[r].ext = nwords;
        for (u32 k = 0; k < n_ext_req; ++k) want(ext_row[k].d_base + i, ext_row[k].nwords, ext_row[k].stride);
        row_at[r].base_salt = base_salts_on_device ? want((const u64*)(base_salts + 24 * i), 3, 1) : 0;
        row_at[r].ext_salt = ext_salts_on_device ? want((const u64*)(ext_salts + 24 * i), 3, 1) : 0;
        want_path(0, i);
        want_path(1, i);
    }
    std::vector<u64> leaf_at(uniq_idx.size());
    for (size_t a = 0; a < uniq_idx.size(); ++a) {
        leaf_at[a] = want(d_combination + uniq_idx[a], 3, (u32)combination_stride);
        want_path(2, uniq_idx[a]);
    }
    PinnedLease req_area, res_area;
    BFS_TRY(req_area.get(reqs.size() * sizeof(GatherReq)));
    BFS_TRY(res_area.get(nwords * sizeof(u64)));
    memcpy(req_area.host, reqs.data(), reqs.size() * sizeof(GatherReq));
    hipLaunchKernelGGL(gather_requests_kernel, dim3((u32)((reqs.size() + 255) / 256)), dim3(256), 0, stream, (const GatherReq*)req_area.dev, (u32)reqs.size(),
                       (u64*)res_area.dev);
    BFS_HIP(hipGetLastError());
    BFS_HIP(hipStreamSynchronize(stream));
    const u64* words = (const u64*)res_area.host;
    // ---- objects
    std::vector<rp::Ref> node_obj(node_list.size());
    for (size_t k = 0; k < node_list.size(); ++k) node_obj[k] = rp::mk_bytes(words + node_off[k], 64);
    auto path_obj = [&](int tree, u64 leaf) {
        std::vector<rp::Ref> items;
        for (u64 k = n | leaf; k > 1; k >>= 1) items.push_back(node_obj[node_at[tree][Key(0, k ^ 1)]->ival]);
        return rp::mk_list(items);
    };
    const rp::Ref base_field = ps.world.base_field(base_field_id);
    std::vector<rp::Ref> base_rows(rows.size()), ext_rows(rows.size()), base_salt_obj(rows.size()), ext_salt_obj(rows.size());
    std::vector<FriSession::KeyMap> shared(n_ext_cols);          // per column: i mod modulus -> list node holding the coefficient objects
    for (size_t r = 0; r < rows.size(); ++r) {
        const u64 i = rows[r];
        const u64* bw = words + row_at[r].base;
        std::vector<rp::Ref> items;
        u64 l[3] = {bw[0], bw[1], bw[2]};
        items.push_back(ps.world.xfe_compact(l));                                    // the randomizer codeword's element
        for (u32 k = 3; k < base_words; ++k) items.push_back(ps.world.bfe_in(bw[k], base_field));
        base_rows[r] = rp::mk_tuple(items);
        const u64* ew = words + row_at[r].ext;
        items.clear();
        for (u32 c = 0; c < n_ext_cols; ++c) {
            u64 e[3] = {ew[3 * c], ew[3 * c + 1], ew[3 * c + 2]};
            if (ext_moduli[c] == 0) { items.push_back(ps.world.xfe_compact(e)); continue; }
            const Key cls(0, i % ext_moduli[c]);
            if (!shared[c].count(cls)) {
                const int k = e[2] ? 3 : (e[1] ? 2 : (e[0] ? 1 : 0));
                std::vector<rp::Ref> coeffs;
                for (int j = 0; j < k; ++j) coeffs.push_back(ps.world.bfe(e[j], true));
                shared[c][cls] = rp::mk_list(coeffs);
            }
            items.push_back(ps.world.xfe_from(shared[c][cls]->items));
        }
        ext_rows[r] = rp::mk_tuple(items);
        base_salt_obj[r] = base_salts_on_device ? rp::mk_bytes(words + row_at[r].base_salt, 24) : rp::mk_bytes(base_salts + 24 * i, 24);
        ext_salt_obj[r] = ext_salts_on_device ? rp::mk_bytes(words + row_at[r].ext_salt, 24) : rp::mk_bytes(ext_salts + 24 * i, 24);
    }
    // ---- pushes, in the reference's order
    for (u32 a = 0; a < n_indices; ++a)
        for (u32 b = 0; b < n_distances; ++b) {
            const u64 i = (indices[a] + distances[b]) % n;
            const size_t r = (size_t)(std::find(rows.begin(), rows.end(), i) - rows.begin());
            ps.objects.push_back(base_rows[r]);
            ps.objects.push_back(rp::mk_tuple({base_salt_obj[r], path_obj(0, i)}));
            ps.objects.push_back(ext_rows[r]);
            ps.objects.push_back(rp::mk_tuple({ext_salt_obj[r], path_obj(1, i)}));
        }
    std::vector<rp::Ref> leaf_obj(uniq_idx.size());
    for (size_t a = 0; a < uniq_idx.size(); ++a) {
        u64 l[3] = {words[leaf_at[a]], words[leaf_at[a] + 1], words[leaf_at[a] + 2]};
        leaf_obj[a] = ps.world.xfe_compact(l);
    }
    for (u32 a = 0; a < n_indices; ++a) {
        const size_t u = (size_t)(std::find(uniq_idx.begin(), uniq_idx.end(), indices[a]) - uniq_idx.begin());
        ps.objects.push_back(leaf_obj[u]);
        ps.objects.push_back(path_obj(2, indices[a]));
        out_leaf_handles[a] = ps.add(leaf_obj[u]);
    }
    return BFS_OK;
}

int bfs_gather(const bfs_gather_request* requests, uint32_t count, uint64_t* h_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (count == 0) return BFS_OK;
    static_assert(sizeof(bfs_gather_request) == sizeof(GatherReq), "layout of bfs_gather_request");
    u64 nwords = 0;
    std::vector<GatherReq> reqs(count);
    for (uint32_t i = 0; i < count; ++i) {
        reqs[i] = GatherReq{requests[i].d_base, requests[i].nwords, requests[i].stride, nwords};
        nwords += requests[i].nwords;
    }
    PinnedLease req_area, res_area;
    BFS_TRY(req_area.get(reqs.size() * sizeof(GatherReq)));
    BFS_TRY(res_area.get(nwords * sizeof(u64)));
    memcpy(req_area.host, reqs.data(), reqs.size() * sizeof(GatherReq));
    hipLaunchKernelGGL(gather_requests_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, (const GatherReq*)req_area.dev, count, (u64*)res_area.dev);
    BFS_HIP(hipGetLastError());
    BFS_HIP(hipStreamSynchronize(stream));
    memcpy(h_out, res_area.host, nwords * sizeof(u64));
    return BFS_OK;
}

int bfs_fri_session_round0_tree(void* session, const uint8_t* d_nodes, const uint8_t h_root[64]) {
    FriSession* S = (FriSession*)session;
    S->round0_nodes = (const u64*)d_nodes;
    memcpy(S->round0_root, h_root, 64);
    return BFS_OK;
}

int bfs_fri_session_alias(void* session, void* ps, uint32_t round, uint64_t index, uint64_t element_handle) {
    rp::Ref r = ((rp::Transcript*)ps)->get(element_handle);
    if (!r) { set_error("bfs_fri_session_alias: unknown object handle"); return BFS_ERR_BAD_ARG; }
    ((FriSession*)session)->elements[FriSession::Key(round, index)] = r;
    return BFS_OK;
}

void bfs_fri_last_timing(double out[6]) { for (int i = 0; i < 6; ++i) out[i] = g_fri_timing[i]; }

uint32_t bfs_fri_session_rounds(void* session) { return (uint32_t)((FriSession*)session)->rounds.size(); }

int bfs_fri_session_round(void* session, uint32_t r, const uint64_t** d_codeword, uint64_t* length, uint64_t* limb_stride,
                          const uint8_t** d_nodes, uint8_t h_root[64]) {
    FriSession* S = (FriSession*)session;
    if (r >= S->rounds.size()) { set_error("round %u out of range", r); return BFS_ERR_BAD_ARG; }
    const FriRound& fr = S->rounds[r];
    *d_codeword = fr.cw; *length = fr.length; *limb_stride = fr.stride; *d_nodes = (const uint8_t*)fr.nodes;
    memcpy(h_root, fr.root, 64);
    return BFS_OK;
}

int bfs_xfe_fold(const uint64_t* d_in, uint64_t in_stride, uint64_t* d_out, uint64_t out_stride, uint32_t log_n, const uint64_t alpha[3],
                 uint64_t offset, uint64_t omega, void* stream) {
    const u64 N = 1ull << log_n;
    if (log_n == 0) { set_error("cannot fold a codeword of length 1"); return BFS_ERR_BAD_ARG; }
    if (gl_pow(omega, N) != 1 || gl_pow(omega, N / 2) == 1) { set_error("error in commit: omega does not have the right order!"); return BFS_ERR_NOT_ROOT; }
    const u64 *lo = nullptr, *hi = nullptr;
    u32 lo_bits = 0;
    BFS_TRY(ntt_power_tables(gl_inv(omega), log_n, &lo, &hi, &lo_bits));
    Xfe a{{alpha[0] % GL_P, alpha[1] % GL_P, alpha[2] % GL_P}};
    const u64 half = N / 2;
    u32 grid = (u32)((half + 255) / 256);
    hipLaunchKernelGGL(fri_fold_kernel, dim3(grid > 4096 ? 4096 : grid), dim3(256), 0, (hipStream_t)stream, d_in, in_stride, d_out, out_stride, half, a,
                       gl_mul(gl_inv(2), gl_inv(offset % GL_P)), lo, hi, lo_bits, 0u);
    BFS_HIP(hipGetLastError());
    return BFS_OK;
}

}  // extern "C"
