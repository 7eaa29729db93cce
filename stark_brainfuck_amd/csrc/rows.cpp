// rows.cpp -- commitment to "zipped" codewords: leaf i is the TUPLE of the i-th elements of several codewords plus a salt
// (the reference's SaltedMerkle(list(zip(*codewords))), /root/reference/code/brainfuck_stark.py:178-179,197-198 with
// salted_merkle.py:22-47).  A leaf's preimage is pickle.dumps(tuple) || pickle.dumps(salt); its layout depends on how
// many coefficients every extension element of the row stores (memo indices shift), so rows are pickled by the generic
// emitter (refpickle.hpp) on host threads -- one private object world per thread -- and hashed on the GPU in one batch.
#include <thread>
#include <vector>

#include "../../include/bfstark.h"
#include "refpickle.hpp"
#include "runtime.hpp"

namespace bfs {
int merkle_build_bytes_launch(const u64* d_data, const u64* d_offsets, const u32* d_lengths, u64 n, u64* d_nodes, hipStream_t stream);
}

using namespace bfs;

extern "C" int bfs_merkle_build_rows(const bfs_row_column* columns, uint32_t ncols, uint64_t n, const uint8_t* h_salts, uint8_t* d_nodes,
                                     uint32_t threads, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n == 0) return BFS_OK;
    if (((uintptr_t)d_nodes & 15) != 0) { set_error("d_nodes must be 16-byte aligned"); return BFS_ERR_BAD_ARG; }
    // 1. codewords to the host, column-major
    std::vector<std::vector<u64>> host(ncols);
    for (uint32_t c = 0; c < ncols; ++c) {
        const int planes = columns[c].is_ext ? 3 : 1;
        host[c].resize((size_t)planes * n);
        BFS_HIP(hipMemcpyAsync(host[c].data(), columns[c].d_values, host[c].size() * sizeof(u64), hipMemcpyDeviceToHost, stream));
    }
    BFS_HIP(hipStreamSynchronize(stream));
    // 2. pickle the rows
    if (threads == 0) threads = std::thread::hardware_concurrency();
    if (threads == 0) threads = 1;
    if (threads > 64) threads = 64;
    if ((u64)threads > n) threads = (uint32_t)n;
    std::vector<std::string> chunk_blob(threads);
    std::vector<std::vector<u32>> chunk_len(threads);
    auto work = [&](uint32_t t) {
        const u64 lo = n * t / threads, hi = n * (t + 1) / threads;
        rp::World world;
        rp::Pickler pickler(&world);
        std::string& blob = chunk_blob[t];
        std::vector<u32>& lens = chunk_len[t];
        lens.reserve(hi - lo);
        std::vector<rp::Ref> items(ncols);
        for (u64 i = lo; i < hi; ++i) {
            for (uint32_t c = 0; c < ncols; ++c) {
                if (columns[c].is_ext) {
                    const u64 l[3] = {host[c][i], host[c][n + i], host[c][2 * n + i]};
                    items[c] = world.xfe_compact(l);
                } else {
                    items[c] = world.bfe_in(host[c][i], world.base_field(columns[c].field_id));
                }
            }
            std::string s = pickler.dumps(rp::mk_tuple(items));
            if (h_salts) s += pickler.dumps(rp::mk_bytes(h_salts + 24 * i, 24));
            lens.push_back((u32)s.size());
            s.resize((s.size() + 7) & ~(size_t)7, '\0');     // every message starts on a 64-bit word
            blob += s;
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    // 3. one blob, word offsets, lengths -> HBM -> leaf digests and the tree
    size_t total = 0;
    for (auto& b : chunk_blob) total += b.size();
    std::vector<u64> offsets(n);
    std::vector<u32> lengths(n + (n & 1));
    void *d_blob = nullptr, *d_off = nullptr, *d_len = nullptr;
    BFS_HIP(hipMalloc(&d_blob, total ? total : 8));
    size_t pos = 0, row = 0;
    for (uint32_t t = 0; t < threads; ++t) {
        BFS_HIP(hipMemcpyAsync((char*)d_blob + pos, chunk_blob[t].data(), chunk_blob[t].size(), hipMemcpyHostToDevice, stream));
        size_t p = pos;
        for (u32 len : chunk_len[t]) {
            offsets[row] = p / 8;
            lengths[row] = len;
            p += (len + 7) & ~(size_t)7;
            ++row;
        }
        pos += chunk_blob[t].size();
    }
    BFS_HIP(hipMalloc(&d_off, offsets.size() * sizeof(u64)));
    BFS_HIP(hipMalloc(&d_len, lengths.size() * sizeof(u32)));
    BFS_HIP(hipMemcpyAsync(d_off, offsets.data(), offsets.size() * sizeof(u64), hipMemcpyHostToDevice, stream));
    BFS_HIP(hipMemcpyAsync(d_len, lengths.data(), lengths.size() * sizeof(u32), hipMemcpyHostToDevice, stream));
    int rc = merkle_build_bytes_launch((const u64*)d_blob, (const u64*)d_off, (const u32*)d_len, n, (u64*)d_nodes, stream);
    hipError_t e = hipStreamSynchronize(stream);
    (void)hipFree(d_blob); (void)hipFree(d_off); (void)hipFree(d_len);
    if (rc) return rc;
    BFS_HIP(e);
    return BFS_OK;
}
