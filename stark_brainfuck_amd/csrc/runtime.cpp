// runtime.cpp -- error state, scratch workspaces, table cache (see runtime.hpp)
#include "runtime.hpp"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace bfs {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}

const char* last_error() { return g_error; }

namespace {
struct Scratch {
    void* ptr = nullptr;
    size_t bytes = 0;
};
std::mutex g_mu;
std::map<std::tuple<int, hipStream_t, int>, Scratch> g_scratch;
std::map<std::tuple<int, uint64_t, uint64_t, uint64_t>, const u64*> g_tables;
// Tables are keyed by caller-supplied values (roots, coset shifts, post-scales), so a caller that sweeps arbitrary shifts would grow
// the cache without bound.  At TABLE_CAP entries the cache starts over: the tables of the generation before LAST are freed (after a
// device synchronisation), the current ones are retired -- a pointer handed out by a lookup therefore stays valid for at least
// TABLE_CAP further insertions, far longer than the lookup-to-launch window of any caller.
constexpr size_t TABLE_CAP = 4096;
std::vector<std::pair<int, const u64*>> g_retired_tables;

void retire_tables_locked() {
    (void)hipDeviceSynchronize();
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto& t : g_retired_tables) {
        if (hipSetDevice(t.first) == hipSuccess) { (void)hipDeviceSynchronize(); (void)hipFree((void*)t.second); }
    }
    (void)hipSetDevice(cur);
    g_retired_tables.clear();
    for (auto& kv : g_tables) g_retired_tables.emplace_back(std::get<0>(kv.first), kv.second);
    g_tables.clear();
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// memory pools (runtime.hpp)
namespace {
size_t size_class(size_t bytes) {
    if (bytes <= 256) return 256;
    size_t p = 256;
    while (p < bytes) p <<= 1;                   // smallest power of two >= bytes
    if (p <= (1u << 20)) return p;
    const size_t step = p >> 4;                  // eight classes per octave: (p/2, p] in steps of p/16
    return (bytes + step - 1) / step * step;
}

struct Block {
    void* ptr;
    hipStream_t released_on;
};

struct Pool {
    bool pinned = false;
    std::mutex mu;
    std::map<std::pair<int, size_t>, std::vector<Block>> free_lists;     // (device, class) -> blocks
    std::unordered_map<void*, std::pair<int, size_t>> live;              // ptr -> (device, class)
    size_t live_bytes = 0, cached_bytes = 0;

    hipError_t raw_alloc(void** p, size_t bytes) {
        return pinned ? hipHostMalloc(p, bytes, hipHostMallocMapped | hipHostMallocCoherent) : hipMalloc(p, bytes);
    }
    void raw_free(void* p) { (void)(pinned ? hipHostFree(p) : hipFree(p)); }

    void trim_locked() {
        for (auto& kv : free_lists)
            for (Block& b : kv.second) raw_free(b.ptr);
        free_lists.clear();
        cached_bytes = 0;
    }

    int alloc(size_t bytes, hipStream_t stream, void** out) {
        int dev = 0;
        if (!pinned) BFS_HIP(hipGetDevice(&dev));
        const size_t cls = size_class(bytes);
        std::lock_guard<std::mutex> lock(mu);
        auto it = free_lists.find(std::make_pair(dev, cls));
        if (it != free_lists.end() && !it->second.empty()) {
            // prefer a block whose release needs no waiting
            std::vector<Block>& v = it->second;
            size_t pick = v.size() - 1;
            for (size_t i = v.size(); i-- > 0;)
                if (v[i].released_on == NO_STREAM || v[i].released_on == stream) { pick = i; break; }
            Block b = v[pick];
            v.erase(v.begin() + (long)pick);
            if (b.released_on != NO_STREAM && b.released_on != stream)
                if (hipStreamSynchronize(b.released_on) != hipSuccess) BFS_HIP(hipDeviceSynchronize());
            cached_bytes -= cls;
            live_bytes += cls;
            live[b.ptr] = std::make_pair(dev, cls);
            *out = b.ptr;
            return BFS_OK;
        }
        void* p = nullptr;
        hipError_t e = raw_alloc(&p, cls);
        if (e != hipSuccess) {                   // out of memory: give the cache back to the driver and try once more
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();
            trim_locked();
            e = raw_alloc(&p, cls);
        }
        if (e != hipSuccess) {
            set_error("%s of %zu bytes failed: %s", pinned ? "hipHostMalloc" : "hipMalloc", cls, hipGetErrorString(e));
            return BFS_ERR_HIP;
        }
        live_bytes += cls;
        live[p] = std::make_pair(dev, cls);
        *out = p;
        return BFS_OK;
    }

    int release(void* p, hipStream_t stream) {
        if (!p) return BFS_OK;
        std::lock_guard<std::mutex> lock(mu);
        auto it = live.find(p);
        if (it == live.end()) {
            set_error("release of a pointer this library did not allocate");
            return BFS_ERR_BAD_ARG;
        }
        const std::pair<int, size_t> key = it->second;
        live.erase(it);
        live_bytes -= key.second;
        static const bool keep = [] { const char* v = getenv("BFS_POOL"); return !(v && v[0] == '0'); }();
        if (!keep) {                             // BFS_POOL=0: straight to the driver (debugging aid)
            raw_free(p);
            return BFS_OK;
        }
        free_lists[key].push_back(Block{p, stream});
        cached_bytes += key.second;
        return BFS_OK;
    }
};

Pool g_device_pool, g_host_pool;
struct PoolInit { PoolInit() { g_host_pool.pinned = true; } } g_pool_init;
}  // namespace

int device_alloc(size_t bytes, hipStream_t stream, void** out) { return g_device_pool.alloc(bytes, stream, out); }
int device_release(void* ptr, hipStream_t stream) {
    // a transform route remembered for a pair of buffers (bfs_ntt_tune) dies with either of them: the next owner of this block is
    // another buffer at the same address
    size_t cls = 0;
    {
        std::lock_guard<std::mutex> lock(g_device_pool.mu);
        auto it = g_device_pool.live.find(ptr);
        if (it != g_device_pool.live.end()) cls = it->second.second;
    }
    if (cls) (void)ntt_route_forget_range(ptr, cls, false);
    return g_device_pool.release(ptr, stream);
}
int device_pool_trim() {
    BFS_HIP(hipDeviceSynchronize());
    ntt_route_trim();
    std::lock_guard<std::mutex> lock(g_device_pool.mu);
    g_device_pool.trim_locked();
    return BFS_OK;
}
void device_pool_stats(size_t* live_bytes, size_t* cached_bytes) {
    std::lock_guard<std::mutex> lock(g_device_pool.mu);
    if (live_bytes) *live_bytes = g_device_pool.live_bytes;
    if (cached_bytes) *cached_bytes = g_device_pool.cached_bytes;
}
int host_alloc(size_t bytes, void** out) { return g_host_pool.alloc(bytes, NO_STREAM, out); }
int host_release(void* ptr) { return g_host_pool.release(ptr, NO_STREAM); }

namespace {
bool is_pageable(const void* h) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, h) != hipSuccess) {
        (void)hipGetLastError();
        return true;
    }
    return attr.type == hipMemoryTypeUnregistered;
}

int copy_staged(void* dst, const void* src, size_t bytes, bool to_device, hipStream_t stream) {
    constexpr size_t CHUNK = 8u << 20;
    void* bounce[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    int rc = BFS_OK;
    auto fail = [&](hipError_t e, const char* what) {
        set_error("%s failed: %s", what, hipGetErrorString(e));
        rc = BFS_ERR_HIP;
    };
    for (int i = 0; i < 2 && rc == BFS_OK; ++i) {
        rc = host_alloc(CHUNK, &bounce[i]);
        if (rc != BFS_OK) break;
        hipError_t e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
        if (e != hipSuccess) fail(e, "hipEventCreate");
    }
    const size_t chunks = (bytes + CHUNK - 1) / CHUNK;
    auto span = [&](size_t k) { return bytes - k * CHUNK < CHUNK ? bytes - k * CHUNK : CHUNK; };
    if (to_device) {
        for (size_t k = 0; rc == BFS_OK && k < chunks; ++k) {
            const int b = (int)(k & 1);
            hipError_t e = k >= 2 ? hipEventSynchronize(done[b]) : hipSuccess;       // the DMA out of this buffer two chunks ago
            if (e != hipSuccess) { fail(e, "hipEventSynchronize"); break; }
            memcpy(bounce[b], (const char*)src + k * CHUNK, span(k));
            e = hipMemcpyAsync((char*)dst + k * CHUNK, bounce[b], span(k), hipMemcpyHostToDevice, stream);
            if (e == hipSuccess) e = hipEventRecord(done[b], stream);
            if (e != hipSuccess) fail(e, "hipMemcpyAsync");
        }
    } else {
        // chunk k lands in bounce[k & 1] while the host drains chunk k - 1
        for (size_t k = 0; rc == BFS_OK && k <= chunks; ++k) {
            if (k < chunks) {
                const int b = (int)(k & 1);
                hipError_t e = hipMemcpyAsync(bounce[b], (const char*)src + k * CHUNK, span(k), hipMemcpyDeviceToHost, stream);
                if (e == hipSuccess) e = hipEventRecord(done[b], stream);
                if (e != hipSuccess) { fail(e, "hipMemcpyAsync"); break; }
            }
            if (k >= 1) {
                const int b = (int)((k - 1) & 1);
                hipError_t e = hipEventSynchronize(done[b]);
                if (e != hipSuccess) { fail(e, "hipEventSynchronize"); break; }
                memcpy((char*)dst + (k - 1) * CHUNK, bounce[b], span(k - 1));
            }
        }
    }
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess && rc == BFS_OK) fail(e, "hipStreamSynchronize");
    for (int i = 0; i < 2; ++i) {
        if (done[i]) (void)hipEventDestroy(done[i]);
        if (bounce[i]) (void)host_release(bounce[i]);
    }
    return rc;
}
}  // namespace

int copy_h2d(void* d, const void* h, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return BFS_OK;
    if (bytes >= (256u << 10) && is_pageable(h)) return copy_staged(d, h, bytes, true, stream);
    BFS_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, stream));
    BFS_HIP(hipStreamSynchronize(stream));
    return BFS_OK;
}

int copy_d2h(void* h, const void* d, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return BFS_OK;
    if (bytes >= (256u << 10) && is_pageable(h)) return copy_staged(h, d, bytes, false, stream);
    BFS_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, stream));
    BFS_HIP(hipStreamSynchronize(stream));
    return BFS_OK;
}

int workspace(int slot, size_t bytes, hipStream_t stream, void** out) {
    int dev = 0;
    BFS_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mu);
    Scratch& s = g_scratch[std::make_tuple(dev, stream, slot)];
    if (s.bytes < bytes) {
        if (s.ptr) {
            BFS_HIP(hipStreamSynchronize(stream));  // earlier kernels may still use the old buffer
            BFS_HIP(hipFree(s.ptr));
            s.ptr = nullptr;
            s.bytes = 0;
        }
        BFS_HIP(hipMalloc(&s.ptr, bytes));
        s.bytes = bytes;
    }
    *out = s.ptr;
    return BFS_OK;
}

bool workspace_peek(int slot, hipStream_t stream, void** ptr, size_t* bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_scratch.find(std::make_tuple(dev, stream, slot));
    if (it == g_scratch.end() || !it->second.ptr) return false;
    if (ptr) *ptr = it->second.ptr;
    if (bytes) *bytes = it->second.bytes;
    return true;
}

// give one scratch buffer back to the driver (the NTT's route measurement keeps only the candidate it chose); the stream must be idle
int workspace_release(int slot, hipStream_t stream) {
    int dev = 0;
    BFS_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_scratch.find(std::make_tuple(dev, stream, slot));
    if (it != g_scratch.end()) {
        if (it->second.ptr) (void)hipFree(it->second.ptr);
        g_scratch.erase(it);
    }
    return BFS_OK;
}

// bfs_stream_destroy: the stream's scratch buffers go back to the driver and pooled blocks that were released on it no longer
// name it (a later stream may get the same handle value)
int stream_retire(hipStream_t stream) {
    BFS_HIP(hipStreamSynchronize(stream));
    int dev = 0;
    BFS_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lock(g_mu);
        for (auto it = g_scratch.begin(); it != g_scratch.end();) {
            if (std::get<0>(it->first) == dev && std::get<1>(it->first) == stream) {
                if (it->second.ptr) (void)hipFree(it->second.ptr);
                it = g_scratch.erase(it);
            } else {
                ++it;
            }
        }
    }
    for (Pool* pool : {&g_device_pool, &g_host_pool}) {
        std::lock_guard<std::mutex> lock(pool->mu);
        for (auto& kv : pool->free_lists)
            for (Block& b : kv.second)
                if (b.released_on == stream) b.released_on = NO_STREAM;     // synchronised above: nothing in flight on it
    }
    return BFS_OK;
}

bool cached_table_lookup(uint64_t a, uint64_t b, uint64_t c, const u64** d_out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_tables.find(std::make_tuple(dev, a, b, c));
    if (it == g_tables.end()) return false;
    *d_out = it->second;
    return true;
}

int cached_table(uint64_t a, uint64_t b, uint64_t c, const u64* host, size_t count, const u64** d_out) {
    int dev = 0;
    BFS_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_mu);
    auto key = std::make_tuple(dev, a, b, c);
    auto it = g_tables.find(key);
    if (it != g_tables.end()) {
        *d_out = it->second;
        return BFS_OK;
    }
    if (g_tables.size() >= TABLE_CAP) retire_tables_locked();
    void* d = nullptr;
    BFS_HIP(hipMalloc(&d, count * sizeof(u64)));
    if (hipMemcpy(d, host, count * sizeof(u64), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        set_error("upload of a %zu-entry table failed", count);
        return BFS_ERR_HIP;
    }
    g_tables[key] = (const u64*)d;
    *d_out = (const u64*)d;
    return BFS_OK;
}

}  // namespace bfs
