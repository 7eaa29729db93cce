// runtime.cpp -- error state, scratch workspaces, table cache (see runtime.hpp)
#include "runtime.hpp"

#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <tuple>

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
}  // namespace

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
    void* d = nullptr;
    BFS_HIP(hipMalloc(&d, count * sizeof(u64)));
    BFS_HIP(hipMemcpy(d, host, count * sizeof(u64), hipMemcpyHostToDevice));
    g_tables[key] = (const u64*)d;
    *d_out = (const u64*)d;
    return BFS_OK;
}

}  // namespace bfs
