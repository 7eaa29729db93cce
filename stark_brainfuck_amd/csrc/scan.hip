// scan.hip -- the column extensions of the tables on the GPU: running products and running evaluations
//   ProcessorTable.extend   /root/reference/code/processor_table.py:329-427
//   InstructionTable.extend instruction_table.py:167-231
//   MemoryTable.extend      memory_table.py:172-206
//   IOTable.extend_iotable  io_table.py:77-110
// The reference walks the rows one after the other.  Each row is an affine map of the running value,
//   kind 0 (running product):     s -> s * (c0 - c1 x1[i] - c2 x2[i] - c3 x3[i])         (masked rows; identity otherwise)
//   kind 1 (running evaluation):  s -> s * c0 + (c1 x1[i] + c2 x2[i] + c3 x3[i])
// and maps compose associatively, so the column is a prefix scan: per-block aggregates (scan_reduce_kernel), one workgroup
// scanning the <= 256 block aggregates (scan_spine_kernel), then every thread replays its rows from its own starting value
// (scan_apply_kernel).  Same semantics as the host primitive bfs_xfe_scan (hostscan.cpp), which the tests compare it with;
// the trace columns are already in HBM for the low-degree extension, and the extension columns never visit the host.
#include <string.h>

#include <vector>

#include "../../include/bfstark.h"
#include "gl.hpp"
#include "runtime.hpp"

namespace bfs {

struct ScanArgs {
    const u64 *x1, *x2, *x3;      // device columns (null = absent)
    const unsigned char* mask;    // device bytes (null = every row)
    u64 n;
    u64 shift1;                   // x1 is read at (i + shift1) mod n: "the next row's memory value" of the input evaluation
    u64 items;                    // consecutive rows per thread
    Xfe c[4];
    Xfe initial;
    int record_before;
    u64* out;                     // three limb planes
    u64 out_stride;
    Xfe* block_m;                 // per block: aggregate map x -> x * m + c ...
    Xfe* block_c;
    Xfe* block_state;             // ... and, after the spine, the running value the block starts from; [gridDim.x] = terminal
    u64* terminal;                // optional device copy of the terminal (three words)
    u32 kind, blocks;             // batched launches (scan_*_many_kernel): which scan this is, and how many workgroups it uses
};

struct Affine {
    Xfe m, c;
};

template <int KIND>
__device__ __forceinline__ Affine row_map(const ScanArgs& a, u64 i) {
    Affine r{Xfe{{1, 0, 0}}, Xfe{{0, 0, 0}}};
    if (a.mask && !a.mask[i]) return r;
    Xfe lin{{0, 0, 0}};
    if (a.x1) {
        u64 j = i + a.shift1;
        if (j >= a.n) j -= a.n;
        lin = xfe_add(lin, xfe_scale(a.c[1], a.x1[j]));
    }
    if (a.x2) lin = xfe_add(lin, xfe_scale(a.c[2], a.x2[i]));
    if (a.x3) lin = xfe_add(lin, xfe_scale(a.c[3], a.x3[i]));
    if (KIND == 0) r.m = xfe_sub(a.c[0], lin);
    else { r.m = a.c[0]; r.c = lin; }
    return r;
}

// first `f`, then `g`
template <int KIND>
__device__ __forceinline__ Affine compose(const Affine& f, const Affine& g) {
    Affine r;
    r.m = xfe_mul(f.m, g.m);
    if (KIND == 0) r.c = Xfe{{0, 0, 0}};
    else r.c = xfe_add(xfe_mul(f.c, g.m), g.c);
    return r;
}

template <int KIND>
__device__ __forceinline__ Xfe apply(const Affine& f, const Xfe& s) {
    Xfe r = xfe_mul(s, f.m);
    return KIND == 0 ? r : xfe_add(r, f.c);
}

template <int KIND>
__device__ __forceinline__ Affine thread_aggregate(const ScanArgs& a, u64 first) {
    Affine agg{Xfe{{1, 0, 0}}, Xfe{{0, 0, 0}}};
    const u64 last = first + a.items < a.n ? first + a.items : a.n;
    for (u64 i = first; i < last; ++i) agg = compose<KIND>(agg, row_map<KIND>(a, i));
    return agg;
}

// inclusive scan of 256 maps in LDS (Hillis-Steele: map t becomes the composition of maps 0..t)
template <int KIND>
__device__ __forceinline__ Affine block_inclusive_scan(Affine mine, Affine* lds) {
    const u32 t = threadIdx.x;
    lds[t] = mine;
    __syncthreads();
    for (u32 d = 1; d < 256; d <<= 1) {
        Affine left;
        const bool has = t >= d;
        if (has) left = lds[t - d];
        __syncthreads();
        if (has) {
            mine = compose<KIND>(left, mine);
            lds[t] = mine;
        }
        __syncthreads();
    }
    return mine;
}

template <int KIND>
__device__ __forceinline__ void scan_reduce_body(const ScanArgs& a, Affine* lds) {
    const u64 first = ((u64)blockIdx.x * 256 + threadIdx.x) * a.items;
    const Affine total = block_inclusive_scan<KIND>(thread_aggregate<KIND>(a, first), lds);
    if (threadIdx.x == 255) { a.block_m[blockIdx.x] = total.m; a.block_c[blockIdx.x] = total.c; }
}

template <int KIND>
__global__ void __launch_bounds__(256) scan_reduce_kernel(const ScanArgs a) {
    __shared__ Affine lds[256];
    scan_reduce_body<KIND>(a, lds);
}

template <int KIND>
__device__ __forceinline__ void scan_spine_body(const ScanArgs& a, u32 blocks, Affine* lds) {
    const u32 t = threadIdx.x;
    Affine mine{Xfe{{1, 0, 0}}, Xfe{{0, 0, 0}}};
    if (t < blocks) { mine.m = a.block_m[t]; mine.c = a.block_c[t]; }
    const Affine incl = block_inclusive_scan<KIND>(mine, lds);
    // block t + 1 starts from the value after blocks 0..t; slot `blocks` is the terminal
    if (t < blocks) {
        const Xfe after = apply<KIND>(incl, a.initial);
        a.block_state[t + 1] = after;
        if (t + 1 == blocks && a.terminal) { a.terminal[0] = after.c[0]; a.terminal[1] = after.c[1]; a.terminal[2] = after.c[2]; }
    }
    if (t == 0) a.block_state[0] = a.initial;
}

template <int KIND>
__global__ void __launch_bounds__(256) scan_spine_kernel(const ScanArgs a, u32 blocks) {
    __shared__ Affine lds[256];
    scan_spine_body<KIND>(a, blocks, lds);
}

template <int KIND>
__device__ __forceinline__ void scan_apply_body(const ScanArgs& a, Affine* lds) {
    const u32 t = threadIdx.x;
    const u64 first = ((u64)blockIdx.x * 256 + t) * a.items;
    const Affine incl = block_inclusive_scan<KIND>(thread_aggregate<KIND>(a, first), lds);
    __syncthreads();
    lds[t] = incl;
    __syncthreads();
    Xfe state = a.block_state[blockIdx.x];
    if (t > 0) state = apply<KIND>(lds[t - 1], state);
    const u64 last = first + a.items < a.n ? first + a.items : a.n;
    for (u64 i = first; i < last; ++i) {
        if (a.record_before) { a.out[i] = state.c[0]; a.out[a.out_stride + i] = state.c[1]; a.out[2 * a.out_stride + i] = state.c[2]; }
        state = apply<KIND>(row_map<KIND>(a, i), state);
        if (!a.record_before) { a.out[i] = state.c[0]; a.out[a.out_stride + i] = state.c[1]; a.out[2 * a.out_stride + i] = state.c[2]; }
    }
}

template <int KIND>
__global__ void __launch_bounds__(256) scan_apply_kernel(const ScanArgs a) {
    __shared__ Affine lds[256];
    scan_apply_body<KIND>(a, lds);
}

// Several scans per launch (blockIdx.y = which one): the nine scans of a proof over short tables are nine chains of three tiny,
// latency-bound kernels (10-25 us each: eight dependent compose steps per workgroup scan); side by side they cost as much as one.
__global__ void __launch_bounds__(256) scan_reduce_many_kernel(const ScanArgs* args) {
    __shared__ Affine lds[256];
    const ScanArgs a = args[blockIdx.y];
    if (blockIdx.x >= a.blocks) return;
    if (a.kind == 0) scan_reduce_body<0>(a, lds); else scan_reduce_body<1>(a, lds);
}
__global__ void __launch_bounds__(256) scan_spine_many_kernel(const ScanArgs* args) {
    __shared__ Affine lds[256];
    const ScanArgs a = args[blockIdx.y];
    if (a.kind == 0) scan_spine_body<0>(a, a.blocks, lds); else scan_spine_body<1>(a, a.blocks, lds);
}
__global__ void __launch_bounds__(256) scan_apply_many_kernel(const ScanArgs* args) {
    __shared__ Affine lds[256];
    const ScanArgs a = args[blockIdx.y];
    if (blockIdx.x >= a.blocks) return;
    if (a.kind == 0) scan_apply_body<0>(a, lds); else scan_apply_body<1>(a, lds);
}

template <int KIND>
static int scan_launch(ScanArgs& a, u32 blocks, hipStream_t stream) {
    hipLaunchKernelGGL(scan_reduce_kernel<KIND>, dim3(blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(scan_spine_kernel<KIND>, dim3(1), dim3(256), 0, stream, a, blocks);
    hipLaunchKernelGGL(scan_apply_kernel<KIND>, dim3(blocks), dim3(256), 0, stream, a);
    BFS_HIP(hipGetLastError());
    return BFS_OK;
}

}  // namespace bfs

using namespace bfs;

static void scan_fill(ScanArgs& a, int kind, const u64* d_x1, const u64* d_x2, const u64* d_x3, u64 shift1, const unsigned char* d_mask, u64 n,
                      const u64 constants[12], const u64 initial[3], int record_before, u64* d_out, u64 out_stride, u64* d_terminal) {
    a = ScanArgs{};
    a.x1 = d_x1; a.x2 = d_x2; a.x3 = d_x3; a.mask = d_mask; a.n = n; a.shift1 = d_x1 ? shift1 % n : 0;
    for (int j = 0; j < 4; ++j) a.c[j] = Xfe{{constants[3 * j] % GL_P, constants[3 * j + 1] % GL_P, constants[3 * j + 2] % GL_P}};
    a.initial = Xfe{{initial[0] % GL_P, initial[1] % GL_P, initial[2] % GL_P}};
    a.record_before = record_before;
    a.out = d_out; a.out_stride = out_stride; a.terminal = d_terminal;
    u64 blocks = (n + 255) / 256;
    if (blocks > 256) blocks = 256;
    a.items = (n + blocks * 256 - 1) / (blocks * 256);
    blocks = (n + a.items * 256 - 1) / (a.items * 256);          // no empty blocks at the end
    a.blocks = (u32)blocks;
    a.kind = (u32)kind;
}

extern "C" int bfs_xfe_scan_device_many(const bfs_scan_spec* specs, uint32_t count, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (count == 0) return BFS_OK;
    if (count > 64) { set_error("bfs_xfe_scan_device_many: at most 64 scans per call"); return BFS_ERR_BAD_ARG; }
    std::vector<ScanArgs> args(count);
    const size_t per = (3 * 256 + 8) * sizeof(Xfe);
    void* w = nullptr;
    BFS_TRY(workspace(7, count * per + count * sizeof(ScanArgs) + 64, stream, &w));
    u32 max_blocks = 0;
    for (uint32_t k = 0; k < count; ++k) {
        const bfs_scan_spec& sp = specs[k];
        if (sp.kind != 0 && sp.kind != 1) { set_error("bfs_xfe_scan_device_many: kind must be 0 (product) or 1 (evaluation)"); return BFS_ERR_BAD_ARG; }
        if (sp.n == 0 || sp.out_stride < sp.n) { set_error("bfs_xfe_scan_device_many: n >= 1 and out_stride >= n"); return BFS_ERR_BAD_ARG; }
        scan_fill(args[k], sp.kind, sp.d_x1, sp.d_x2, sp.d_x3, sp.shift1, sp.d_mask, sp.n, sp.constants, sp.initial, sp.record_before, sp.d_out,
                  sp.out_stride, sp.d_terminal);
        args[k].block_m = (Xfe*)((char*)w + k * per);
        args[k].block_c = args[k].block_m + 256;
        args[k].block_state = args[k].block_c + 256;
        if (args[k].blocks > max_blocks) max_blocks = args[k].blocks;
    }
    ScanArgs* d_args = (ScanArgs*)((char*)w + count * per);
    BFS_HIP(hipMemcpyAsync(d_args, args.data(), count * sizeof(ScanArgs), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(scan_reduce_many_kernel, dim3(max_blocks, count), dim3(256), 0, stream, (const ScanArgs*)d_args);
    hipLaunchKernelGGL(scan_spine_many_kernel, dim3(1, count), dim3(256), 0, stream, (const ScanArgs*)d_args);
    hipLaunchKernelGGL(scan_apply_many_kernel, dim3(max_blocks, count), dim3(256), 0, stream, (const ScanArgs*)d_args);
    BFS_HIP(hipGetLastError());
    BFS_HIP(hipStreamSynchronize(stream));            // `args` is pageable host memory the copy may still be reading
    return BFS_OK;
}

extern "C" int bfs_xfe_scan_device(int kind, const uint64_t* d_x1, const uint64_t* d_x2, const uint64_t* d_x3, uint64_t shift1,
                                   const uint8_t* d_mask, uint64_t n, const uint64_t constants[12], const uint64_t initial[3],
                                   int record_before, uint64_t* d_out, uint64_t out_stride, uint64_t* d_terminal, uint64_t* terminal,
                                   void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (kind != 0 && kind != 1) { set_error("bfs_xfe_scan_device: kind must be 0 (product) or 1 (evaluation)"); return BFS_ERR_BAD_ARG; }
    if (n == 0) {
        u64 t[3] = {initial[0] % GL_P, initial[1] % GL_P, initial[2] % GL_P};
        if (terminal) memcpy(terminal, t, sizeof t);
        if (d_terminal) {
            BFS_HIP(hipMemcpyAsync(d_terminal, t, sizeof t, hipMemcpyHostToDevice, stream));
            BFS_HIP(hipStreamSynchronize(stream));
        }
        return BFS_OK;
    }
    if (out_stride < n) { set_error("bfs_xfe_scan_device: out_stride < n"); return BFS_ERR_BAD_ARG; }
    ScanArgs a{};
    a.x1 = d_x1; a.x2 = d_x2; a.x3 = d_x3; a.mask = d_mask; a.n = n; a.shift1 = d_x1 ? shift1 % n : 0;
    for (int j = 0; j < 4; ++j) a.c[j] = Xfe{{constants[3 * j] % GL_P, constants[3 * j + 1] % GL_P, constants[3 * j + 2] % GL_P}};
    a.initial = Xfe{{initial[0] % GL_P, initial[1] % GL_P, initial[2] % GL_P}};
    a.record_before = record_before;
    a.out = d_out; a.out_stride = out_stride; a.terminal = d_terminal;
    u64 blocks = (n + 255) / 256;
    if (blocks > 256) blocks = 256;
    a.items = (n + blocks * 256 - 1) / (blocks * 256);
    blocks = (n + a.items * 256 - 1) / (a.items * 256);          // no empty blocks at the end
    void* w = nullptr;
    BFS_TRY(workspace(7, (3 * 256 + 8) * sizeof(Xfe), stream, &w));       // reused by the next scan on this stream: stream order keeps it safe
    a.block_m = (Xfe*)w; a.block_c = a.block_m + 256; a.block_state = a.block_c + 256;
    BFS_TRY(kind == 0 ? scan_launch<0>(a, (u32)blocks, stream) : scan_launch<1>(a, (u32)blocks, stream));
    if (terminal) {
        BFS_HIP(hipMemcpyAsync(terminal, a.block_state + blocks, sizeof(Xfe), hipMemcpyDeviceToHost, stream));
        BFS_HIP(hipStreamSynchronize(stream));
    }
    return BFS_OK;
}


// ---- padding of the trace tables on the device (round 5) -----------------------------------------------------------------------------
// Table.pad (/root/reference/code/table.py:25 with processor_table.py:24-35, instruction_table.py:19-25, memory_table.py:40-44,
// io_table.py:17-21) brings a table to a power-of-two height; the prover then wants the padded table COLUMN-major (one codeword per
// column) and, for the scans above, a byte per row saying whether the row takes part.  Doing that on the host cost 0.3-0.6 ms of a
// 14 ms proof with the GPU waiting (scalar loops over 10^6 words): the host now only copies the rows it was given into pinned memory,
// and this kernel transposes, reduces mod p, appends the padding rows and writes the masks.  One thread per row.
namespace bfs {

struct PadTables { bfs_trace_pad_table t[5]; u32 count; };

__global__ void trace_pad_kernel(const PadTables a) {
    const bfs_trace_pad_table& T = a.t[blockIdx.y];
    const u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.y >= a.count || r >= T.height) return;
    const u64 rows = T.rows, h = T.height;
    const u32 w = T.width;
    auto canonical = [](u64 v) { return v >= GL_P ? v - GL_P : v; };
    u64 v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < rows) {
        for (u32 c = 0; c < w; ++c) v[c] = canonical(T.d_rows[r * T.row_stride + c]);
    } else if (rows && T.kind <= 2) {
        u64 last[8];
        for (u32 c = 0; c < w; ++c) last[c] = canonical(T.d_rows[(rows - 1) * T.row_stride + c]);
        const u64 j = (r - rows + 1) % GL_P;                    // the j-th padding row
        if (T.kind == 0) { v[0] = gl_add(last[0], j); v[1] = last[1]; v[4] = last[4]; v[5] = last[5]; v[6] = last[6]; }   // the cycle count keeps counting; ip, mp, mv, mvi stay
        else if (T.kind == 1) v[0] = last[0];                   // the last address repeats
        else { v[0] = gl_add(last[0], j); v[1] = last[1]; v[2] = last[2]; v[3] = 1; }                                      // dummy rows, the cycle counts up
    } else if (T.kind == 2) {
        v[0] = (r + 1) % GL_P; v[3] = 1;                        // (a memory table without rows: "last" is the zero row)
    } else if (T.kind == 0) {
        v[0] = (r + 1) % GL_P;
    }
    for (u32 c = 0; c < w; ++c) T.d_out[(u64)c * h + r] = v[c];
    if (T.kind == 0) {
        T.d_mask0[r] = v[2] != 0; T.d_mask1[r] = v[2] == (u64)','; T.d_mask2[r] = v[2] == (u64)'.';
    } else if (T.kind == 1) {
        bool same = false;
        if (r > 0) {
            const u64 above = r - 1 < rows ? canonical(T.d_rows[(r - 1) * T.row_stride]) : (rows ? canonical(T.d_rows[(rows - 1) * T.row_stride]) : 0);
            same = v[0] == above;
        }
        T.d_mask0[r] = v[1] != 0 && same;                       // product rows
        T.d_mask1[r] = !same;                                   // evaluation rows
    } else if (T.kind == 2) {
        T.d_mask0[r] = v[3] == 0;                               // non-dummy rows
    }
}

}  // namespace bfs

extern "C" int bfs_trace_pad(const bfs_trace_pad_table* tables, uint32_t count, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (count == 0) return BFS_OK;
    if (count > 5) { bfs::set_error("bfs_trace_pad: at most 5 tables"); return BFS_ERR_BAD_ARG; }
    bfs::PadTables a{};
    a.count = count;
    u64 tallest = 0;
    for (uint32_t k = 0; k < count; ++k) {
        const bfs_trace_pad_table& t = tables[k];
        if (t.width == 0 || t.width > 8 || t.rows > t.height || (t.rows && (t.row_stride < t.width || t.d_rows == nullptr)) || (t.height && t.d_out == nullptr)) {
            bfs::set_error("bfs_trace_pad: table %u: width 1..8, rows <= height, row_stride >= width", k);
            return BFS_ERR_BAD_ARG;
        }
        if (t.kind < 0 || t.kind > 4 || (t.kind == 0 && (t.width < 7 || !t.d_mask0 || !t.d_mask1 || !t.d_mask2)) || (t.kind == 1 && (t.width < 2 || !t.d_mask0 || !t.d_mask1)) ||
            (t.kind == 2 && (t.width < 4 || !t.d_mask0))) {
            bfs::set_error("bfs_trace_pad: table %u: kind 0..4 with its width and masks", k);
            return BFS_ERR_BAD_ARG;
        }
        a.t[k] = t;
        tallest = t.height > tallest ? t.height : tallest;
    }
    if (tallest == 0) return BFS_OK;
    hipLaunchKernelGGL(bfs::trace_pad_kernel, dim3((u32)((tallest + 255) / 256), count), dim3(256), 0, stream, a);
    BFS_HIP(hipGetLastError());
    return BFS_OK;
}
