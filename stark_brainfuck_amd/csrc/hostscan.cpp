// hostscan.cpp -- the sequential column extensions of the tables (running products and running evaluations over a few
// thousand trace rows), host code like the reference's loops but without a Python interpreter in the loop:
//   ProcessorTable.extend   /root/reference/code/processor_table.py:329-427
//   InstructionTable.extend instruction_table.py:167-231
//   MemoryTable.extend      memory_table.py:172-206
//   IOTable.extend_iotable  io_table.py:77-110
// One primitive covers all of them: over rows i = 0..n-1 with base-field columns x1, x2, x3 and a row mask,
//   kind 0 (running product):     state <- state * (c0 - c1 x1[i] - c2 x2[i] - c3 x3[i])      on masked rows
//   kind 1 (running evaluation):  state <- state * c0 + c1 x1[i] + c2 x2[i] + c3 x3[i]        on masked rows
// and the state is recorded for every row either before or after the row's update, as three limb planes of n words (the layout
// the low-degree extension uploads).
#include <string.h>

#include "../../include/bfstark.h"
#include "gl.hpp"
#include "runtime.hpp"

using namespace bfs;

extern "C" int bfs_xfe_scan(int kind, const uint64_t* x1, const uint64_t* x2, const uint64_t* x3, const uint8_t* mask, uint64_t n,
                            const uint64_t constants[12], const uint64_t initial[3], int record_before, uint64_t* out, uint64_t terminal[3]) {
    if (kind != 0 && kind != 1) { set_error("bfs_xfe_scan: kind must be 0 (product) or 1 (evaluation)"); return BFS_ERR_BAD_ARG; }
    Xfe c[4];
    for (int j = 0; j < 4; ++j) c[j] = Xfe{{constants[3 * j] % GL_P, constants[3 * j + 1] % GL_P, constants[3 * j + 2] % GL_P}};
    Xfe state{{initial[0] % GL_P, initial[1] % GL_P, initial[2] % GL_P}};
    for (uint64_t i = 0; i < n; ++i) {
        if (record_before) { out[i] = state.c[0]; out[n + i] = state.c[1]; out[2 * n + i] = state.c[2]; }
        if (!mask || mask[i]) {
            Xfe lin{{0, 0, 0}};
            if (x1) lin = xfe_add(lin, xfe_scale(c[1], x1[i]));
            if (x2) lin = xfe_add(lin, xfe_scale(c[2], x2[i]));
            if (x3) lin = xfe_add(lin, xfe_scale(c[3], x3[i]));
            state = kind == 0 ? xfe_mul(state, xfe_sub(c[0], lin)) : xfe_add(xfe_mul(state, c[0]), lin);
        }
        if (!record_before) { out[i] = state.c[0]; out[n + i] = state.c[1]; out[2 * n + i] = state.c[2]; }
    }
    terminal[0] = state.c[0]; terminal[1] = state.c[1]; terminal[2] = state.c[2];
    return BFS_OK;
}

// rows x width (row-major: the VM's matrices) -> width columns of `rows` words, dst_stride words apart (the layout the tables pad and
// upload: table.py pads every column to a power of two in staging memory).  Blocked so that both sides stream.
extern "C" int bfs_host_transpose(const uint64_t* src, size_t rows, size_t src_stride, size_t width, uint64_t* dst, size_t dst_stride) {
    if (width > src_stride || rows > dst_stride) { set_error("bfs_host_transpose: width > src_stride or rows > dst_stride"); return BFS_ERR_BAD_ARG; }
    constexpr size_t B = 64;
    for (size_t r0 = 0; r0 < rows; r0 += B) {
        const size_t r1 = r0 + B < rows ? r0 + B : rows;
        for (size_t c = 0; c < width; ++c) {
            uint64_t* d = dst + c * dst_stride;
            const uint64_t* sp = src + c;
            for (size_t r = r0; r < r1; ++r) d[r] = sp[r * src_stride];
        }
    }
    return BFS_OK;
}
