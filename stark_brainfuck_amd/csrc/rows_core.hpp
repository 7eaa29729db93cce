// rows_core.hpp -- the per-lane half of the zipped-row leaf kernel (rows.hip): how one row's preimage,
//     pickle.dumps(tuple of the row's elements) || pickle.dumps(salt)      /root/reference/code/salted_merkle.py:32-35
// is produced from a flattened template and fed to BLAKE2b block by block without ever existing in memory.
//
// A template is a list of STEPS of at most 8 preimage bytes each (constant bytes carried in the step; an integer of the row is two
// steps: the first eight bytes of its opcode, then the rest).  All lanes of a wave take the same step at the same time; what
// differs between them is how many bytes their integers took, i.e. where in its block buffer a lane stands.  A lane's buffer is
// ROW_LANE_BYTES (> 128) of LDS, written with unaligned 8-byte stores at the lane's own byte position (bytes of `data` above the
// step's length are zero and are overwritten by the next store), so there is no shifting of partial words in registers.  The wave
// compresses TOGETHER when the lane furthest ahead has filled its buffer; a lane that does not hold a complete block by then sits
// that compression out (block counters are per lane), and a lane's last, padded block is compressed after the end of the input.
//
// These functions are pure functions of a lane's state and compile for the host as well: tests/emu/emu_rows.cpp walks random
// templates with 64 simulated lanes and compares every digest with hashlib (test infrastructure only).
#pragma once
#include <string.h>

#include "blake2b.hpp"
#include "gl.hpp"

namespace bfs {

enum { SEG_CONST = 0, SEG_INT = 1, SEG_FRAMELEN = 2, SEG_SALT = 3, SEG_INT_HI = 4 };

// what the kernel walks: the template flattened into steps of at most 8 preimage bytes each
struct RowStep {
    u32 kind;
    u32 a;      // CONST: number of bytes (0..8); INT / INT_HI: column; SALT: word 0..2
    u64 data;   // CONST: the bytes (zero padded); INT: limb | (column | limb << 8 of the integer ROW_PREFETCH places further on) << 8
                //        | (index of this integer mod ROW_PREFETCH) << 24
};
// integers of a row are requested ROW_PREFETCH places ahead of where they are written; templates are padded to ROW_STEP_PAD steps
constexpr u32 ROW_PREFETCH = 3, ROW_STEP_PAD = 4;
#ifndef BFS_ROW_UNROLL
#define BFS_ROW_UNROLL 2
#endif
constexpr u32 ROW_UNROLL = BFS_ROW_UNROLL;              // steps per turn of the kernel's loop (one scalar load of their descriptors)
// Round 3 (profiles/r03/ab_rows_occupancy.txt): 152 bytes per lane (19 words; an odd word count keeps neighbouring lanes off the same LDS
// banks) and two steps per turn instead of 200 and four: 38 KiB per 256-row workgroup, so FOUR workgroups fit a CU where three did.  A lane
// is "full" 16 bytes before the end of its buffer, i.e. above 136: lanes that trail the wave's leader by more than 8 bytes sit that
// compression out (the integers of a row of random field elements are 10 or 11 bytes each: ~13 bytes of spread over a 27-integer row),
// which costs less than the fourth wave per SIMD brings: 2 x 2^22 rows 5.74 -> 5.30 ms.
#ifndef BFS_ROW_LANE_BYTES
#define BFS_ROW_LANE_BYTES 152
#endif
constexpr u32 ROW_LANE_BYTES = BFS_ROW_LANE_BYTES;
static_assert(ROW_STEP_PAD % ROW_UNROLL == 0 && ROW_LANE_BYTES % 8 == 0 && ROW_LANE_BYTES - 8 * ROW_UNROLL >= 128,
              "a lane that must compress before the next ROW_UNROLL stores holds a complete block");

struct RowLane {
    u64 h[8];
    u32 pos;            // bytes in the buffer
    u32 consumed;       // bytes already compressed
    u32 total;          // length of the preimage
    u32 hashed_any;
};

BFS_HD void row_lane_init(RowLane& s, u32 total) {
    blake2b_init(s.h);
    s.pos = 0; s.consumed = 0; s.total = total; s.hashed_any = 0;
}

// the next ROW_UNROLL stores would not fit
BFS_HD bool row_lane_full(const RowLane& s) { return s.pos > ROW_LANE_BYTES - 8 * ROW_UNROLL; }

// while the input is still coming: compress now?  (any_full: some lane of the wave is full)  A full lane always says yes: it
// holds more than 128 bytes of the message, so its block is complete and not the last one.
BFS_HD bool row_lane_wants_mid(const RowLane& s, bool any_full) { return any_full && s.pos >= 128 && s.consumed + 128 < s.total; }

// after the end of the input: blocks left?  (an empty message is one padded block)
BFS_HD bool row_lane_wants_end(const RowLane& s) { return s.consumed < s.total || !s.hashed_any; }

BFS_HD void row_store8(unsigned char* p, u64 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef u64 __attribute__((aligned(1))) unaligned_u64;
    *(unaligned_u64*)p = v;                             // ds_write_b64 at any byte address (gfx950: unaligned LDS access)
#else
    memcpy(p, &v, 8);
#endif
}

// append the low nb (0..8) bytes of data; bytes of data above nb must be zero
BFS_HD void row_lane_put(RowLane& s, unsigned char* buf, u64 data, u32 nb) {
    row_store8(buf + s.pos, data);
    s.pos += nb;
}

// end of the input: zeros behind the last byte, so that the padded block can be read word by word
BFS_HD void row_lane_finish(RowLane& s, unsigned char* buf) { row_store8(buf + s.pos, 0); }

// compress the block at the front of the buffer and move what lies behind it to the front
BFS_HD void row_lane_compress(RowLane& s, unsigned char* buf, bool at_end) {
    const u64* w = (const u64*)buf;
    const bool last = at_end && s.consumed + 128 >= s.total;
    const u32 valid = last ? s.total - s.consumed : 128;            // message bytes in this block
    u64 m[16];
    BFS_UNROLL
    for (int j = 0; j < 16; ++j) m[j] = (u32)(8 * j) < valid ? w[j] : 0;
    blake2b_compress(s.h, m, last ? (u64)s.total : (u64)s.consumed + 128, last);
    s.consumed += 128;
    s.hashed_any = 1;
    u64* ww = (u64*)buf;
    BFS_UNROLL
    for (u32 j = 0; j < ROW_LANE_BYTES / 8 - 16; ++j)
        if (128 + 8 * j < s.pos) ww[j] = ww[16 + j];
    s.pos = s.pos > 128 ? s.pos - 128 : 0;
}

// pickle's integer opcodes as CPython's save_long writes them: BININT1 / BININT2 / BININT / LONG1.  lo = the first eight bytes
// of the opcode (zero padded), hi = the rest (LONG1 of 7..9 value bytes), len = its length
BFS_HD void row_int_opcode(u64 v, u64& lo, u64& hi, u32& len) {
    hi = 0;
    if (v < (1ull << 8)) { lo = 0x4b | (v << 8); len = 2; }
    else if (v < (1ull << 16)) { lo = 0x4d | (v << 8); len = 3; }
    else if (v < (1ull << 31)) { lo = 0x4a | (v << 8); len = 5; }
    else {
        const u32 nn = (64 - (u32)__builtin_clzll(v)) / 8 + 1;
        lo = 0x8a | ((u64)nn << 8) | (v << 16);
        hi = v >> 48;
        len = 2 + nn;
    }
}
// the same when v >= 2^31 is known (no branches)
BFS_HD void row_long1_opcode(u64 v, u64& lo, u64& hi, u32& len) {
    const u32 nn = (64 - (u32)__builtin_clzll(v)) / 8 + 1;
    lo = 0x8a | ((u64)nn << 8) | (v << 16);
    hi = v >> 48;
    len = 2 + nn;
}

}  // namespace bfs
