// emu_rows.cpp -- the per-lane half of the zipped-row leaf kernel (csrc/rows_core.hpp) driven on the host: `lanes` simulated
// lanes walk one flattened template in lockstep exactly as row_leaves_kernel does (same compression site, same ROW_UNROLL steps per
// turn, __any replaced by a loop over the lanes).  Test infrastructure: the product never runs this on the CPU.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../stark_brainfuck_amd/csrc/leaf_encode.hpp"
#include "../../stark_brainfuck_amd/csrc/rows_core.hpp"

using namespace bfs;

// steps: nsteps x {kind, a, data} (nsteps a multiple of ROW_STEP_PAD); an INT step's `a` is the index of the integer in `values`
// (lanes x nints, row-major); salts: lanes x 3 words; out: lanes x 8 words.  Returns 0, or a negative number when a buffer
// would overflow / a full lane could not compress (which the kernel's invariants exclude).
extern "C" int emu_row_leaves(const uint32_t* kinds, const uint32_t* as, const uint64_t* datas, uint32_t nsteps, const uint64_t* values, uint32_t nints,
                              const uint64_t* salts, uint32_t tuple_const_bytes, uint32_t salt_bytes, uint32_t lanes, uint64_t* out, uint32_t* max_skew) {
    if (nsteps % ROW_STEP_PAD) return -1;
    std::vector<RowLane> st(lanes);
    std::vector<std::vector<unsigned char>> buf(lanes, std::vector<unsigned char>(ROW_LANE_BYTES, 0xEE));   // stale bytes must not matter
    std::vector<uint64_t> int_hi(lanes, 0);
    std::vector<uint32_t> int_hi_bytes(lanes, 0), tuple_len(lanes);
    for (uint32_t l = 0; l < lanes; ++l) {
        uint32_t ib = 0;
        for (uint32_t j = 0; j < nints; ++j) ib += pickle_int_len(values[(size_t)l * nints + j]);
        tuple_len[l] = tuple_const_bytes + ib;
        row_lane_init(st[l], tuple_len[l] + salt_bytes);
    }
    uint32_t k = 0, skew = 0;
    bool input_done = false;
    while (true) {
        std::vector<char> want(lanes);
        if (!input_done) {
            bool any_full = false;
            uint32_t lo = ~0u, hi = 0;
            for (uint32_t l = 0; l < lanes; ++l) {
                any_full |= row_lane_full(st[l]);
                const uint32_t at = st[l].consumed + st[l].pos;
                lo = at < lo ? at : lo; hi = at > hi ? at : hi;
            }
            skew = hi - lo > skew ? hi - lo : skew;
            for (uint32_t l = 0; l < lanes; ++l) {
                want[l] = row_lane_wants_mid(st[l], any_full);
                if (row_lane_full(st[l]) && !want[l]) return -2;
            }
        } else {
            bool any = false;
            for (uint32_t l = 0; l < lanes; ++l) { want[l] = row_lane_wants_end(st[l]); any |= want[l]; }
            if (!any) break;
        }
        for (uint32_t l = 0; l < lanes; ++l)
            if (want[l]) row_lane_compress(st[l], buf[l].data(), input_done);
        if (input_done) continue;
        if (k < nsteps) {
            for (uint32_t u = 0; u < ROW_UNROLL; ++u, ++k) {
                for (uint32_t l = 0; l < lanes; ++l) {
                    uint64_t data;
                    uint32_t nb;
                    if (kinds[k] == SEG_CONST) { data = datas[k]; nb = as[k]; }
                    else if (kinds[k] == SEG_INT) {
                        uint32_t len;
                        row_int_opcode(values[(size_t)l * nints + as[k]], data, int_hi[l], len);
                        uint64_t d2, h2; uint32_t l2;
                        if (values[(size_t)l * nints + as[k]] >= (1ull << 31)) {         // the branch-free form must agree where it applies
                            row_long1_opcode(values[(size_t)l * nints + as[k]], d2, h2, l2);
                            if (d2 != data || h2 != int_hi[l] || l2 != len) return -4;
                        }
                        nb = len < 8 ? len : 8;
                        int_hi_bytes[l] = len > 8 ? len - 8 : 0;
                    } else if (kinds[k] == SEG_INT_HI) { data = int_hi[l]; nb = int_hi_bytes[l]; }
                    else if (kinds[k] == SEG_FRAMELEN) { data = (uint64_t)tuple_len[l] - 11; nb = 8; }
                    else { data = salts[(size_t)3 * l + as[k]]; nb = 8; }
                    if (st[l].pos + 8 > ROW_LANE_BYTES) return -3;
                    row_lane_put(st[l], buf[l].data(), data, nb);
                }
            }
        } else {
            for (uint32_t l = 0; l < lanes; ++l) {
                if (st[l].pos + 8 > ROW_LANE_BYTES) return -3;
                row_lane_finish(st[l], buf[l].data());
            }
            input_done = true;
        }
    }
    for (uint32_t l = 0; l < lanes; ++l) memcpy(out + (size_t)8 * l, st[l].h, 64);
    if (max_skew) *max_skew = skew;
    return 0;
}
