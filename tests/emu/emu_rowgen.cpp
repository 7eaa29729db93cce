// emu_rowgen.cpp -- the generated row walks (csrc/rows_generated.hpp, tools/gen_rows.py) driven on the host: `lanes` simulated lanes
// take one segment at a time in lockstep, and between segments the wave does what row_leaves_generated_kernel (csrc/rows.hip) does:
// if some lane is full, every lane holding a complete block compresses.  Test infrastructure: the product never runs this on the CPU.
#include <stdint.h>
#include <string.h>

#include <vector>

#define __device__
#define __forceinline__ inline
#include "../../stark_brainfuck_amd/csrc/leaf_encode.hpp"
#include "../../stark_brainfuck_amd/csrc/rows_core.hpp"
#include "../../stark_brainfuck_amd/csrc/rows_generated.hpp"

using namespace bfs;

namespace {

template <class G>
struct HostLane {
    RowLane st8;
    unsigned char* buf;
    const uint64_t* values;     // this lane's integers, in preimage order
    const uint64_t* salts;      // this lane's three words
    uint32_t tuple_len;
    uint32_t variant = 0;
    int error = 0;
    void store(uint32_t at, uint64_t data) {
        if (at + 8 > ROW_LANE_BYTES) { error = -3; return; }
        row_store8(buf + at, data);
    }
    void st(uint32_t off, uint64_t data) { store(st8.pos + off, data); }
    void adv(uint32_t n) { st8.pos += n; }
    void framelen(uint32_t off) { store(st8.pos + off, (uint64_t)tuple_len - 11); }
    void salt(uint32_t off, uint32_t w) { store(st8.pos + off, salts[w]); }
    template <uint32_t K, int V = -1> void integer() {
        uint64_t lo, hi;
        uint32_t len;
        row_int_opcode(values[K], lo, hi, len);
        store(st8.pos, lo);
        store(st8.pos + 8, hi);
        st8.pos += len;
    }
    bool full() const { return true; }      // one segment per call: the driver below takes the wave's vote
    void finish() { store(st8.pos, 0); }
};

template <class G>
int run(uint32_t variant, const uint64_t* values, const uint64_t* salts, uint32_t lanes, uint64_t* out, uint32_t* compressions) {
    if (variant >= G::NUM_VARIANTS) return -1;
    const uint32_t nints = G::NUM_INTS[variant];
    std::vector<HostLane<G>> c(lanes);
    std::vector<std::vector<unsigned char>> buf(lanes, std::vector<unsigned char>(ROW_LANE_BYTES, 0xEE));   // stale bytes must not matter
    for (uint32_t l = 0; l < lanes; ++l) {
        c[l].buf = buf[l].data();
        c[l].values = values + (size_t)l * nints;
        c[l].variant = variant;
        c[l].salts = salts + (size_t)3 * l;
        uint32_t ib = 0;
        for (uint32_t j = 0; j < nints; ++j) ib += pickle_int_len(c[l].values[j]);
        c[l].tuple_len = G::TUPLE_CONST_BYTES[variant] + ib;
        row_lane_init(c[l].st8, c[l].tuple_len + G::SALT_BYTES);
    }
    // block 0: constants and the frame length -- the kernel takes BLAKE2b's state behind it from a table the host makes exactly like this
    // (csrc/rows.hip, generated_match), and starts the walk with what the straddling segment appends beyond byte 128
    uint32_t resume = 0, sites = 0;
    for (uint32_t l = 0; l < lanes; ++l) {
        unsigned char block[128];
        memcpy(block, G::BLOCK0, 128);
        const uint64_t frame = (uint64_t)c[l].tuple_len - 11;
        memcpy(block + 3, &frame, 8);
        uint64_t m[16];
        memcpy(m, block, 128);
        blake2b_compress(c[l].st8.h, m, 128, false);
        c[l].st8.consumed = 128;
        c[l].st8.hashed_any = 1;
        unsigned r = 0;
        G::enter_after_block0(c[l], r);
        if (c[l].error) return c[l].error;
        resume = r;
    }
    while (true) {
        if (resume < G::NUM_SEGMENTS) {
            unsigned next = resume;
            for (uint32_t l = 0; l < lanes; ++l) {
                unsigned r = resume;
                G::segments(c[l], r);
                if (c[l].error) return c[l].error;
                next = r;
            }
            if (next <= resume) return -5;          // every call takes exactly one segment here (or the jump to the variant's own part)
            resume = next;
        }
        const bool at_end = resume >= G::NUM_SEGMENTS;
        bool any_full = false, any_want = false;
        for (uint32_t l = 0; l < lanes; ++l) any_full |= row_lane_full(c[l].st8);
        if (!at_end && !any_full) continue;         // (the kernel falls through to the next segment)
        std::vector<char> want(lanes);
        for (uint32_t l = 0; l < lanes; ++l) {
            want[l] = at_end ? row_lane_wants_end(c[l].st8) : row_lane_wants_mid(c[l].st8, true);
            if (!at_end && row_lane_full(c[l].st8) && !want[l]) return -2;
            any_want |= want[l];
        }
        if (at_end && !any_want) break;
        ++sites;
        for (uint32_t l = 0; l < lanes; ++l)
            if (want[l]) row_lane_compress(c[l].st8, c[l].buf, at_end);
    }
    for (uint32_t l = 0; l < lanes; ++l) memcpy(out + (size_t)8 * l, c[l].st8.h, 64);
    if (compressions) *compressions = sites;
    return 0;
}

}  // namespace

// values: lanes x NUM_INTS[variant] integers in preimage order; salts: lanes x 3 words; out: lanes x 8 words.  info = {NUM_VARIANTS,
// NUM_INTS[variant], TUPLE_CONST_BYTES[variant], SALT_BYTES, CODES[variant]} of the layout (also with lanes == 0).  Returns 0, -1 for an
// unknown layout or variant, or another negative number when a store would leave the lane's buffer / a full lane could not compress
// (which the kernel's invariants exclude).
template <class G>
static int entry(uint32_t variant, const uint64_t* values, const uint64_t* salts, uint32_t lanes, uint64_t* out, uint32_t* info, uint32_t* compressions) {
    if (variant >= G::NUM_VARIANTS) return -1;
    if (info) { info[0] = G::NUM_VARIANTS; info[1] = G::NUM_INTS[variant]; info[2] = G::TUPLE_CONST_BYTES[variant]; info[3] = G::SALT_BYTES; info[4] = G::CODES[variant]; }
    return lanes ? run<G>(variant, values, salts, lanes, out, compressions) : 0;
}

extern "C" int emu_rowgen_leaves(uint32_t layout, uint32_t variant, const uint64_t* values, const uint64_t* salts, uint32_t lanes, uint64_t* out,
                                 uint32_t* info, uint32_t* compressions) {
    if (layout == 0) return entry<rowgen::Layout0>(variant, values, salts, lanes, out, info, compressions);
    if (layout == 1) return entry<rowgen::Layout1>(variant, values, salts, lanes, out, info, compressions);
    return -1;
}
