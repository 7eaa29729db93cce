#!/usr/bin/env python3
"""Generates stark_brainfuck_amd/csrc/rows_generated.hpp: the zipped-row leaf kernel's template walk (csrc/rows.hip,
rows_core.hpp) unrolled into straight-line code for the column layouts the prover commits to
(/root/reference/code/brainfuck_stark.py:178-179 and :197-198): the randomizer codeword plus the 16 base columns, and the
9 extension columns in the nine row patterns (coefficients stored per extension element) LAYOUTS lists; the patterns of a
layout share their segments up to the first column they differ in and then go their own way (`TAIL_FIRST`).

The template itself comes from the library (bfs_row_template_steps: the generic pickle emitter run on a row of sentinels),
so the generated code and the interpreter kernel describe the same bytes; the header carries the template's hash and the
library uses a generated kernel only for a template that hashes alike (any other layout, and any row of another
pattern, goes through the interpreter kernel as before).

A SEGMENT is a run of stores whose start offsets (from the lane's byte position when the segment begins) are all <= 8, so
that a lane that was not "full" (rows_core.hpp: position <= ROW_LANE_BYTES - 16) stays inside its buffer; after every
segment the wave asks whether some lane is full and, if so, leaves for the one compression site and comes back to the
next segment.  An integer is a segment of its own (two stores, 0 and 8; up to 11 bytes).

Block 0 of the preimage (128 bytes: constants and the frame length, no integer yet) is emitted too: the library tabulates
BLAKE2b's state behind it per frame length and the kernels start the walk behind it (BLOCK0, AFTER_BLOCK0,
enter_after_block0).

    python tools/gen_rows.py            (needs the built library: python -c "import __graft_entry__ as g; g.build()")
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "stark_brainfuck_amd", "csrc", "rows_generated.hpp")

SEG_CONST, SEG_INT, SEG_FRAMELEN, SEG_SALT, SEG_INT_HI = range(5)

# (name, columns: 1 = extension column, salted, patterns).  A pattern is 2 bits per extension column: how many coefficients the
# column's elements store.  The columns of a commitment are codewords of low-degree polynomials, so a column is "3" unless its
# polynomial is constant: the evaluation columns of the input and output tables are 0 without symbols (the zero polynomial), 1 with
# one symbol (a base-field constant) and 3 from two symbols on -- nine patterns for the extension commitment, the last two columns.
def _ext_patterns():
    head = sum(3 << (2 * j) for j in range(7))
    return [head | (a << 14) | (b << 16) for a in (0, 1, 3) for b in (0, 1, 3)]


LAYOUTS = [
    ("base commitment: randomizer codeword + 16 base columns (brainfuck_stark.py:178)", [1] + [0] * 16, True, [3]),
    ("extension commitment: 9 extension columns (brainfuck_stark.py:197)", [1] * 9, True, _ext_patterns()),
]


def template(lib, RowColumn, layout, salted, code):
    n = len(layout)
    cols = (RowColumn * n)()
    for c, e in enumerate(layout):
        cols[c].d_values = None
        cols[c].is_ext = e
        cols[c].field_id = 0
    hdr = (ctypes.c_uint32 * 4)()
    steps = np.zeros(2 * 4096, dtype=np.uint64)
    ints = np.zeros(256, dtype=np.uint32)
    h = ctypes.c_uint64()
    rc = lib.bfs_row_template_steps(cols, n, code, 1 if salted else 0, hdr, steps.ctypes.data, 4096, ints.ctypes.data, 256, ctypes.byref(h))
    if rc != 0:
        raise RuntimeError("bfs_row_template_steps: %d" % rc)
    out = []
    for k in range(hdr[0]):
        out.append((int(steps[2 * k]) & 0xFFFFFFFF, int(steps[2 * k]) >> 32, int(steps[2 * k + 1])))
    return code, list(hdr), out, [int(x) for x in ints[:hdr[1]]], h.value


def segments(steps):
    """[(kind, ...)] units grouped so that every store of a group starts at most 8 bytes after the group's first"""
    segs, cur, off = [], [], 0
    int_index = 0

    def close():
        nonlocal cur, off
        if cur:
            segs.append(cur)
        cur, off = [], 0

    for kind, a, data in steps:
        if kind == SEG_CONST:
            if a == 0:
                continue                      # padding of the interpreter's step list
            if off > 8:
                close()
            cur.append(("st", off, data, a))
            off += a
        elif kind == SEG_FRAMELEN:
            if off > 8:
                close()
            cur.append(("framelen", off))
            off += 8
        elif kind == SEG_SALT:
            if off > 8:
                close()
            cur.append(("salt", off, a))
            off += 8
        elif kind == SEG_INT:
            close()
            cur.append(("int", int_index))
            int_index += 1
            off = 99                          # nothing may follow an integer: its length is the lane's own
        elif kind == SEG_INT_HI:
            pass                              # part of the integer unit
        else:
            raise ValueError(kind)
    close()
    return segs


def statement(seg, variant):
    """the C++ text of one segment's stores (without its case label and its check)"""
    parts, adv = [], 0
    for u in seg:
        if u[0] == "st":
            parts.append("c.st(%d, 0x%xull);" % (u[1], u[2]))
            adv = u[1] + u[3]
        elif u[0] == "framelen":
            parts.append("c.framelen(%d);" % u[1])
            adv = u[1] + 8
        elif u[0] == "salt":
            parts.append("c.salt(%d, %d);" % (u[1], u[2]))
            adv = u[1] + 8
        else:
            parts.append("c.template integer<%d%s>();" % (u[1], "" if variant is None else ", %d" % variant))
            adv = None
    if adv is not None:
        parts.append("c.adv(%d);" % adv)
    return " ".join(parts)


def constant_prefix(steps):
    """the preimage bytes in front of the row's first integer, the eight bytes of the frame length as zeros"""
    out = bytearray()
    for kind, a, data in steps:
        if kind == SEG_CONST:
            out += data.to_bytes(8, "little")[:a]
        elif kind == SEG_FRAMELEN:
            out += bytes(8)
        else:
            break
    return bytes(out)


def segment_bytes(seg):
    """bytes a segment of constants appends (None for an integer)"""
    n = 0
    for u in seg:
        if u[0] == "st":
            n = u[1] + u[3]
        elif u[0] in ("framelen", "salt"):
            n = u[1] + 8
        else:
            return None
    return n


def ints_in(segs):
    return sum(1 for seg in segs for u in seg if u[0] == "int")


def main():
    from stark_brainfuck_amd import _lib

    lib = _lib.load()
    w = []
    w.append("// rows_generated.hpp -- GENERATED by tools/gen_rows.py from the row templates of csrc/rows.hip; do not edit.")
    w.append("// The template walk of row_leaves_kernel unrolled for the prover's two column layouts: one function per layout, a switch over")
    w.append("// segments with fall-through (a coroutine: the caller compresses and comes back with `resume` = the segment to go on with).")
    w.append("// A layout has one VARIANT per row pattern it knows; the variants share the segments up to the first column they differ in.")
    w.append("#pragma once")
    w.append("")
    w.append("namespace bfs {")
    w.append("namespace rowgen {")
    w.append("")
    for li, (name, layout, salted, codes) in enumerate(LAYOUTS):
        variants = []
        for code in codes:
            _, hdr, steps, ints, h = template(lib, _lib.RowColumn, layout, salted, code)
            variants.append((code, hdr, segments(steps), ints, h))
        nv = len(variants)
        common = 0
        if nv > 1:
            shortest = min(len(v[2]) for v in variants)
            while common < shortest and all(statement(v[2][common], None) == statement(variants[0][2][common], None) for v in variants):
                common += 1
            # an integer of the shared part must be the same integer in every variant
            k = ints_in(variants[0][2][:common])
            assert all(v[3][:k] == variants[0][3][:k] for v in variants)
            common_ints = k
        else:
            common = len(variants[0][2])
            common_ints = len(variants[0][3])
        max_ints = max(len(v[3]) for v in variants)
        w.append("// layout %d: %s" % (li, name))
        w.append("//   columns %s, %s; %d pattern%s, the first %d segments (%d integers) shared" % (
            "".join("E" if e else "B" for e in layout), "salted" if salted else "unsalted", nv, "" if nv == 1 else "s", common, common_ints))
        for vi, (code, hdr, segs, ints, h) in enumerate(variants):
            w.append("//   variant %d: pattern 0x%x, %d steps, %d integers, %d constant bytes + %d bytes of salt pickle" % (vi, code, hdr[0], hdr[1], hdr[2], hdr[3]))
        # case numbering: shared segments 0..common-1, [the split: common], then every variant's tail and its finish
        case = common + (1 if nv > 1 else 0)
        tail_first, finish_case = [], []
        for vi, v in enumerate(variants):
            tail_first.append(case)
            case += len(v[2]) - common
            finish_case.append(case)
            case += 1
        num_cases = case
        w.append("struct Layout%d {" % li)
        w.append("    static constexpr unsigned NUM_VARIANTS = %du, MAX_INTS = %du, COMMON_INTS = %du, SALT_BYTES = %du, NUM_SEGMENTS = %du;" % (
            nv, max_ints, common_ints, variants[0][1][3], num_cases))
        assert all(v[1][3] == variants[0][1][3] for v in variants)
        w.append("    static constexpr unsigned long long HASHES[%d] = {%s};" % (nv, ", ".join("0x%016xull" % v[4] for v in variants)))
        w.append("    static constexpr unsigned CODES[%d] = {%s};" % (nv, ", ".join("0x%xu" % v[0] for v in variants)))
        w.append("    static constexpr unsigned NUM_INTS[%d] = {%s};" % (nv, ", ".join("%du" % len(v[3]) for v in variants)))
        w.append("    static constexpr unsigned TUPLE_CONST_BYTES[%d] = {%s};" % (nv, ", ".join("%du" % v[1][2] for v in variants)))
        w.append("    static constexpr unsigned TAIL_FIRST[%d] = {%s};" % (nv, ", ".join("%du" % t for t in tail_first)))
        w.append("    // column | limb << 8 of every integer of the row, in preimage order (padded with the last one)")
        w.append("    static constexpr unsigned INTS[%d][%d] = {" % (nv, max_ints))
        for v in variants:
            padded = v[3] + [v[3][-1]] * (max_ints - len(v[3]))
            w.append("        {%s}," % ", ".join("0x%x" % x for x in padded))
        w.append("    };")
        # block 0 of the preimage is constant but for the frame length (the first integer comes later, in every variant alike): its
        # BLAKE2b state is tabulated per length on the host, and the walk starts behind it
        prefixes = [constant_prefix(template(lib, _lib.RowColumn, layout, salted, v[0])[2]) for v in variants]
        assert all(len(q) >= 128 and q[:128] == prefixes[0][:128] for q in prefixes), "block 0 must be the same constant in every variant"
        block0 = prefixes[0][:128]
        assert block0[3:11] == bytes(8) and block0[2] == 0x95
        cum, k = 0, 0
        while True:
            nb = segment_bytes(variants[0][2][k])
            assert nb is not None and k < common, "block 0 ends inside the shared constants"
            if cum + nb >= 128:
                break
            cum += nb
            k += 1
        leftover = prefixes[0][128:cum + nb]                    # what segment k appends behind byte 128
        assert len(leftover) < 16
        w.append("    // block 0 (128 bytes, frame length at 3..10 zero) as little-endian words; the host tabulates BLAKE2b's state behind it per frame length")
        w.append("    static constexpr unsigned long long BLOCK0[16] = {%s};" % ", ".join("0x%016xull" % int.from_bytes(block0[8 * i:8 * i + 8], "little") for i in range(16)))
        w.append("    static constexpr unsigned AFTER_BLOCK0 = %du;        // the segment the walk goes on with" % (k + 1))
        w.append("    // a lane that starts behind block 0: the %d bytes segment %d appends beyond byte 128" % (len(leftover), k))
        parts = []
        for i in range(0, len(leftover), 8):
            parts.append("c.st(%d, 0x%xull);" % (i, int.from_bytes(leftover[i:i + 8], "little")))
        parts.append("c.adv(%d);" % len(leftover))
        w.append("    template <class C> static __device__ __forceinline__ void enter_after_block0(C& c, unsigned& resume) { %s resume = AFTER_BLOCK0; }" % " ".join(parts))
        w.append("    template <class C> static __device__ __forceinline__ void segments(C& c, unsigned& resume) {")
        w.append("        switch (resume) {")
        for si in range(common):
            lead = "        case %d: " % si if si == 0 else "        [[fallthrough]]; case %d: " % si
            w.append(lead + statement(variants[0][2][si], None) + " if (c.full()) { resume = %d; return; }" % (si + 1))
        if nv > 1:
            w.append("        [[fallthrough]]; case %d: resume = TAIL_FIRST[c.variant]; return;" % common)
        for vi, v in enumerate(variants):
            tail = v[2][common:]
            if nv > 1:
                w.append("        // variant %d (pattern 0x%x)" % (vi, v[0]))
            for ti, seg in enumerate(tail):
                si = tail_first[vi] + ti
                lead = "        case %d: " % si if (ti == 0 and nv > 1) else "        [[fallthrough]]; case %d: " % si
                w.append(lead + statement(seg, vi if nv > 1 else None) + " if (c.full()) { resume = %d; return; }" % (si + 1))
            lead = "        case %d: " % finish_case[vi] if (not tail and nv > 1) else "        [[fallthrough]]; case %d: " % finish_case[vi]
            w.append(lead + "c.finish(); resume = %d; return;" % num_cases)
        w.append("        default: return;")
        w.append("        }")
        w.append("    }")
        w.append("};")
        w.append("")
    w.append("constexpr unsigned NUM_LAYOUTS = %d;" % len(LAYOUTS))
    w.append("")
    w.append("}  // namespace rowgen")
    w.append("}  // namespace bfs")
    text = "\n".join(w) + "\n"
    if not os.path.exists(OUT) or open(OUT).read() != text:
        with open(OUT, "w") as f:
            f.write(text)
    print("wrote %s (%d layouts)" % (OUT, len(LAYOUTS)))


if __name__ == "__main__":
    main()
