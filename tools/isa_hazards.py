#!/usr/bin/env python3
"""Scan the gfx950 ISA the library is built from for a VALU carry hazard the compiler cannot see.

gfx950 wants two wait states between a VALU instruction that WRITES an SGPR pair (the carry-out of v_add_co / v_sub_co /
v_mad_u64_u32, a v_cmp result) and a VALU instruction that READS that pair as a carry-in or a select mask (v_addc_co, v_subb_co,
v_subbrev_co, v_cndmask, v_div_fmas).  hipcc pads its own sequences (`s_nop 1`), but it does not look inside inline asm, and
csrc/gl.hpp and csrc/lazy.hpp keep carries in explicit SGPR pairs across asm statements -- so the padding there is ours to get
right (round-3 advice: gl_mul_fused once had a single `s_nop 0` between the two for a constant operand).

    python tools/isa_hazards.py [listing.s ...]      default: the listings of every .hip unit, made by stark_brainfuck_amd.build.build_listings()
                                                      (a device-only -S pass with the library's flags, cached in _build/listings/)

Prints one line per site and exits 1 when there is any.  tests/test_host_logic.py runs it over the shipped build."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEED = 2                      # wait states between the write and the read

SREG = r"(vcc|s\[\d+:\d+\])"
# VALU instructions whose SECOND operand is an SGPR-pair result (carry-out), or whose FIRST is (compares in VOP3 form / implicit vcc)
CARRY_OUT = re.compile(r"^(v_(?:add|sub|subrev|addc|subb|subbrev)_co_u32|v_mad_[ui]64_[ui]32)\s+[^,]+,\s*" + SREG + r"(?=[,\s]|$)")
CMP_OUT = re.compile(r"^v_cmpx?_[a-z0-9_]+\s+" + SREG + r"(?=[,\s]|$)")
CMP_E32 = re.compile(r"^v_cmp_[a-z0-9_]+_e32\b")
# readers: the carry-in / mask is the LAST operand
CARRY_IN = re.compile(r"^(v_(?:addc|subb|subbrev)_co_u32|v_cndmask_b32(?:_e64)?)\s+.*,\s*" + SREG + r"\s*$")
CNDMASK_E32 = re.compile(r"^v_cndmask_b32(?:_e32)?\s+[^,]+,[^,]+,[^,]+$")      # implicit vcc
DIV_FMAS = re.compile(r"^v_div_fmas_")


def instructions(path):
    """(kernel name, [instruction text]) per function of a listing; labels and directives dropped"""
    name, body = None, []
    for raw in open(path, errors="replace"):
        line = raw.split(";")[0].rstrip()
        if not line:
            continue
        if not line[0].isspace():
            m = re.match(r"^([A-Za-z_][\w$.]*):", line)
            if m and not m.group(1).startswith(".L"):
                if name and body:
                    yield name, body
                name, body = m.group(1), []
            continue
        text = line.strip()
        if text.startswith(".") or text.endswith(":"):
            if text.startswith(".Lfunc_end") or text.startswith(".end_amdhsa_kernel"):
                pass
            continue
        body.append(text)
    if name and body:
        yield name, body


def written(text):
    m = CARRY_OUT.match(text)
    if m:
        return m.group(2)
    m = CMP_OUT.match(text)
    if m:
        return m.group(1)
    if CMP_E32.match(text):
        return "vcc"
    return None


def read(text):
    m = CARRY_IN.match(text)
    if m:
        return m.group(2)
    if CNDMASK_E32.match(text) or DIV_FMAS.match(text):
        return "vcc"
    return None


def overlaps(a, b):
    def span(r):
        if r == "vcc":
            return ("vcc", 0, 1)
        lo, hi = map(int, re.findall(r"\d+", r))
        return ("s", lo, hi)
    ka, la, ha = span(a)
    kb, lb, hb = span(b)
    return ka == kb and la <= hb and lb <= ha


def scan(path):
    sites = []
    for kernel, body in instructions(path):
        pending = []                                   # (register, wait states seen since, producer text)
        for text in body:
            r = read(text)
            if r:
                for reg, waited, producer in pending:
                    if overlaps(reg, r) and waited < NEED:
                        sites.append((os.path.basename(path), kernel, producer, text, waited))
            m = re.match(r"^s_nop\s+(\d+)", text)
            states = int(m.group(1)) + 1 if m else 1
            w = written(text)
            # an SALU write of the register ends the VALU-written value's life
            ms = re.match(r"^s_[a-z0-9_]+\s+" + SREG + r"(?=[,\s]|$)", text)
            pending = [(reg, waited + states, prod) for reg, waited, prod in pending
                       if waited + states < NEED and not (ms and overlaps(reg, ms.group(1))) and not (w and overlaps(reg, w))]
            if w:
                pending.append((w, 0, text))
    return sites


def main(argv):
    paths = argv
    if not paths:
        sys.path.insert(0, ROOT)
        from stark_brainfuck_amd import build
        paths = build.build_listings()                 # device-only -S pass with the library's flags (cached by content)
    if not paths:
        print("no listings", file=sys.stderr)
        return 2
    total = 0
    for p in paths:
        for unit, kernel, producer, consumer, waited in scan(p):
            total += 1
            print("%s %s: `%s` -> `%s` with %d wait state(s) between" % (unit, kernel[:60], producer, consumer, waited))
    print("%d listing(s), %d site(s) with fewer than %d wait states between a VALU SGPR write and its VALU carry / mask read" % (len(paths), total, NEED))
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
