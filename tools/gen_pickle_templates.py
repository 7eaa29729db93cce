#!/usr/bin/env python3
"""Generates stark_brainfuck_amd/csrc/pickle_templates.hpp: the constant byte segments of the pickle
(protocol 4) of one standalone Merkle leaf of the reference --

    pickle.dumps(ExtensionFieldElement)   (leaf of every FRI round tree, /root/reference/code/merkle.py:30)
    pickle.dumps(BaseFieldElement)

split around the variable-length integer opcodes, so that the GPU leaf kernel can re-assemble the exact byte
stream from the three 64-bit limbs.  The segments are obtained from CPython's own pickler applied to objects
whose classes carry the reference's module / class / attribute names (no reference code is involved), and the
re-assembly rule is verified here against that pickler over a few thousand values before the header is written.

    python tools/gen_pickle_templates.py
"""
import os
import pickle
import random
import sys
import types

P = 18446744069414584321
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stark_brainfuck_amd", "csrc", "pickle_templates.hpp")


def make_classes():
    mods = {n: types.ModuleType(n) for n in ("algebra", "univariate", "extension_field")}

    class BaseField:
        def __init__(self, p): self.p = p

    class BaseFieldElement:
        def __init__(self, value, field): self.value = value; self.field = field

    class Polynomial:
        def __init__(self, coefficients): self.coefficients = list(coefficients)

    class ExtensionField:
        def __init__(self, modulus): self.modulus = modulus

    class ExtensionFieldElement:
        def __init__(self, polynomial, field): self.polynomial = polynomial; self.field = field

    for mod, classes in (("algebra", (BaseField, BaseFieldElement)), ("univariate", (Polynomial,)),
                         ("extension_field", (ExtensionField, ExtensionFieldElement))):
        for c in classes:
            c.__module__ = mods[mod].__name__
            c.__qualname__ = c.__name__
            setattr(mods[mod], c.__name__, c)
        sys.modules[mod] = mods[mod]
    return BaseField, BaseFieldElement, Polynomial, ExtensionField, ExtensionFieldElement


BaseField, BaseFieldElement, Polynomial, ExtensionField, ExtensionFieldElement = make_classes()
BF = BaseField(P)
_one = BaseFieldElement(1, BF)
XF = ExtensionField(Polynomial([_one, BaseFieldElement(P - 1, BF), BaseFieldElement(0, BF), _one]))
BF_STANDALONE = BaseField(P)


def xfe(limbs):
    l = list(limbs)
    while l and l[-1] == 0:
        l.pop()
    return ExtensionFieldElement(Polynomial([BaseFieldElement(v, BF) for v in l]), XF)


def dumps(o):
    return pickle.dumps(o, protocol=4)


def enc_int(v):
    """pickle's integer opcodes for 0 <= v < 2^64 (save_long in Modules/_pickle.c)."""
    if v < 1 << 8:
        return bytes([0x4b, v])
    if v < 1 << 16:
        return bytes([0x4d]) + v.to_bytes(2, "little")
    if v < 1 << 31:
        return bytes([0x4a]) + v.to_bytes(4, "little")
    nn = v.bit_length() // 8 + 1
    return bytes([0x8a, nn]) + v.to_bytes(nn, "little")


def split(bs, values):
    """split the pickle body around the encodings of `values` (in order)."""
    parts, pos = [], 0
    for v in values:
        e = enc_int(v)
        i = bs.index(e, pos)
        assert bs.find(e, i + 1) == -1 or True
        parts.append(bs[pos:i])
        pos = i + len(e)
    parts.append(bs[pos:])
    return parts


S = [0xA1B2C3D4E5F60718, 0x9182736455463728, 0xF1E2D3C4B5A69788]   # sentinels (9-byte LONG1 form)

k3 = split(dumps(xfe(S)), S)
k2 = split(dumps(xfe(S[:2])), S[:2])
k1 = split(dumps(xfe(S[:1])), S[:1])
k0 = dumps(xfe([]))
b1 = split(dumps(BaseFieldElement(S[0], BF_STANDALONE)), S[:1])

HDR = 11   # 80 04 95 + u64 frame length
for parts in (k3, k2, k1, b1):
    assert parts[0][:3] == b"\x80\x04\x95"
pre3, mid_a, mid_b, post3 = k3[0][HDR:], k3[1], k3[2], k3[3]
pre2, mid2_a, post2 = k2[0][HDR:], k2[1], k2[2]
pre1, post1 = k1[0][HDR:], k1[1]
assert pre2 == pre3 and mid2_a == mid_a
# k = 1 has no MARK (0x28) in front of the first coefficient
mark = [i for i in range(len(pre3)) if pre3[:i] + pre3[i + 1:] == pre1 and pre3[i] == 0x28]
assert len(mark) >= 1
mark = mark[-1]
pre_a, pre_b = pre3[:mark], pre3[mark + 1:]
assert pre_a + pre_b == pre1 and pre_a + b"\x28" + pre_b == pre3
bfe_pre, bfe_post = b1[0][HDR:], b1[1]


def assemble_xfe(limbs):
    l = list(limbs)
    while l and l[-1] == 0:
        l.pop()
    k = len(l)
    if k == 0:
        return k0
    body = pre_a + (b"\x28" if k >= 2 else b"") + pre_b + enc_int(l[0])
    if k == 1:
        body += post1
    else:
        body += mid_a + enc_int(l[1])
        body += post2 if k == 2 else mid_b + enc_int(l[2]) + post3
    return b"\x80\x04\x95" + len(body).to_bytes(8, "little") + body


def assemble_bfe(v):
    body = bfe_pre + enc_int(v) + bfe_post
    return b"\x80\x04\x95" + len(body).to_bytes(8, "little") + body


rng = random.Random(1)
edges = [0, 1, 255, 256, 65535, 65536, (1 << 31) - 1, 1 << 31, (1 << 32) - 1, 1 << 32, (1 << 39), (1 << 40) - 1, (1 << 47),
         (1 << 48) - 1, (1 << 55), (1 << 56) - 1, 1 << 56, (1 << 63) - 1, 1 << 63, P - 1]
for trial in range(4000):
    l = [rng.choice(edges) if rng.random() < 0.5 else rng.randrange(P) >> rng.randrange(64) for _ in range(3)]
    if trial % 7 == 0:
        l = l[:rng.randrange(4)]
    assert assemble_xfe(l) == dumps(xfe(l)), l
    assert assemble_bfe(l[0] if l else 0) == dumps(BaseFieldElement(l[0] if l else 0, BF_STANDALONE))


def words(bs):
    padded = bs + bytes((-len(bs)) % 8)
    return [int.from_bytes(padded[i:i + 8], "little") for i in range(0, len(padded), 8)]


def emit(name, bs):
    w = words(bs)
    body = ", ".join("0x%016xULL" % x for x in w) if w else "0"
    return ("constexpr int %s_LEN = %d;\nBFS_TPL_CONST u64 %s[%d] = {%s};\n" % (name, len(bs), name, max(len(w), 1), body))


with open(OUT, "w") as f:
    f.write("// pickle_templates.hpp -- GENERATED by tools/gen_pickle_templates.py, do not edit.\n"
            "// Constant segments of pickle.dumps(leaf) (protocol 4) for the reference's element classes, split around the\n"
            "// integer opcodes; little-endian 8-byte words, zero padded.  Layout of a leaf with k stored coefficients:\n"
            "//   k=0: XFE_K0 (whole pickle)\n"
            "//   k>=1: 80 04 95 <u64 frame_len> XFE_PRE_A [28 if k>=2] XFE_PRE_B INT(c0)\n"
            "//         k=1: XFE_POST1 | k=2: XFE_MID_A INT(c1) XFE_POST2 | k=3: XFE_MID_A INT(c1) XFE_MID_B INT(c2) XFE_POST3\n"
            "//   base element: 80 04 95 <u64 frame_len> BFE_PRE INT(value) BFE_POST\n"
            "// (see SURVEY.md Appendix A; replaces pickle.dumps at /root/reference/code/merkle.py:30)\n"
            "#pragma once\n#include \"gl.hpp\"\n\nnamespace bfs {\nnamespace tpl {\n\n"
            "#if defined(__HIP_DEVICE_COMPILE__)\n#define BFS_TPL_CONST __constant__ const\n#else\n#define BFS_TPL_CONST static const\n#endif\n\n")
    # first BLAKE2b block of a leaf = 11-byte header + PRE_A (113) + [MARK] + the first 3 (k >= 2) or 4 (k = 1) bytes of PRE_B:
    # constant up to the length field, so its compression is tabulated per length (merkle_core.hpp leaf_midstates)
    assert 11 + len(pre_a) + 1 + 3 == 128 and 11 + len(pre_a) + 4 == 128
    for name, bs in (("XFE_K0", k0), ("XFE_PRE_A", pre_a), ("XFE_PRE_B", pre_b), ("XFE_PRE_B3", pre_b[3:]), ("XFE_PRE_B4", pre_b[4:]), ("XFE_POST1", post1), ("XFE_MID_A", mid_a),
                     ("XFE_POST2", post2), ("XFE_MID_B", mid_b), ("XFE_POST3", post3), ("BFE_PRE", bfe_pre), ("BFE_POST", bfe_post)):
        f.write(emit(name, bs))
    f.write("\n}  // namespace tpl\n}  // namespace bfs\n")
print("wrote", OUT)
print("segment lengths:", {n: len(b) for n, b in (("K0", k0), ("PRE_A", pre_a), ("PRE_B", pre_b), ("POST1", post1), ("MID_A", mid_a),
                                                   ("POST2", post2), ("MID_B", mid_b), ("POST3", post3), ("BFE_PRE", bfe_pre), ("BFE_POST", bfe_post))})
