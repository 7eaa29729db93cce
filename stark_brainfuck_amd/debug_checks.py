"""The prover's DEBUG degree checks (round-5 verdict, next #6).

With `DEBUG` in the environment the reference interpolates every quotient codeword right where it is made and asserts that it is a
polynomial of less than maximal degree --

    Table.boundary_quotients    /root/reference/code/table.py:170-176    assert degree < fri_domain.length - 1
    Table.transition_quotients  table.py:219-234                         assert(False) behind a dump when degree >= length - 1, then
                                                                         "quotient polynomial has maximal degree in table <class>"
    Table.terminal_quotients    table.py:264-284                         assert(False) behind a dump when degree >= length - 1

-- and, while it assembles the terms of the non-linear combination, every shifted base / extension codeword and every quotient before
and after its shift (brainfuck_stark.py:251-290):

    shifted base / extension codeword i        assert degree <= max_degree
    quotient, unshifted                        "for unshifted quotient polynomial {i}, interpolated degree is {d} but > degree bound i = {bound}"
    quotient, shifted                          "for (shifted) quotient polynomial {i}, interpolated degree is {d} but > max_degree = {max_degree}"

(`i` in the two messages is the reference's stale loop variable: the index of the last extension codeword, whatever the quotient.)

Here the same checks run on the device's codewords: one inverse transform per limb plane (bfs_gl_ntt with the inverse root -- the coset
shift only multiplies coefficient k by offset^-k, which does not change which coefficients are zero, so it is left out), then the
highest non-zero coefficient on the host.  A shifted codeword x^s f(x) on the coset is f's coefficient vector rotated by s places
(x^n = offset^n there), so its degree is max over the non-zero k of (k + s) mod n: no second transform.  The checks need the quotient
codewords written out, so a DEBUG proof takes the Python-stage path with `keep_intermediates` (same field elements, same proof bytes).
Enabled by `DEBUG` (the reference's switch) or `BFS_DEBUG=1`.  An AssertionError raised here carries `.where = (function, table class
name, constraint index)` for the table-level checks and `.where = ("prove", kind, index)` for the others."""
import ctypes
import os

import numpy as np

from . import _lib
from .arrays import raw_ntt
from .device import DeviceBuffer, current_stream, synchronize


def enabled():
    return os.environ.get("DEBUG") is not None or os.environ.get("BFS_DEBUG") == "1"


def _fail(message, where):
    e = AssertionError(message)
    e.where = where
    return e


def support(ptr, planes, count, n, omega):
    """(count, n) boolean array: which coefficients of the interpolants of `count` codewords (each `planes` limb planes of n words at
    `ptr`, codeword-major) are non-zero"""
    lib = _lib.load()
    log_n = n.bit_length() - 1
    out = DeviceBuffer(count * planes * n)
    raw_ntt(ptr, n, n, out.ptr, n, log_n, count * planes, lib.bfs_gl_inv(omega), 1, 1)
    host = out.to_numpy(count * planes * n).reshape(count, planes, n)
    synchronize(current_stream())
    return (host != 0).any(axis=1)


def degrees(nonzero, shift=0):
    """degree of each interpolant (rows of `nonzero`), of x^shift times it when shift != 0; -1 for the zero polynomial"""
    n = nonzero.shape[1]
    out = []
    for row in nonzero:
        k = np.nonzero(row)[0]
        out.append(int(((k + shift) % n).max()) if len(k) else -1)
    return out


def check_table_quotients(table, buffer, n, omega):
    """table.py:170-176, 219-234, 264-284 on the table's quotient buffer (boundary / transition / terminal order); returns the
    support array for the prover-level checks"""
    lib = _lib.load()
    counts = (ctypes.c_int * 3)()
    _lib.check(lib.bfs_air_counts(table.table_index, counts))
    total = counts[0] + counts[1] + counts[2]
    nonzero = support(buffer.ptr, 3, total, n, omega)
    deg = degrees(nonzero)
    name = type(table).__name__
    at = 0
    for kind, function, message in ((0, "boundary_quotients", ""), (1, "transition_quotients", ""), (2, "terminal_quotients", "")):
        for l in range(counts[kind]):
            if deg[at] >= n - 1:
                raise _fail(message or "%s: quotient %d of %s is not a polynomial of less than maximal degree (interpolated degree %d, domain length %d)"
                            % (function, l, name, deg[at], n), (function, name, l))
            at += 1
    return nonzero


def check_terms(stark, n, omega, base_degree_bounds, extension_degree_bounds, quotient_supports, quotient_degree_bounds):
    """brainfuck_stark.py:251-290: the terms of the non-linear combination, in the reference's order"""
    max_degree = stark.max_degree
    i = -1
    at = 0
    for table in stark.tables:
        if table.base_width == 0 or table.base_codewords is None:
            continue
        nz = support(table.base_codewords.ptr, 1, table.base_width, n, omega)
        for c in range(table.base_width):
            i = at + c
            d = degrees(nz[c:c + 1], max_degree - base_degree_bounds[i])[0]
            if not d <= max_degree:
                raise _fail("shifted base codeword %d: interpolated degree %d > max_degree = %d" % (i, d, max_degree), ("prove", "base", i))
        at += table.base_width
    at = 0
    for table in stark.tables:
        width = table.full_width - table.base_width
        if width == 0 or table.ext_codewords is None:
            continue
        nz = support(table.ext_codewords.ptr, 3, width, n, omega)
        for c in range(width):
            i = at + c
            d = degrees(nz[c:c + 1], max_degree - extension_degree_bounds[i])[0]
            if not d <= max_degree:
                raise _fail("shifted extension codeword %d: interpolated degree %d > max_degree = %d" % (i, d, max_degree), ("prove", "extension", i))
        at += width
    # the quotients; `i` stays what the extension loop left it at, as in the reference's messages
    q = 0
    for nz in quotient_supports:
        for row in range(nz.shape[0]):
            bound = quotient_degree_bounds[q]
            d = degrees(nz[row:row + 1])[0]
            if not (d == -1 or d <= bound):
                raise _fail("for unshifted quotient polynomial %d, interpolated degree is %d but > degree bound i = %d" % (i, d, bound), ("prove", "quotient", q))
            d = degrees(nz[row:row + 1], max_degree - bound)[0]
            if not (d == -1 or d <= max_degree):
                raise _fail("for (shifted) quotient polynomial %d, interpolated degree is %d but > max_degree = %d" % (i, d, max_degree), ("prove", "shifted quotient", q))
            q += 1
    assert q == len(quotient_degree_bounds)
