"""One-off parity check of the large plans against the oracle: 2^21..2^26 (three- and four-pass plans), plain / inverse-scaled /
coset with zero padding.  ~2 minutes of CPU time for the oracle.   python tools/check_large_ntt.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import ref_oracle as o
from stark_brainfuck_amd import _lib
from stark_brainfuck_amd.device import DeviceBuffer, synchronize

lib = _lib.load()
for logn in (21, 22, 23, 25, 26):
    n = 1 << logn
    w = o.primitive_nth_root(n)
    v = o.felt_array(0x5EED + logn, 0, n)
    din, dout = DeviceBuffer.from_numpy(v), DeviceBuffer(n)
    t0 = time.time()
    _lib.check(lib.bfs_gl_ntt(din.ptr, n, n, dout.ptr, n, logn, 1, w, 1, 1, 0))
    synchronize(0)
    fwd = dout.to_numpy()
    assert (fwd == o.ntt(w, v)).all(), "forward 2^%d" % logn
    _lib.check(lib.bfs_gl_ntt(dout.ptr, n, n, din.ptr, n, logn, 1, o.inv(w), 1, o.inv(n), 0))
    synchronize(0)
    assert (din.to_numpy() == v).all(), "inverse 2^%d" % logn
    d = n // 4 + 3
    dc = DeviceBuffer.from_numpy(v[:d])
    _lib.check(lib.bfs_gl_ntt(dc.ptr, d, d, dout.ptr, n, logn, 1, w, 7, 1, 0))
    synchronize(0)
    assert (dout.to_numpy() == o.fast_coset_evaluate(v[:d], 7, w, n)).all(), "coset 2^%d" % logn
    print("2^%d ok (forward, inverse, coset with %d of %d coefficients), %.1f s" % (logn, d, n, time.time() - t0), flush=True)
