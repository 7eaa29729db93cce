"""Is the slow state of the out-of-place NTT pass address translation?  (development tool, round 4)
Runs the 8 x 2^24 step over every ordered pair of K separately allocated 1 GiB buffers, S steps per pair, in a fixed order; under
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d DIR -o t -- python tools/tlb_pairs.py run K S
and then   python tools/tlb_pairs.py report DIR K S   prints, per pair and pass, the median dispatch duration next to the counters."""
import csv, glob, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def order(K):
    return [(i, j) for i in range(K) for j in range(K) if i != j]


if sys.argv[1] == "run":
    import numpy as np
    from stark_brainfuck_amd import _lib
    from stark_brainfuck_amd.device import DeviceBuffer, synchronize
    K, S = int(sys.argv[2]), int(sys.argv[3])
    lib = _lib.load()
    n, cols, logn = 1 << 24, 8, 24
    bufs = [DeviceBuffer(n * cols) for _ in range(K)]
    v = (np.arange(n * cols, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) % np.uint64(0xFFFFFFFF00000001)
    for b in bufs:
        _lib.check(lib.bfs_memcpy_h2d(b.ptr, v.ctypes.data, v.nbytes, 0))
    synchronize(0)
    w = lib.bfs_gl_primitive_root(logn)
    for i, j in order(K):
        for _ in range(S):
            _lib.check(lib.bfs_gl_ntt(bufs[i].ptr, n, n, bufs[j].ptr, n, logn, cols, w, 1, 1, 0))
        synchronize(0)
    print("addresses", [hex(b.ptr) for b in bufs])
else:
    d, K, S = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    trace = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(trace)) if "ntt_tile_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    counters = {}
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if cc:
        for r in csv.DictReader(open(cc[0])):
            counters.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    names = sorted({c for v in counters.values() for c in v})
    assert len(rows) == len(order(K)) * S * 3, (len(rows), K, S)
    print("pair   pass  median_us  " + "  ".join(names))
    at = 0
    for i, j in order(K):
        chunk = rows[at:at + 3 * S]
        at += 3 * S
        for p in range(3):
            sel = chunk[p::3][1:]                 # drop the first step of a pair
            dur = statistics.median(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel) / 1e3
            vals = [statistics.median(counters.get(r["Dispatch_Id"], {}).get(c, float("nan")) for r in sel) for c in names]
            print("%d->%d   %d   %8.1f   %s" % (i, j, p, dur, "  ".join("%.4g" % x for x in vals)))
