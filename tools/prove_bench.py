"""BrainfuckStark.prove as bench.py times it (bench.bench_stark), on its own: Hello World (FRI domain 2^17) and the 37 254-cycle
nested-loop program (2^22).  Used alone for A/B work on the prover kernels and under rocprofv3 for their statistics.
    python tools/prove_bench.py [--small] [--large] [--reps N]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

LARGE = "+" * 64 + "[>" + "+" * 64 + "[>++++<-]<-]+++."

if __name__ == "__main__":
    small = "--small" in sys.argv or "--large" not in sys.argv
    large = "--large" in sys.argv or "--small" not in sys.argv
    out = {}
    if small:
        out["stark_prove"] = bench.bench_stark()
    if large:
        out["stark_prove_2p22"] = bench.bench_stark(LARGE, "nested loops, 37 254 cycles")
    for k, v in out.items():
        print(k, "%.3f ms" % v["ms"], "verified", v["verified"], json.dumps(v["breakdown_ms"]))
