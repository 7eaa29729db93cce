"""Known answers for BASELINE config 5 (8 columns x 2^24-point forward NTT, SURVEY.md 8d: column c = felt(seed + c * 2^32, i)).
The reference would need ~1.8 h per column, so these digests come from the CPU oracle (oracle/gl_oracle.c, the
restatement of ntt.py:4-23 that tests/test_oracle_golden.py pins against the reference's own outputs up to 2^20), not from the
reference itself: the fixture says so.  ~80 s on one core.

    python tests/golden/gen_ntt24_oracle.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_oracle as o  # noqa: E402

SEED, LOGN, COLS = 0x5EED, 24, 8
n = 1 << LOGN
w = o.primitive_nth_root(n)
out = {"source": "oracle/gl_oracle.c (CPU restatement; the reference cannot run this size)", "seed": SEED, "log_n": LOGN, "root": w, "columns": []}
for c in range(COLS):
    v = o.felt_array(SEED + (c << 32), 0, n)
    f = o.ntt(w, v)
    out["columns"].append({"input_sha256": hashlib.sha256(np.ascontiguousarray(v, dtype="<u8").tobytes()).hexdigest(),
                           "output_sha256": hashlib.sha256(np.ascontiguousarray(f, dtype="<u8").tobytes()).hexdigest(),
                           "output_head": [int(x) for x in f[:3]]})
    print(c, out["columns"][-1]["output_sha256"], flush=True)
# the guard's edge-value columns (bench.py: edge_columns): operands next to 0, p and 2^32, all-(p - 1), alternating 1 / p - 1
from bench import edge_columns  # noqa: E402
out["edge_recipe"] = "bench.py: edge_columns(n, 8) -- edge_array(0xED6E + c, n), column 1 all p - 1, column 2 alternating 1 / p - 1"
out["edge_columns"] = []
for c, v in enumerate(edge_columns(n, COLS)):
    f = o.ntt(w, v)
    out["edge_columns"].append({"input_sha256": hashlib.sha256(np.ascontiguousarray(v, dtype="<u8").tobytes()).hexdigest(),
                                "output_sha256": hashlib.sha256(np.ascontiguousarray(f, dtype="<u8").tobytes()).hexdigest(),
                                "output_head": [int(x) for x in f[:3]]})
    print("edge", c, out["edge_columns"][-1]["output_sha256"], flush=True)
json.dump(out, open(os.path.join(HERE, "ntt24_oracle.json"), "w"), indent=1)
