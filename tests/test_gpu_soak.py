"""Bounded slices of the randomised soaks (tools/soak.py, tools/soak_stark.py) inside the `-m gpu` suite.

Round 4's wrong lazy-sum rule in the NTT passed every fixed test of the suite and was caught by tools/soak_stark.py, which only the
builder ran.  These tests put a time-boxed run of both soaks in front of whoever runs the suite: random transform shapes (half of them
on operands next to 0, p and 2^32) against the CPU oracle, and random Brainfuck programs -- cells wrapping through p - 1 included --
proved on both prover paths, verified, and refused under an altered claim.  The seed changes from day to day so that repeated runs of
the suite widen what has been covered; a failure prints the seed and the offending configuration."""
import datetime
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
SEED = int(os.environ.get("BFS_SOAK_SEED", datetime.date.today().strftime("%Y%m%d")))
SECONDS = float(os.environ.get("BFS_SOAK_SECONDS", "25"))


def run_tool(args, timeout):
    res = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, "seed %d\n%s\n%s" % (SEED, res.stdout[-3000:], res.stderr[-3000:])
    return res.stdout


def test_soak_slice_transforms_and_trees_against_the_oracle():
    out = run_tool([os.path.join(ROOT, "tools", "soak.py"), str(SEED), str(SECONDS)], SECONDS * 4 + 240)
    assert "soak ok" in out
    stats = eval(out[out.index("{"):out.index("}") + 1])
    assert stats["ntt"] >= 50 and stats.get("edge", 0) >= 10 and stats["selftest"] == 1, out


def test_soak_slice_random_programs_proved_on_both_paths_and_verified():
    out = run_tool([os.path.join(ROOT, "tools", "soak_stark.py"), str(SECONDS), str(SEED)], SECONDS * 4 + 240)
    assert "both prover paths byte-identical" in out
    assert int(out.split()[0]) >= 100, out
