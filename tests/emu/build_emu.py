"""Builds tests/emu/libbfs_emu.so: the product's DEVICE code compiled for the host (g++), test infrastructure for the CPU
suite.  Every .cpp of this directory goes in, and the library is rebuilt whenever the content of a source or of a csrc
header differs from what it was made of (content hash kept beside the .so; file times are not consulted)."""
import os
import sys

EMU = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(EMU))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build_emulation(force=False):
    from stark_brainfuck_amd.build import CSRC, hashed_build
    srcs = sorted(os.path.join(EMU, f) for f in os.listdir(EMU) if f.endswith(".cpp"))
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    out = os.path.join(EMU, "libbfs_emu.so")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-DBFS_CHECK_CANONICAL", "-shared", "-fPIC", "-o", out + ".tmp"] + srcs
    return hashed_build(out, srcs + headers, cmd, force=force, what="libbfs_emu.so")


if __name__ == "__main__":
    print("built", build_emulation(force="--force" in sys.argv))
