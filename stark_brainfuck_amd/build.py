"""Builds libbfstark_hip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m stark_brainfuck_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.  The .so stays inside the
package directory (git-ignored, but it travels to the GPU box with the repo snapshot).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbfstark_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function",
         "-ffp-contract=off",
         "-Xarch_host", "-mbmi", "-Xarch_host", "-mbmi2"]      # host side: andn / rorx for Keccak (keccak.hpp), any x86-64-v3 CPU


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps():
    inc = os.path.join(os.path.dirname(HERE), "include")
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    files += [os.path.join(inc, f) for f in os.listdir(inc)]
    return files + [os.path.abspath(__file__)]


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(f) <= t for f in _deps())


def build_library(force=False, verbose=False, extra_flags=()):
    if not force and up_to_date():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + list(extra_flags) + ["-o", LIB + ".tmp"] + sources()
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout)
        raise RuntimeError("hipcc failed building libbfstark_hip.so")
    if verbose and res.stdout:
        print(res.stdout)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose="--verbose" in sys.argv or True,
                  extra_flags=["-Rpass-analysis=kernel-resource-usage"] if "--resources" in sys.argv else [])
    print("built", LIB)
