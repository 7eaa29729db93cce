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


OBJ = os.path.join(HERE, "_build")


def _stale(obj, dep, key):
    """an object is rebuilt when it, its dependency file or the flag record is missing, or any file it included is newer"""
    if not (os.path.exists(obj) and os.path.exists(dep) and os.path.exists(obj + ".flags")):
        return True
    if open(obj + ".flags").read() != key:
        return True
    t = os.path.getmtime(obj)
    text = open(dep).read().replace("\\\n", " ")
    files = text.split(":", 1)[1].split() if ":" in text else []
    return any((not os.path.exists(f)) or os.path.getmtime(f) > t for f in files)


def build_library(force=False, verbose=False, extra_flags=()):
    """one object per translation unit (compiled in parallel, rebuilt only when a file it includes changed), then one link"""
    if not force and up_to_date():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    compile_flags = [f for f in FLAGS if f != "-shared"] + list(extra_flags)
    key = " ".join([hipcc] + compile_flags)
    jobs, objects = [], []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src) + ".o")
        dep = obj + ".d"
        objects.append(obj)
        if force or _stale(obj, dep, key):
            jobs.append((src, obj, dep))

    def compile_one(job):
        src, obj, dep = job
        cmd = [hipcc] + compile_flags + ["-c", "-MD", "-MF", dep, "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode == 0:
            open(obj + ".flags", "w").write(key)
        return src, res
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        results = list(pool.map(compile_one, jobs))
    failed = [(src, res) for src, res in results if res.returncode != 0]
    for src, res in results:
        if res.stdout and (verbose or res.returncode != 0):
            sys.stderr.write(res.stdout)
    if failed:
        raise RuntimeError("hipcc failed building " + ", ".join(os.path.basename(src) for src, _ in failed))
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB + ".tmp"] + objects
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout)
        raise RuntimeError("hipcc failed linking libbfstark_hip.so")
    os.replace(LIB + ".tmp", LIB)
    return LIB


def build_fastlist(force=False):
    """the CPython helper for list <-> buffer conversions (cpyext/fastlist.c), compiled in-tree with gcc"""
    import sysconfig
    src = os.path.join(HERE, "cpyext", "fastlist.c")
    out = os.path.join(HERE, "_fastlist" + sysconfig.get_config_var("EXT_SUFFIX"))
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cmd = [os.environ.get("CC", "gcc"), "-O2", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"], "-o", out + ".tmp", src]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout)
        raise RuntimeError("gcc failed building _fastlist")
    os.replace(out + ".tmp", out)
    return out


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose="--verbose" in sys.argv or True,
                  extra_flags=["-Rpass-analysis=kernel-resource-usage"] if "--resources" in sys.argv else [])
    print("built", LIB)
    print("built", build_fastlist(force="--force" in sys.argv))
