"""In-tree build of libgnss_sdr_hip.so (hipcc cross-compiles gfx950 without a GPU)."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB = os.path.join(_HERE, "libgnss_sdr_hip.so")

# -ffp-contract=off is load-bearing: the code-phase arithmetic must round once per operation to be
# bit-exact with the reference's float32 chip selection (see csrc/multicorrelator.hip).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function"]


# translation units that #include another .hip (the same kernels at another work-group size)
INCLUDED_SOURCES = {"multicorrelator_t128.hip": ["multicorrelator.hip"]}


def sources() -> list[str]:
    return sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")))


def _headers() -> list[str]:
    return glob.glob(os.path.join(_HERE, "csrc", "*.h")) + glob.glob(os.path.join(_ROOT, "include", "*.h"))


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in sources() + _headers())


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 (one object per file, stale ones only, in parallel) and link them
    into one shared library.  Returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.exists(LIB):
            return LIB  # GPU box without a compiler: use the prebuilt library that travelled with the tree
        raise RuntimeError("hipcc not found and no prebuilt libgnss_sdr_hip.so present")
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(_HERE, "_build")
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    inc = ["-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_HERE, "csrc")]
    objs, jobs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        src_t = max([os.path.getmtime(src)] + [os.path.getmtime(os.path.join(_HERE, "csrc", d)) for d in INCLUDED_SOURCES.get(os.path.basename(src), [])])
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(src_t, hdr_t):
            jobs.append([hipcc] + flags + inc + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs)
    os.replace(LIB + ".tmp", LIB)
    return LIB
