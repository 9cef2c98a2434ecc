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


def sources() -> list[str]:
    return sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")))


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(_HERE, "csrc", "*.h")) + glob.glob(os.path.join(_ROOT, "include", "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP translation unit for gfx950 into one shared library.  Returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        if os.path.exists(LIB):
            return LIB  # GPU box without a compiler: use the prebuilt library that travelled with the tree
        raise RuntimeError("hipcc not found and no prebuilt libgnss_sdr_hip.so present")
    cmd = [hipcc] + HIPCC_FLAGS + ["-I" + os.path.join(_ROOT, "include"), "-I" + os.path.join(_HERE, "csrc"),
                                   "-o", LIB + ".tmp"] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB
