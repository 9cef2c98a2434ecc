"""Build a variant of the library for an A/B run: the named translation units recompiled with extra compiler flags, everything else taken from the
objects of the regular build (gnss-sdr_amd/_build).   python profiles/ab/build_variant.py <tag> <unit>[,<unit>...] [flags ...]
-> build/variants/lib_<tag>.so   (use with GSH_LIB_PATH=...)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gnss_sdr_amd  # noqa: E402
from gnss_sdr_amd import build as B  # noqa: E402

tag, units, extra = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
B.build_library()
out = os.path.join(ROOT, "build", "variants")
os.makedirs(out, exist_ok=True)
flags = [f for f in B.HIPCC_FLAGS if f != "-shared"]
inc = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "gnss-sdr_amd", "csrc")]
objs = []
for src in B.sources():
    name = os.path.basename(src)[:-4]
    obj = os.path.join(ROOT, "gnss-sdr_amd", "_build", name + ".o")
    if name in units:
        obj = os.path.join(out, f"{name}_{tag}.o")
        subprocess.run(["hipcc"] + flags + inc + extra + ["-c", src, "-o", obj], check=True)
    objs.append(obj)
lib = os.path.join(out, f"lib_{tag}.so")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
print(lib)
