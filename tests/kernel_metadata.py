"""Reads the AMDGPU kernel metadata (registers, scratch, LDS) out of a HIP shared library: the clang offload bundles in .hip_fatbin, the gfx950 code object of
each, its NT_AMDGPU_METADATA note (msgpack).  Test helper; no ROCm tool needed."""
from __future__ import annotations

import struct

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob: bytes, arch: str = "gfx950"):
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if arch in triple and size > 0:
                yield blob[pos + off:pos + off + size]
        pos += len(MAGIC)


def _notes(elf: bytes):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        if sh_type != 7:  # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            yield name, ntype, desc


def kernels(lib_path: str, arch: str = "gfx950") -> dict[str, dict]:
    """kernel name (mangled) -> metadata map ('.private_segment_fixed_size', '.vgpr_count', '.sgpr_count', '.group_segment_fixed_size', ...)"""
    blob = open(lib_path, "rb").read()
    out = {}
    for co in _code_objects(blob, arch):
        for name, ntype, desc in _notes(co):
            if name == b"AMDGPU" and ntype == 32:
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    out[k[".name"]] = k
    return out
