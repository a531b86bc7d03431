#!/usr/bin/env python3
"""Register / LDS / spill figures of the kernels inside a HIP shared library (the code object's AMDGPU metadata).

    python tools/kernel_meta.py gffx_amd/lib/libgffx_hip.so [substring of the demangled name ...]
"""
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def code_objects(lib, tmp):
    """The gfx950 code objects of every translation unit: .hip_fatbin is a run of clang offload bundles."""
    sec = tmp + "/fatbin"
    subprocess.check_call([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, sec])
    data = open(sec, "rb").read()
    magic, out, at = b"__CLANG_OFFLOAD_BUNDLE__", [], 0
    while True:
        at = data.find(magic, at)
        if at < 0:
            break
        n = struct.unpack_from("<Q", data, at + 24)[0]
        q = at + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                path = "%s/co%d.o" % (tmp, len(out))
                open(path, "wb").write(data[at + off:at + off + size])
                out.append(path)
        at += 24
    return out


def main(lib, pats):
    with tempfile.TemporaryDirectory() as tmp:
        notes = "".join(subprocess.check_output([LLVM + "llvm-readelf", "--notes", co], text=True) for co in code_objects(lib, tmp))
    rows = []
    for blk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        rows.append((name, g("vgpr_count"), g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"),
                     g("group_segment_fixed_size")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    rows = [(n,) + r[1:] for n, r in zip(names, rows)]
    for r in rows:
        short = re.sub(r"\(.*", "", r[0])
        if not pats or any(p in short for p in pats):
            print("%-90s vgpr %-4s sgpr %-4s sgpr_spill %-4s vgpr_spill %-3s scratch %-5s lds %s" % ((short[:90],) + r[1:]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
