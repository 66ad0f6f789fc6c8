#!/usr/bin/env python
"""Per-kernel register / scratch / LDS figures of the gfx950 code objects inside libcfgpp_hip.so (or any object file):
walks the clang offload bundles (`__CLANG_OFFLOAD_BUNDLE__`), writes each gfx950 ELF to a temp file and reads the
kernel metadata notes with llvm-readelf.  Usage: python scripts/kernel_resources.py [file] [--all]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def bundles(blob):
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = 0
    while True:
        pos = blob.find(magic, pos)
        if pos < 0:
            return
        n = struct.unpack_from("<Q", blob, pos + len(magic))[0]
        q = pos + len(magic) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, q)
            ident = blob[q + 24:q + 24 + idlen].decode()
            q += 24 + idlen
            if "gfx950" in ident and size > 0:
                yield blob[pos + off:pos + off + size]
        pos += len(magic)


def kernels(path):
    blob = open(path, "rb").read()
    out = []
    for elf in bundles(blob):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(elf)
        txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        os.unlink(f.name)
        for blk in txt.split("- .agpr_count:")[1:]:
            g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]  # noqa: E731
            out.append(dict(name=g("name"), vgpr=g("vgpr_count"), agpr=blk.split()[0], sgpr=g("sgpr_count"), spill=g("vgpr_spill_count"),
                            scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size")))
    return out


if __name__ == "__main__":
    path = next((a for a in sys.argv[1:] if not a.startswith("--")), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cfgpp_amd", "libcfgpp_hip.so"))
    ks = kernels(path)
    bad = [k for k in ks if k["spill"] not in ("0", "?") or k["scratch"] not in ("0", "?")]
    print(f"{len(ks)} kernels, {len(bad)} with spills or scratch")
    for k in (ks if "--all" in sys.argv else bad):
        name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
        print(f"  vgpr {k['vgpr']:>4} agpr {k['agpr']:>4} sgpr {k['sgpr']:>4} spill {k['spill']:>4} scratch {k['scratch']:>6} lds {k['lds']:>7}  {name[:150]}")
