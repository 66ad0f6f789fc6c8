"""Build libcfgpp_hip.so (gfx950) in-tree with hipcc.  No JIT cache, no torch
extension machinery: plain ``hipcc -c`` per source + one link, so the ``.so``
travels with the repo snapshot to the GPU box."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libcfgpp_hip.so")
OBJ = os.path.join(CSRC, "_obj")

SOURCES = [
    # (file, extra flags)
    ("errors.cpp", []),
    ("step_kernels.hip", ["-ffp-contract=off"]),     # bit-exact sampler arithmetic: no fma contraction
    ("norm_kernels.hip", []),
    ("small_kernels.hip", []),
    ("igemm_kernel.hip", []),
    ("big4_kernel.hip", []),
    ("attn_kernel.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),   # scores are consumed by VALU: keep MFMA results in VGPRs
    ("unet.hip", []),
    ("vae.hip", []),
    ("text.hip", []),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


_INC = None


def _deps(src: str) -> set:
    """the quoted headers `src` includes, transitively (csrc/ and include/ only): a TU is rebuilt when one of ITS headers changed,
    not when any header did (igemm_kernel.hip alone takes minutes)"""
    import re
    global _INC
    if _INC is None:
        _INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)
    seen, todo = set(), [src]
    while todo:
        f = todo.pop()
        try:
            text = open(f, errors="replace").read()
        except OSError:
            continue
        for inc in _INC.findall(text):
            h = os.path.normpath(os.path.join(os.path.dirname(f), inc))
            if h not in seen and os.path.exists(h):
                seen.add(h)
                todo.append(h)
    return seen


def _newer(src: str, dst: str) -> bool:
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(d) > t for d in [src, *_deps(src)] if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for f, extra in SOURCES:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ, os.path.splitext(f)[0] + ".o")
        if force or _newer(src, obj):
            cmd = [hipcc] + COMMON + extra + ["-x", "hip", "-c", src, "-o", obj]
            jobs.append((f, cmd))

    def run(job):
        f, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        return f, r.returncode, r.stdout + r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for f, rc, log in ex.map(run, jobs):
                if verbose:
                    print(f"[cfgpp build] {f}: {'ok' if rc == 0 else 'FAILED'}", flush=True)
                if rc != 0:
                    raise RuntimeError(f"hipcc failed on {f}:\n{log}")
    objs = [os.path.join(OBJ, os.path.splitext(f)[0] + ".o") for f, _ in SOURCES]
    if force or jobs or not os.path.exists(OUT):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        if verbose:
            print(f"[cfgpp build] linked {OUT}", flush=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
