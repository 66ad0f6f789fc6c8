"""Build libcfgpp_hip.so (gfx950) in-tree with hipcc.  No JIT cache, no torch
extension machinery: plain ``hipcc -c`` per source + one link, so the ``.so``
travels with the repo snapshot to the GPU box."""
from __future__ import annotations

import hashlib
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
    ("big4p_kernel.hip", []),
    ("attn_kernel.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]),   # scores are consumed by VALU: keep MFMA results in VGPRs
    ("unet.hip", []),
    ("vae.hip", []),
    ("text.hip", []),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# `python -m cfgpp_amd.build --asan`: a SECOND library, libcfgpp_hip_asan.so, whose HOST code is AddressSanitizer-instrumented
# (device code is not: -fno-gpu-sanitize; GPU ASan / xnack+ objects are not available on the MI355X pool).  Development only:
# load it with CFGPP_LIB=.../libcfgpp_hip_asan.so and LD_PRELOAD=$(clang -print-file-name=libclang_rt.asan-x86_64.so)
# (scripts/r06_runs/asan_repro.sh).  The shipped library is never built this way.
ASAN_FLAGS = ["-fsanitize=address", "-fno-gpu-sanitize", "-g", "-fno-omit-frame-pointer"]


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


def _digest(src: str, flags) -> str:
    """content digest of one translation unit: the source, every header it reaches, its flags, the compiler.  Objects are NAMED by
    it (csrc/_obj/<stem>.<digest>.o), so an object built from other text or other flags can never be linked - mtimes play no part."""
    h = hashlib.sha256()
    h.update((" ".join(flags) + "\0" + _hipcc_version()).encode())
    for f in [src, *sorted(_deps(src))]:
        h.update(os.path.relpath(f, HERE).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


_HIPCC_VERSION = None


def _hipcc_version() -> str:
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        try:
            out = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout
            _HIPCC_VERSION = " ".join(ln.strip() for ln in out.splitlines() if "version" in ln.lower())[:200]
        except OSError:
            _HIPCC_VERSION = "unknown"
    return _HIPCC_VERSION


def source_digest() -> str:
    """digest of everything libcfgpp_hip.so is built from; `cfgpp_build_id()` of a library built from this tree starts with it"""
    h = hashlib.sha256()
    for f, extra in SOURCES:
        h.update(_digest(os.path.join(CSRC, f), COMMON + extra).encode())
    return h.hexdigest()[:16]


def _git_head() -> str:
    try:
        root = os.path.dirname(HERE)
        head = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
        dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "cfgpp_amd/csrc", "include"], capture_output=True,
                               text=True).stdout.strip()
        return (head or "nogit") + ("+local" if dirty else "")
    except OSError:
        return "nogit"


def library_build_id(path: str = OUT) -> str:
    """the `cfgpp_build_id()` string of a built library, read from the file (no dlopen): '' when absent"""
    try:
        data = open(path, "rb").read()
    except OSError:
        return ""
    i = data.find(b"cfgpp-build:")
    return data[i:data.index(b"\0", i)].decode() if i >= 0 else ""


def build_asan(verbose: bool = True) -> str:
    """host-ASan twin of the library (see ASAN_FLAGS): own object directory, own output name"""
    out = os.path.join(HERE, "libcfgpp_hip_asan.so")
    obj_dir = os.path.join(CSRC, "_obj_asan")
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    jobs, objs = [], []
    for f, extra in SOURCES:
        src = os.path.join(CSRC, f)
        flags = COMMON + ASAN_FLAGS + extra
        obj = os.path.join(obj_dir, f"{os.path.splitext(f)[0]}.{_digest(src, flags)}.o")
        objs.append(obj)
        if not os.path.exists(obj):
            jobs.append((f, [hipcc] + flags + ["-x", "hip", "-c", src, "-o", obj]))
    with ThreadPoolExecutor(max_workers=8) as ex:
        for f, r in zip([j[0] for j in jobs], ex.map(lambda j: subprocess.run(j[1], capture_output=True, text=True), jobs)):
            if verbose:
                print(f"[cfgpp build asan] {f}: {'ok' if r.returncode == 0 else 'FAILED'}", flush=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed on {f}:\n{r.stdout}{r.stderr}")
    bid_src = os.path.join(obj_dir, "build_id.cpp")
    with open(bid_src, "w") as fh:
        fh.write('extern "C" const char* cfgpp_build_id(void) { return "cfgpp-build:%s:%s+asan"; }\n' % (source_digest(), _git_head()))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan", "-o", out] + objs + [bid_src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("asan link failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(f"[cfgpp build asan] linked {out}", flush=True)
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    jobs, objs = [], []
    for f, extra in SOURCES:
        src = os.path.join(CSRC, f)
        stem = os.path.splitext(f)[0]
        obj = os.path.join(OBJ, f"{stem}.{_digest(src, COMMON + extra)}.o")
        objs.append(obj)
        if force or not os.path.exists(obj):
            jobs.append((f, [hipcc] + COMMON + extra + ["-x", "hip", "-c", src, "-o", obj]))

    def run(job):
        f, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0 and os.path.exists(cmd[-1]):
            os.remove(cmd[-1])
        return f, r.returncode, r.stdout + r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for f, rc, log in ex.map(run, jobs):
                if verbose:
                    print(f"[cfgpp build] {f}: {'ok' if rc == 0 else 'FAILED'}", flush=True)
                if rc != 0:
                    raise RuntimeError(f"hipcc failed on {f}:\n{log}")
    keep = set(objs)
    for old in os.listdir(OBJ):                       # objects of earlier source text
        if old.endswith(".o") and os.path.join(OBJ, old) not in keep:
            os.remove(os.path.join(OBJ, old))
    want = source_digest()
    have = library_build_id()
    if force or jobs or not have.startswith(f"cfgpp-build:{want}:"):
        # the provenance string: <digest of the sources this library is built from>:<git HEAD when it was linked>[+local]
        bid_src, bid_obj = os.path.join(OBJ, "build_id.cpp"), os.path.join(OBJ, "build_id.o")
        with open(bid_src, "w") as fh:
            fh.write('extern "C" const char* cfgpp_build_id(void) { return "cfgpp-build:%s:%s"; }\n' % (want, _git_head()))
        r = subprocess.run([hipcc, "-O2", "-fPIC", "-c", bid_src, "-o", bid_obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("build_id.cpp failed:\n" + r.stdout + r.stderr)
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT + ".tmp"] + objs + [bid_obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        os.replace(OUT + ".tmp", OUT)
        os.remove(bid_obj)
        if verbose:
            print(f"[cfgpp build] linked {OUT} ({library_build_id()})", flush=True)
    return OUT


if __name__ == "__main__":
    if "--asan" in sys.argv:
        print(build_asan())
    else:
        build(force="--force" in sys.argv)
        print(library_build_id())
