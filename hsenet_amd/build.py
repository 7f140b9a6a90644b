"""Build libhsenet_hip.so (gfx950 only) in-tree with hipcc.  `python -m hsenet_amd.build [--force]`.

The shared object lands in hsenet_amd/lib/ (git-ignored, but it travels to the GPU box with the tree)."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libhsenet_hip.so")
SOURCES = ["gemm.hip", "attn.hip", "norm.hip", "small.hip", "vit.hip", "path.hip", "prof.hip", "prep.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# attn.hip: no SLP vectorizer.  It packs the independent f32 adds/multiplies of two softmax streams into v_pk_*_f32, which
# take more issue time beside MFMAs than the scalar ops they replace (MI355X_MICROARCH.md, cycle constants).
EXTRA = {"attn.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"],
         # prep.hip: the reference applies scale and shift (and torch its interpolation taps) as separate roundings
         "prep.hip": ["-ffp-contract=off"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: hsenet_amd needs the ROCm toolchain to build its gfx950 kernels")
    return exe


def _hdrs():
    return [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "driver.h"), os.path.join(CSRC, "attn_util.h"),
            os.path.join(os.path.dirname(HERE), "include", "hsenet_hip.h")]


def source_id(src: str = None) -> str:
    """content hash (sha256, 16 hex digits) of one source + the headers + its flags, or — src None — of everything the library is built
    from.  Staleness is decided by CONTENT, not by modification times (a checkout or a copy to another box resets those), and the
    library-wide id is compiled into the shared object (hs_build_id) so that a loaded .so can be matched to the tree it came from."""
    h = hashlib.sha256()
    names = [src] if src else SOURCES
    for n in names:
        h.update(n.encode())
        h.update(open(os.path.join(CSRC, n), "rb").read())
        h.update(" ".join(FLAGS + EXTRA.get(n, [])).encode())
    for f in _hdrs():
        h.update(open(f, "rb").read())
    h.update(_toolchain_id().encode())
    return h.hexdigest()[:16]


def _toolchain_id() -> str:
    """the compiler's identity without running it (this is hashed on every library load): the ROCm release stamp of the toolchain
    hipcc belongs to; an object built by another ROCm release is stale."""
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.realpath(exe)))
    for cand in (os.path.join(root, ".info", "version"), "/opt/rocm/.info/version"):
        v = _read(cand)
        if v:
            return "rocm " + v
    return "rocm ?"


def _read(path: str) -> str:
    try:
        return open(path).read().strip()
    except OSError:
        return ""


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    objs, jobs, stamps = [], [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        sid, tag = source_id(s), obj + ".id"
        if force or not os.path.exists(obj) or _read(tag) != sid:
            jobs.append([hipcc, *FLAGS, *EXTRA.get(s, []), "-c", src, "-o", obj])
            stamps.append((tag, sid))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
        for tag, sid in stamps:
            open(tag, "w").write(sid)
    # the build id: always regenerated and compiled (a few hundred ms) — every build() call runs hipcc at least once
    lid = source_id()
    idc, ido = os.path.join(LIBDIR, "build_id.cpp"), os.path.join(LIBDIR, "build_id.o")
    open(idc, "w").write(f'extern "C" const char* hs_build_id(void) {{ return "{lid}"; }}\n')
    run([hipcc, "-O1", "-fPIC", "-c", idc, "-o", ido])
    if jobs or force or not os.path.exists(LIB) or _read(LIB + ".id") != lid:
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, ido])
        open(LIB + ".id", "w").write(lid)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
