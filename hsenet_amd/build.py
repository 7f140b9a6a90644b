"""Build libhsenet_hip.so (gfx950 only) in-tree with hipcc.  `python -m hsenet_amd.build [--force]`.

The shared object lands in hsenet_amd/lib/ (git-ignored, but it travels to the GPU box with the tree)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libhsenet_hip.so")
SOURCES = ["gemm.hip", "attn.hip", "attn_bwd1.hip", "norm.hip", "small.hip", "vit.hip", "path.hip", "prof.hip", "prep.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# attn.hip: no SLP vectorizer.  It packs the independent f32 adds/multiplies of two softmax streams into v_pk_*_f32, which
# take more issue time beside MFMAs than the scalar ops they replace (MI355X_MICROARCH.md, cycle constants).
EXTRA = {"attn.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"],
         "attn_bwd1.hip": ["-fno-slp-vectorize"],
         # prep.hip: the reference applies scale and shift (and torch its interpolation taps) as separate roundings
         "prep.hip": ["-ffp-contract=off"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: hsenet_amd needs the ROCm toolchain to build its gfx950 kernels")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "driver.h"), os.path.join(CSRC, "attn_util.h"), os.path.join(os.path.dirname(HERE), "include", "hsenet_hip.h")]
    hipcc = _hipcc()
    objs, jobs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(LIBDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc, *FLAGS, *EXTRA.get(s, []), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
