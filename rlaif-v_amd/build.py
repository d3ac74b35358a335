"""Builds librlaifv_hip.so (the C-ABI kernel library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "librlaifv_hip.so")
SOURCES = ["gemm.hip", "elementwise.hip", "attention.hip"]
HEADERS = [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "gemm.hpp"), os.path.join(CSRC, "attn_agpr.inc"), os.path.join(CSRC, "attn_fwd3.inc"), os.path.join(CSRC, "attn_dkv5_body.inc"), os.path.join(CSRC, "attn_dkv5_skip.inc"), os.path.join(CSRC, "attn_dkv5_prefetch.inc"), os.path.join(INCLUDE, "rlaifv_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", f"-I{INCLUDE}", f"-I{CSRC}"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_extension(force: bool = False, verbose: bool = True, defines=(), tag: str = "") -> str:
    """The shipped library: ``build_extension()``.  ``defines`` / ``tag`` build a SEPARATE experiment library
    (librlaifv_hip<tag>.so, selected with RV_HIP_LIB for A/B runs) - e.g. defines=("RV_GEMM_EXPERIMENTS",) compiles the
    ablation / schedule-experiment GEMM variants, which the shipped library does not contain."""
    bdir = os.path.join(HERE, "build" + tag)
    os.makedirs(bdir, exist_ok=True)
    hipcc = _hipcc()
    lib = LIB if not tag else LIB.replace(".so", f"{tag}.so")
    objs, jobs = [], []
    only = tuple(x for x in os.environ.get("RV_BUILD_ONLY", "").split(",") if x)     # experiment builds: sources the defines touch
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(bdir, src.replace(".hip", ".o"))
        if tag and only and src not in only:
            o = os.path.join(HERE, "build", src.replace(".hip", ".o"))       # unchanged source: the shipped build's object
            if not os.path.exists(o):
                raise RuntimeError("build the shipped library first (its objects are reused by experiment builds)")
            objs.append(o)
            continue
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            jobs.append([hipcc] + FLAGS + [f"-D{d}" for d in defines] + ["-c", s, "-o", o])
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            for cmd, res in zip(jobs, ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs)):
                if res.returncode != 0:
                    raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + res.stderr[-4000:])
                if verbose:
                    print("[build]", os.path.basename(cmd[-1]), file=sys.stderr)
    if force or jobs or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("link failed:\n" + res.stderr[-4000:])
        if verbose:
            print("[build] linked", lib, file=sys.stderr)
    return lib


if __name__ == "__main__":
    if "--experiments" in sys.argv:
        print(build_extension(force="--force" in sys.argv, defines=("RV_GEMM_EXPERIMENTS",), tag="_exp"))
    else:
        print(build_extension(force="--force" in sys.argv))
