"""Compile csrc/*.hip for gfx950 into the in-tree C-ABI library libsodt_hip.so.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is
git-ignored but travels to the GPU box with the snapshot.  Usage:
    python small-object-detection-transformers_amd/build.py [--force]
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libsodt_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=fast", "-Wno-unused-value"]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdr = _deps_mtime()
    jobs = []
    objs = []
    for src in _sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return src
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for s in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[build] compiled {os.path.basename(s)}", flush=True)
    if jobs or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):      # (objects built by hand count too)
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
