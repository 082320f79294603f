"""Builds libmaster_hip.so (hand-written gfx950 kernels + C ABI) in-tree with hipcc.

    python -m mastermetastyletransfer_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to the GPU box with
the working tree; nothing is JIT-compiled at run time.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
INCLUDE = os.path.abspath(os.path.join(CSRC, "..", "..", "include"))
LIB = os.path.join(CSRC, "libmaster_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", f"-I{INCLUDE}"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    srcs = _sources()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "master_hip.h")]
    objs, jobs = [], []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(CSRC, s[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([HIPCC, *FLAGS, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
