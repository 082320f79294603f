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
# Per-file extras.  -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs instead of AGPRs.  By default hipcc gives every accumulator an
# AGPR, and each value the VALU touches afterwards (bias, ReLU, softmax, bf16 packing) is first copied out with v_accvgpr_read_b32:
# 345 of them per window in the fused attention block, 320 + 320 moves in conv_ring, all of the GEMM epilogues.  With the flag those
# kernels lose the moves (and 30-70 registers).  Measured per kernel on MI355X (profiles/r02_mfma_vgpr_form_ab.txt): attention block
# -3 %, ring convolutions -2...-10 %, the fp32 GEMMs / window attention of the parity mode and the train step -1...-4 %;  NOT for the
# two fused MLP kernels (they need the AGPR half of the register file: +5 % slower, or spills) nor linear_ts (neutral).
VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
EXTRA_FLAGS = {f: VGPR_FORM for f in (
    "attn_block_bf16.hip", "conv_ring_bf16.hip", "conv_tail_bf16.hip", "gemm_bf16.hip", "gemm_f32.hip", "gemm_f32_t.hip", "gemm_x3.hip",
    "window_attn_f32.hip", "window_attn_bwd_f32.hip", "window_attn_bf16.hip", "patch_embed_bf16.hip",
    # one wave per SIMD with 490 live registers (Wproj rows + bias fragments resident): with AGPR accumulators hipcc spills 53 of them
    "window_attn_proj_bf16.hip")}
# gemm_x3.hip MUST be built with VGPR accumulators (r04): with AGPR accumulators hipcc copies the 16 accumulator registers of the 128x32 tile
# out to VGPRs and back around the divergent tap arithmetic of the convolution gather in every k chunk, and the first five copies
# (a11..a15) came back wrong in a fraction of the workgroups of the 32 -> 32 convolution at 256^2 (nondeterministic output, found by
# profiles/tools/x3_determinism.py, now tests/test_gpu_x3.py::test_x3_kernels_repeatable_at_scale); the VGPR form has no such copies
# and is bit-repeatable.
# The two fp32 GEMM files are scheduled by the compiler (no hand-placed sched_barriers): its max-ILP strategy is worth 0.4-1.2 % on the
# fp32 mode and the train step; on the whole library it costs the bf16 step 5 % (the hand-scheduled kernels), so only here.
for _f in ("gemm_f32.hip", "gemm_f32_t.hip"):
    EXTRA_FLAGS[_f] = EXTRA_FLAGS[_f] + ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    srcs = _sources()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "master_hip.h"),
                                                                                        os.path.abspath(__file__)]     # the flags live here
    objs, jobs = [], []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(CSRC, s[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([HIPCC, *FLAGS, *EXTRA_FLAGS.get(s, []), "-c", src, "-o", obj])

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
