"""RCCL on hardware, once: no 8-GPU node has ever been available, so the multi-GPU path was covered by gloo tests
only.  This test starts a fresh child process that forms a ONE-rank "nccl" (= RCCL) group on the GPU and drives the collective path of
bench.py --gpus N through it: all_gather_into_tensor on device buffers, AsyncImageGather(submit / release_source / wait) on RCCL's own
stream against two alternating hipGraph output buffers.  It says nothing about scaling: no curve exists."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_one_rank_rccl_group_runs_the_gather_path(cuda):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank_child.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"child failed:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["backend"] == "nccl" and res["world"] == 1 and res["all_gather_into_tensor"] == "ok" and res["pooled_buffers"] == "ok"
    print("RCCL one-rank group:", res)
