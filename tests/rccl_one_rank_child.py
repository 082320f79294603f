"""Child process of tests/test_gpu_rccl.py: a ONE-rank RCCL group on the GPU (backend "nccl" = RCCL on ROCm).  Proves on hardware what the
world-size-2 gloo tests cannot: that librccl loads and initialises, that all_gather_into_tensor runs on device memory, and that the
asynchronous image gather (its own stream) is ordered correctly against hipGraph replays that reuse two static output buffers
(GraphedStylizer(copies=2) + AsyncImageGather.release_source) - the exact objects bench.py --gpus N drives.  Started BEFORE anything in
this process touches the GPU.  Prints one JSON line; exit code 0 = every check passed."""
import json
import os
import sys
import warnings

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    from mastermetastyletransfer_amd import GraphedStylizer, MasterStyleTransferModel
    from mastermetastyletransfer_amd.distributed import AsyncImageGather, all_gather_images, all_gather_tensor

    # 1. a plain collective on device memory
    ids = torch.full((1,), -1, dtype=torch.int64, device=dev)
    all_gather_tensor(ids, torch.tensor([0], dtype=torch.int64, device=dev))
    torch.cuda.synchronize()
    assert ids.tolist() == [0], ids
    x = torch.randn(3, 3, 16, 16, device=dev)
    y = torch.empty_like(x)
    all_gather_tensor(y, x)
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    assert all_gather_images(x, 3, 0, 1) is x
    res["all_gather_into_tensor"] = "ok"

    # 2. the bench's N > 1 step loop on one rank: two captured graphs with alternating output buffers, the gather reading them in place
    torch.manual_seed(1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = MasterStyleTransferModel(style_encoder_window_size=[7, 7], style_decoder_window_size=[7, 7], allow_random_swin_encoder=True)
    m = m.to(dev).eval().set_compute_dtype(torch.bfloat16)
    B, S, steps = 4, 64, 6
    g = torch.Generator().manual_seed(3)
    cs = [torch.rand(B, 3, S, S, generator=g).to(dev) for _ in range(steps)]
    ss = [torch.rand(B, 3, S, S, generator=g).to(dev) for _ in range(steps)]
    with torch.no_grad():
        want = [m(c, s, 1).clone() for c, s in zip(cs, ss)]
    graphed = GraphedStylizer(m, B, S, S, 1, copies=2)
    gather = AsyncImageGather(B, 0, 1, depth=2, force_collective=True)
    outs = [torch.empty_like(want[0]) for _ in range(steps)]
    with torch.no_grad():
        for i in range(steps):
            graphed.content.copy_(cs[i]); graphed.style.copy_(ss[i])
            gather.release_source(graphed.outs[i % 2])        # the gather of step i - 2 may still read this buffer
            local = graphed.replay(i)
            gather.submit(local, out=outs[i])
        gather.wait()
    torch.cuda.synchronize()
    bad = [i for i in range(steps) if not torch.equal(outs[i], want[i])]
    assert not bad, f"gathered images of steps {bad} differ from the eager forward"
    res["graph_replay_plus_async_gather_steps"] = steps
    # pooled destination buffers (what bench.py uses): the last result is valid after wait()
    with torch.no_grad():
        for i in range(steps):
            graphed.content.copy_(cs[i]); graphed.style.copy_(ss[i])
            gather.release_source(graphed.outs[i % 2])
            gather.submit(graphed.replay(i))
        last = gather.wait()
    torch.cuda.synchronize()
    assert torch.equal(last, want[-1])
    res["pooled_buffers"] = "ok"
    dist.destroy_process_group()
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
