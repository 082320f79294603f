"""Which torch (aten) operators still run inside one train step, and from which line of the package.

The kernel trace names torch's kernels (vectorized_elementwise_kernel<add>, reduce_kernel, ...) but not who asked for them.  This runs
one step of `bench.py --mode train` (k given on the command line) under a TorchDispatchMode with the autograd engine kept on the
calling thread, and prints, per (operator, output shape, call site), how often it ran.  Call sites are the innermost frame inside
mastermetastyletransfer_amd/; operators the autograd engine issues itself (gradient accumulation of a tensor with several consumers)
show up as "<autograd engine>".

usage (GPU box):  python profiles/tools/train_torch_ops.py [k=2]
"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

PKG = os.sep + "mastermetastyletransfer_amd" + os.sep
NO_KERNEL = {"aten.view.default", "aten.detach.default", "aten.t.default", "aten.transpose.int", "aten.permute.default", "aten.reshape.default",
             "aten._unsafe_view.default", "aten.expand.default", "aten.slice.Tensor", "aten.select.int", "aten.unsqueeze.default",
             "aten.squeeze.dim", "aten.as_strided.default", "aten.empty.memory_format", "aten.empty_like.default", "aten.alias.default",
             "aten.empty_strided.default", "aten.unbind.int", "aten.split.Tensor", "aten.chunk.default", "aten.narrow.default",
             "aten.new_empty.default", "aten.is_same_size.default", "aten._local_scalar_dense.default", "aten.lift_fresh.default",
             "aten.unflatten.int", "aten.flatten.using_ints", "aten.squeeze.default", "aten.view_as.default", "aten.result_type.Tensor"}


class Tally(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()
        self.phase = "forward"

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if name not in NO_KERNEL:
            site = "<autograd engine>"
            for fr in reversed(traceback.extract_stack()):
                if PKG in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            shape = tuple(out.shape) if isinstance(out, torch.Tensor) else ""
            self.rows[(self.phase, name, str(shape), site)] += 1
        return out


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    torch.autograd.set_multithreading_enabled(False)
    device = torch.device("cuda:0")
    tally = Tally()
    state = {}

    # the bench's own step, rebuilt here so that forward / loss / backward / optimizer can be tallied separately
    from mastermetastyletransfer_amd import FusedAdam, VGG19_custom, custom_loss
    model = bench.build_model(device, "fp32", 7).train()
    for p in model.swin_encoder.parameters():
        p.requires_grad = False
    torch.manual_seed(4)
    vgg = VGG19_custom().features
    loss_fn = custom_loss(default_lambda_value=10.0, vgg_state_dict=vgg.state_dict()).to(device)
    loss_fn.set_compute_dtype(torch.float32)
    opt = FusedAdam([{"params": model.style_transformer.parameters()}, {"params": model.decoder.parameters()}], lr=1e-4)
    content, style = bench.synthetic_batch(8, 256, device, seed=11)

    def step(t=None):
        def ph(name):
            if t is not None:
                t.phase = name
        ph("forward"); out = model(content, style, k)
        ph("loss"); total = loss_fn(content, style, out)
        ph("zero_grad"); opt.zero_grad()
        ph("backward"); total.backward()
        ph("optimizer"); opt.step()
        return total
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with tally:
        step(tally)
    torch.cuda.synchronize()
    tot = collections.Counter()
    for (phase, *_), n in tally.rows.items():
        tot[phase] += n
    print(f"aten operators with a kernel behind them in one k={k} train step: {sum(tot.values())}  " + "  ".join(f"{p}={n}" for p, n in tot.items()))
    for (phase, name, shape, site), n in sorted(tally.rows.items(), key=lambda kv: (-kv[1], kv[0])):
        print(f"{n:4d}  {phase:9s} {name:34s} {shape:24s} {site}")


if __name__ == "__main__":
    main()
