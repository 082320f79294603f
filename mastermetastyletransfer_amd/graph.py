"""HIP-graph replay of the stylization forward for a fixed (batch, H, W, k).

The forward is ~75 kernel launches; at bs=32 the host enqueues them in ~1.0 ms against ~2.4 ms of device time, but at
serving batch sizes (1-4 pairs) the device work shrinks below the enqueue time and the path becomes launch-bound.
Capturing the launches once (torch.cuda.CUDAGraph = hipGraph on ROCm) and replaying them removes the per-launch host
cost: one graph launch per call.  Results are bit-identical to the eager path (same kernels, same order).

Every launch of the package goes to torch's current stream through the C ABI, so stream capture sees all of them; the
lazily built host-side caches (derived weights, window tables, kernel attributes) are filled by eager warm-up calls
before the capture.
"""
from __future__ import annotations

import torch
from torch import Tensor


class GraphedStylizer:
    """stylize = GraphedStylizer(model, batch, H, W, k); out = stylize(content, style)

    `out` is the graph's static output buffer: it is overwritten by the next call (pass copy=True to get a private tensor).

    copies > 1 captures the same forward that many times, each capture with its own static OUTPUT buffer (the inputs are shared):
    `replay(i)` runs capture i % copies and returns outs[i % copies].  A consumer that reads step i's output asynchronously - the
    all-gather of a multi-GPU run (distributed.AsyncImageGather) - then reads it in place while step i + 1 writes the other buffer:
    no per-step clone.  The caller must make the stream wait for that consumer before replaying into the same buffer again
    (AsyncImageGather.release_source).
    """

    def __init__(self, model, batch: int, height: int, width: int, transformer_layer_count: int = 1, warmup: int = 2, copies: int = 1):
        p = next(model.parameters())
        if p.device.type != "cuda":
            raise ValueError("GraphedStylizer needs the model on a GPU (there is no CPU path)")
        self.model, self.k = model, int(transformer_layer_count)
        self.content = torch.zeros(batch, 3, height, width, device=p.device, dtype=torch.float32)
        self.style = torch.zeros_like(self.content)
        side = torch.cuda.Stream(device=p.device)
        side.wait_stream(torch.cuda.current_stream(p.device))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(max(1, warmup)):                    # fills every lazy cache outside the capture
                model(self.content, self.style, self.k)
        torch.cuda.current_stream(p.device).wait_stream(side)
        torch.cuda.synchronize(p.device)
        self.graphs, self.outs = [], []
        for _ in range(max(1, int(copies))):
            g = torch.cuda.CUDAGraph()
            # thread_local: only this thread's calls are checked during capture (a process-group watchdog thread of a multi-rank run
            # may query events concurrently)
            with torch.no_grad(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = model(self.content, self.style, self.k)
            self.graphs.append(g); self.outs.append(out)
        self.graph, self.out = self.graphs[0], self.outs[0]

    def replay(self, i: int = 0) -> Tensor:
        """run capture i % copies on the inputs currently in self.content / self.style; returns that capture's static output buffer"""
        j = i % len(self.graphs)
        self.graphs[j].replay()
        return self.outs[j]

    def __call__(self, content_image: Tensor, style_image: Tensor, copy: bool = False) -> Tensor:
        if content_image.shape != self.content.shape or style_image.shape != self.style.shape:
            raise ValueError(f"this graph was captured for inputs of shape {tuple(self.content.shape)}")
        self.content.copy_(content_image, non_blocking=True)
        self.style.copy_(style_image, non_blocking=True)
        self.graph.replay()
        return self.out.clone() if copy else self.out


class GraphedTrainStep:
    """One inner-loop train step (train_only_inner_loop.py:519-575: train-mode forward, custom_loss, zero_grad, backward, Adam) as a
    hipGraph per transformer-layer count k.

        step = GraphedTrainStep(model, loss_fn, FusedAdam(..., capturable=True))
        total = step(content, style, k)          # device scalar; the graph's static output for that k

    The step is ~800 kernel launches, many of them a few microseconds long (the StyleTransformer's [8192, 256] matrices at batch 8): the
    host cannot enqueue them as fast as the device finishes them and the device idles for ~3 ms of a 30 ms step.  Replaying a captured
    graph removes the host from the loop.  The first call for a given k runs eagerly (a real training step that also fills the lazy
    host-side caches) and then captures; every later call with that k copies the batch into the static input buffers and replays.  The
    kernels and their order are those of the eager step, so are the numbers; StochasticDepth masks come from torch's graph-safe device
    generator (fresh masks every replay).

    Inputs must keep one shape; the optimizer must be FusedAdam(capturable=True) (its step count and learning rate live in device
    memory; a changed `param_groups[i]["lr"]` is pushed before the next replay).  Gradients are in `param.grad` after a call, as after
    an eager step: every captured k owns its gradient buffers (they live in that graph's memory pool), and a replay re-binds
    `param.grad` to the buffers the replayed graph just wrote - so gradient clipping, logging or a data-parallel all-reduce between two
    calls read THIS step's gradients whichever k ran before."""

    def __init__(self, model, loss_fn, optimizer, capture: bool = True):
        from .optim import FusedAdam
        if not isinstance(optimizer, FusedAdam) or not optimizer.capturable:
            raise ValueError("GraphedTrainStep needs FusedAdam(capturable=True)")
        self.model, self.loss_fn, self.opt, self.capture = model, loss_fn, optimizer, bool(capture)
        self.content = self.style = None
        self.graphs, self.totals, self.seen, self.grads = {}, {}, {}, {}
        self._params = [p for g in optimizer.param_groups for p in g["params"]]

    def _eager(self, k: int) -> Tensor:
        out = self.model(self.content, self.style, k)
        total = self.loss_fn(self.content, self.style, out)
        self.opt.zero_grad(set_to_none=True)
        total.backward()
        self.opt.step()
        return total

    def __call__(self, content: Tensor, style: Tensor, k: int) -> Tensor:
        k = int(k)
        if self.content is None:
            self.content, self.style = torch.empty_like(content), torch.empty_like(style)
        if content.shape != self.content.shape or style.shape != self.style.shape:
            raise ValueError(f"GraphedTrainStep was started with inputs of shape {tuple(self.content.shape)}")
        self.content.copy_(content, non_blocking=True)
        self.style.copy_(style, non_blocking=True)
        g = self.graphs.get(k)
        if g is None:
            total = self._eager(k).detach()
            self.seen[k] = self.seen.get(k, 0) + 1
            if self.capture and self.seen[k] >= 1:
                eager_grads = [(p_, p_.grad) for p_ in self._params]
                self._capture(k)                          # leaves param.grad on the graph's (not yet written) buffers
                for p_, g_ in eager_grads:                # this call's gradients are the eager step's
                    p_.grad = g_
            return total
        self.opt.push_hypers()
        g.replay()
        from .optim import _mark_written
        _mark_written(self._params)                       # the replay wrote the parameters behind torch's back
        for p_, g_ in self.grads[k]:                      # ... and THIS graph's gradient buffers: param.grad may still point at another k's
            p_.grad = g_
        return self.totals[k]

    def _capture(self, k: int) -> None:
        dev = self.content.device
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        self.opt.zero_grad(set_to_none=True)              # the capture allocates the gradients in the graph's pool
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            total = self._eager(k)
        self.graphs[k], self.totals[k] = g, total.detach()
        self.grads[k] = [(p_, p_.grad) for p_ in self._params]
