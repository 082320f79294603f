"""Upper bounds for a persistent per-token chain in the bf16 StyleTransformer, by LAUNCH DELETION on one box.

The proposed kernel keeps a 128-token tile on its CU across `attention-out -> proj + residual -> [LN] -> MLP -> + residual` and
streams the next link's weights under the current epilogue.  Whatever its schedule, it still pays each absorbed Linear's MFMA time and
weight stream, so the step with those launches DELETED (outputs left unwritten: timing only, the image is garbage) is a strict upper
bound on what the fusion can gain.  The forward is captured as a hipGraph per variant (no host time in the numbers) and the variants are
replayed round-robin so that box drift cancels.

usage: python profiles/tools/chain_bound.py [batch]
"""
import os, sys, time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                                    # noqa: E402  (build_model / synthetic_batch)
from mastermetastyletransfer_amd import GraphedStylizer, ops    # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
model = bench.build_model(dev, "bf16", 7)
content, style = bench.synthetic_batch(B, 256, dev, seed=11)
T = B * 1024                                                     # StyleTransformer tokens per stream at 256^2

real_linear, real_group, real_mlp, real_ln = ops.linear, ops.linear_group, ops.mlp, ops.linear_ln
log, skip = [], {"fn": lambda kind, M, N, K, res: False}


def linear(a, w, bias=None, r1=None, r2=None, act=ops.ACT_NONE, out=None):
    M, K = a.shape; N = w.shape[0]
    if log is not None:
        log.append(("linear", M, N, K, r2 is not None or r1 is not None))
    if a.dtype == torch.bfloat16 and skip["fn"]("linear", M, N, K, r2 is not None):
        return out if out is not None else torch.empty((M, N), dtype=a.dtype, device=a.device)
    return real_linear(a, w, bias, r1=r1, r2=r2, act=act, out=out)


def linear_group(xs, ws, biases):
    if log is not None:
        log.append(("group", [tuple(x.shape) for x in xs], [w.shape[0] for w in ws], 0, False))
    if xs[0].dtype == torch.bfloat16 and skip["fn"]("group", xs[0].shape[0], ws[0].shape[0], xs[0].shape[1], False):
        return [torch.empty((x.shape[0], w.shape[0]), dtype=x.dtype, device=x.device) for x, w in zip(xs, ws)]
    return real_group(xs, ws, biases)


def mlp(x, w1, b1, w2, b2, res=None, ln_gamma=None, ln_beta=None, **kw):
    if log is not None:
        log.append(("mlp", x.shape[0], w1.shape[0], x.shape[1], res is not None))
    if x.dtype == torch.bfloat16 and skip["fn"]("mlp", x.shape[0], w1.shape[0], x.shape[1], res is not None):
        return torch.empty_like(x)
    return real_mlp(x, w1, b1, w2, b2, res=res, ln_gamma=ln_gamma, ln_beta=ln_beta, **kw)


def linear_ln(x, ln_gamma, ln_beta, w, bias=None, r2=None, eps=1e-5):
    M, K = x.shape; N = w.shape[0]
    if log is not None:
        log.append(("linear_ln", M, N, K, r2 is not None))
    if x.dtype == torch.bfloat16 and skip["fn"]("linear", M, N, K, r2 is not None):
        return torch.empty((M, N), dtype=x.dtype, device=x.device)
    return real_ln(x, ln_gamma, ln_beta, w, bias, r2=r2, eps=eps)


ops.linear, ops.linear_group, ops.mlp, ops.linear_ln = linear, linear_group, mlp, linear_ln

with torch.no_grad():
    model(content, style, 1); log.clear(); model(content, style, 1)
st = [e for e in log if e[0] == "group" or (e[1] % T == 0 and e[3] == 256)]           # the StyleTransformer's launches (C = 256, whole-batch token matrices)
print(f"bs={B}: {len(log)} linear / grouped / MLP calls per forward, of them on StyleTransformer token matrices:")
for e in st:
    print("   ", e)
log = None

is_st = lambda M, K: M % T == 0 and K == 256
variants = {
    "baseline": lambda kind, M, N, K, res: False,
    # the Linear right in front of an MLP in the chain: a projection with the residual add (N = K = 256, one stream)
    "no proj+residual linears": lambda kind, M, N, K, res: kind == "linear" and is_st(M, K) and N == 256 and res,
    "no StyleTransformer linears at all": lambda kind, M, N, K, res: kind in ("linear", "group") and is_st(M, K),
    "no StyleTransformer MLPs": lambda kind, M, N, K, res: kind == "mlp" and is_st(M, K) and N == 1024,
}
graphs = {}
for name, fn in variants.items():
    skip["fn"] = fn
    graphs[name] = GraphedStylizer(model, B, 256, 256, 1)
    graphs[name].content.copy_(content); graphs[name].style.copy_(style)
skip["fn"] = variants["baseline"]
torch.cuda.synchronize()
times = {n: [] for n in variants}
for rnd in range(12):
    for name, g in graphs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.graph.replay()
        e1.record(); torch.cuda.synchronize()
        if rnd >= 2:
            times[name].append(e0.elapsed_time(e1) / 20)
base = sorted(times["baseline"])[len(times["baseline"]) // 2]
print(f"\nhipGraph replay, 10 rounds x 20 steps per variant, round-robin; median ms/step (min)")
for name, ts in times.items():
    ts = sorted(ts); med = ts[len(ts) // 2]
    print(f"  {name:40s} {med:7.3f} ({ts[0]:.3f})   {'' if name == 'baseline' else f'deleting them saves {base - med:6.3f} ms = {100 * (base - med) / base:4.1f} % of the step'}")
