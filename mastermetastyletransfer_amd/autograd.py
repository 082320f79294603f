"""torch.autograd.Function wrappers: HIP forward + HIP backward for every op of the train step (reference
train_only_inner_loop.py:519-575: train-mode forward, custom_loss, total_loss.backward()).

PyTorch's autograd engine only orders the calls and accumulates .grad; every forward and every backward below enqueues the
hand-written kernels of libmaster_hip.so (include/master_hip.h, "Train step").  fp32 mode only.  Shapes: activations are token-major
[M, C] matrices (M = B*H*W), feature maps NHWC.  What is still torch arithmetic here is index plumbing on tiny tensors: weight
repacks ([Cout,Cin,3,3] <-> [Cout,3,3,Cin]), the [heads,64,64] bias tile gathered from the relative-position table, sums of per-image
[B,C] partial bias gradients, and the [B,C] statistics arithmetic of the style loss.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import ops
from .ops import F32, _check, _p, _stream, load_library

F = torch.float32


def _f32c(t: Tensor) -> Tensor:
    if t.dtype != F:
        raise ops.MasterHipError("the train step runs in the fp32 mode (set_compute_dtype(torch.float32))")
    return t if t.is_contiguous() else t.contiguous()


def _ws(nfloats: int, dev) -> Optional[Tensor]:
    return torch.empty((max(int(nfloats), 4),), dtype=F, device=dev)


# ---------------------------------------------------------------------------------------------------------------------------------------
# Gradients of tensors used several times in one graph (the k StyleTransformer layers share their weights, style_transformer.py:1239;
# the StyleEncoder's shared block runs 2k times): autograd would sum them with one add launch per use and tensor (~240 of a step's
# launches).  Instead the forward counts the uses of every such tensor inside the current scope (train_forward.style_transformer),
# every backward call but the last ACCUMULATES its contribution into one buffer inside the kernel's own slab reduce
# (MASTER_ACCUMULATE) and returns None to autograd, and the last call returns the buffer.  Outside a scope, or when a backward meets
# a graph the registry was not counted for (two forwards before a backward, a second backward through a retained graph), every call
# returns its own fresh gradient as before.  The counts belong to the scope object (one per forward), not to the process.
# ---------------------------------------------------------------------------------------------------------------------------------------
ACCUMULATE = 0x100            # include/master_hip.h: MASTER_ACCUMULATE


class _Scope:
    """the use counts of ONE forward (r04: they used to live on the process-wide registry, so a second forward - recompute, another model,
    a forward inside a backward hook - silently discarded the counts an earlier graph's backward still needed); every autograd ctx holds
    the scope it was recorded in, and a scope dies with the graph that references it"""
    __slots__ = ("uses", "checked")

    def __init__(self):
        self.uses = {}                    # id(tensor) -> [remaining uses, buffer or None]
        self.checked = False              # the end-of-backward verification is queued


class _SharedGrads:
    def __init__(self):
        self.counting = None              # scope whose forward is being recorded (None: not counting)

    def begin(self) -> object:
        self.counting = _Scope()
        return self.counting

    def end(self, token) -> None:
        if self.counting is token:
            self.counting = None          # counting stops; the counts stay for the backward of this graph

    def note(self, *ts) -> object:
        """forward: count one use of every tensor that wants a gradient; returns the scope token to keep in ctx (None outside a scope)"""
        if self.counting is None:
            return None
        for t in ts:
            if t is not None and t.requires_grad:
                self.counting.uses.setdefault(id(t), [0, None])[0] += 1
        return self.counting

    def deliver(self, token, t, compute):
        """backward: compute(out, accumulate) -> gradient tensor (added into `out` when accumulate).  Returns what the autograd Function
        should return for `t`: None for every use but the last, then the buffer holding the sum."""
        e = token.uses.get(id(t)) if token is not None else None
        if e is None or (e[0] <= 1 and e[1] is None):
            if e is not None:
                token.uses.pop(id(t), None)
            return compute(None, False)
        if not token.checked:             # once per backward pass: a use that never ran backward would strand its sum - fail loudly
            token.checked = True
            torch.autograd.Variable._execution_engine.queue_callback(lambda: self._verify(token))
        e[1] = compute(None, False) if e[1] is None else compute(e[1], True)
        e[0] -= 1
        if e[0] == 0:
            token.uses.pop(id(t), None)
            return e[1]
        return None

    def slot(self, token, t, alloc):
        """backward, for a launch that writes several gradients at once: -> (buffer to write, accumulate into it?, what to return to
        autograd for `t`: None for every use but the last).  `alloc()` makes the buffer at the first use."""
        e = token.uses.get(id(t)) if token is not None else None
        if e is None or (e[0] <= 1 and e[1] is None):
            if e is not None:
                token.uses.pop(id(t), None)
            out = alloc()
            return out, False, out
        if not token.checked:
            token.checked = True
            torch.autograd.Variable._execution_engine.queue_callback(lambda: self._verify(token))
        acc = e[1] is not None
        if not acc:
            e[1] = alloc()
        e[0] -= 1
        if e[0] == 0:
            token.uses.pop(id(t), None)
            return e[1], acc, e[1]
        return e[1], acc, None

    def deliver_pair(self, token, w, b, compute):
        """weight and bias of one Linear (counted together by `note`, so always at the same use):
        compute(out_w, out_b, accumulate, want_w, want_b) -> (dw, db), one launch for both.  Returns (dw, db) as `deliver` would for each."""
        ew, eb = (token.uses.get(id(w)), token.uses.get(id(b))) if token is not None else (None, None)
        if ew is None or eb is None or ew[0] != eb[0] or (ew[1] is None) != (eb[1] is None):
            # not counted as a pair (a bias that is also used elsewhere, e.g. as an attention pad row): separate launches
            if ew is None and eb is None:
                return compute(None, None, False, True, True)
            dw = self.deliver(token, w, lambda o, acc: compute(o, None, acc, True, False)[0])
            db = self.deliver(token, b, lambda o, acc: compute(None, o, acc, False, True)[1])
            return dw, db
        if ew[0] <= 1 and ew[1] is None:
            token.uses.pop(id(w), None); token.uses.pop(id(b), None)
            return compute(None, None, False, True, True)
        if not token.checked:
            token.checked = True
            torch.autograd.Variable._execution_engine.queue_callback(lambda: self._verify(token))
        if ew[1] is None:
            ew[1], eb[1] = compute(None, None, False, True, True)
        else:
            compute(ew[1], eb[1], True, True, True)
        ew[0] -= 1; eb[0] -= 1
        if ew[0] == 0:
            token.uses.pop(id(w), None); token.uses.pop(id(b), None)
            return ew[1], eb[1]
        return None, None

    def _verify(self, token) -> None:
        token.checked = False
        stranded = [k for k, e in token.uses.items() if e[1] is not None]
        token.uses = {}
        if stranded:
            raise ops.MasterHipError(f"{len(stranded)} shared parameter gradient(s) were accumulated but never delivered: a counted use "
                                     "did not take part in this backward pass (a branch of the forward that does not reach the loss)")


shared_grads = _SharedGrads()


# ---------------------------------------------------------------------------------------------------------------------------------------
# raw launches (device guard: the autograd engine may call backward from another thread / with another current device)
# ---------------------------------------------------------------------------------------------------------------------------------------
@ops._on_device
def linear_act_fwd(a: Tensor, w: Tensor, bias: Optional[Tensor], act: int, want_pre: bool, res: Optional[Tensor] = None,
                   rowscale: Optional[Tensor] = None, rows_per_scale: int = 0) -> Tuple[Tensor, Optional[Tensor]]:
    """act(a . w^T + bias); with `res` / `rowscale`: res + rowscale[row // rows_per_scale] * (that) in the same launch (a residual branch
    under StochasticDepth "row": one factor per image)"""
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=F, device=a.device)
    pre = torch.empty_like(out) if want_pre else None
    lib = load_library()
    if res is None and rowscale is None:
        _check(lib.master_linear_act_fwd(_p(a), _p(w), _p(bias), _p(out), _p(pre), M, N, K, a.stride(0), N, act, F32, _stream()), "master_linear_act_fwd")
    else:
        _check(lib.master_linear_res_fwd(_p(a), _p(w), _p(bias), _p(res), _p(rowscale), int(rows_per_scale), _p(out), _p(pre), M, N, K, a.stride(0), N,
                                         res.stride(0) if res is not None else 0, act, F32, _stream()), "master_linear_res_fwd")
    return out, pre


@ops._on_device
def linear_dgrad(g: Tensor, w: Tensor, gelu_pre: Optional[Tensor] = None, add: Optional[Tensor] = None, rowscale: Optional[Tensor] = None,
                 rows_per_scale: int = 0) -> Tensor:
    """g . w (* gelu'(gelu_pre)) (* rowscale[row // rows_per_scale]: g belongs to a StochasticDepth branch) (+ add: the gradient arriving
    over a skip connection) - all in the GEMM's epilogue"""
    M, N = g.shape
    K = w.shape[1]
    da = torch.empty((M, K), dtype=F, device=g.device)
    if ops._COUNT_ON:
        ops._count("linear_bwd", flops=2.0 * M * N * K, bytes=4.0 * (M * N + N * K + M * K * (1 + (gelu_pre is not None) + (add is not None))))
    _check(load_library().master_linear_dgrad(_p(g), _p(w), _p(da), _p(gelu_pre), _p(add), _p(rowscale), int(rows_per_scale), M, N, K, g.stride(0), K, K,
                                              add.stride(0) if add is not None else 0, F32, _stream()), "master_linear_dgrad")
    return da


@ops._on_device
def linear_wgrad(g: Tensor, a: Optional[Tensor], out: Optional[Tensor] = None, accumulate: bool = False, dbias: Optional[Tensor] = None,
                 want_bias: bool = False, want_weight: bool = True, rowscale: Optional[Tensor] = None, rows_per_scale: int = 0):
    """-> (dw [N,K] = g^T . a, db [N] = column sums of g or None): one GEMM launch + one slab reduce for both (the bias gradient rides
    along in the workgroups of the weight gradient's first column tile).  want_weight False: the bias gradient alone (master_colsum)."""
    M, N = g.shape
    if not want_weight:
        gb = rowscale_add(g, None, rowscale, rowscale.numel(), rows_per_scale) if rowscale is not None else g
        return None, colsum(gb, dbias, accumulate)
    K = a.shape[1]
    lib = load_library()
    dw = out if out is not None else torch.empty((N, K), dtype=F, device=g.device)
    db = dbias if dbias is not None else (torch.empty((N,), dtype=F, device=g.device) if want_bias else None)
    nws = lib.master_gemm_workspace_floats(N, K)
    ws = _ws(nws, g.device)
    if ops._COUNT_ON:
        ops._count("linear_bwd", flops=2.0 * M * N * K, bytes=4.0 * (M * N + M * K + N * K))
    _check(lib.master_linear_wgrad(_p(g), _p(a), _p(dw), _p(db), _p(rowscale), int(rows_per_scale), M, N, K, g.stride(0), a.stride(0), _p(ws), ws.numel(),
                                   F32 | (ACCUMULATE if accumulate else 0), _stream()), "master_linear_wgrad")
    return dw, db


@ops._on_device
def colsum(x: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    M, N = x.shape
    lib = load_library()
    out = out if out is not None else torch.empty((N,), dtype=F, device=x.device)
    ws = _ws(lib.master_colsum_workspace_floats(M, N), x.device)
    _check(lib.master_colsum(_p(x), _p(out), _p(ws), M, N, x.stride(0), F32 | (ACCUMULATE if accumulate else 0), _stream()), "master_colsum")
    return out


@ops._on_device
def reduce_slabs(part: Tensor, n: int, splits: int) -> Tensor:
    out = torch.empty((n,), dtype=F, device=part.device)
    _check(load_library().master_reduce_slabs(_p(part), _p(out), n, n, splits, _stream()), "master_reduce_slabs")
    return out


class _ReduceSeg(ctypes.Structure):            # include/master_hip.h: master_reduce_seg
    _fields_ = [("part", ctypes.c_void_p), ("out", ctypes.c_void_p), ("n", ctypes.c_size_t), ("slab", ctypes.c_size_t),
                ("splits", ctypes.c_int), ("accumulate", ctypes.c_int)]


@ops._on_device
def reduce_segments(segs: Sequence[Tuple[Tensor, Tensor, int, int, int, bool]]) -> None:
    """[(part, out, n, slab, splits, accumulate)] (<= 4): out[i] (+)= sum_k part[k*slab + i], all in one launch"""
    arr = (_ReduceSeg * len(segs))()
    for a, (part, out, n, slab, splits, acc) in zip(arr, segs):
        a.part, a.out, a.n, a.slab, a.splits, a.accumulate = part.data_ptr(), out.data_ptr(), n, slab, splits, int(acc)
    _check(load_library().master_reduce_segments(ctypes.cast(arr, ctypes.c_void_p), len(segs), _stream()), "master_reduce_segments")


@ops._on_device
def layernorm_bwd(x: Tensor, gamma: Tensor, g: Tensor, eps: float, dgb: Optional[Tensor] = None, accumulate: bool = False,
                  add: Optional[Tensor] = None):
    """-> (dx (+ add), dgb [2,C] = dgamma | dbeta); with `dgb` given and accumulate the parameter gradients are added to it"""
    M, C = x.shape
    lib = load_library()
    dx = torch.empty_like(x)
    dgb = dgb if dgb is not None else torch.empty((2, C), dtype=F, device=x.device)
    ws = _ws(lib.master_layernorm_bwd_workspace_floats(M, C), x.device)
    _check(lib.master_layernorm_bwd(_p(x), _p(gamma), _p(g), _p(add), _p(dx), _p(dgb), _p(ws), M, C, eps, F32 | (ACCUMULATE if accumulate else 0),
                                    _stream()), "master_layernorm_bwd")
    return dx, dgb


@ops._on_device
def instnorm_bwd(x: Tensor, g: Tensor, B: int, T: int, pad_row: Optional[Tensor], g_pad: Optional[Tensor], n_pad: int, eps: float):
    C = x.shape[1]
    dx = torch.empty_like(x)
    dpad = torch.empty((B, C), dtype=F, device=x.device) if n_pad > 0 else None
    _check(load_library().master_instnorm_bwd(_p(x), _p(g), _p(pad_row), _p(g_pad), _p(dx), _p(dpad), B, T, C, n_pad, eps, F32, _stream()),
           "master_instnorm_bwd")
    return dx, dpad


@ops._on_device
def rowscale_add(x: Tensor, res: Optional[Tensor], scale: Tensor, B: int, T: int) -> Tensor:
    out = torch.empty_like(x)
    _check(load_library().master_rowscale_add(_p(x), _p(res), _p(scale), _p(out), B, T, x.shape[1], F32, _stream()), "master_rowscale_add")
    return out


# ---------------------------------------------------------------------------------------------------------------------------------------
# Linear / MLP / norms / elementwise
# ---------------------------------------------------------------------------------------------------------------------------------------
def _weight_bias_grads(token, g: Tensor, a: Tensor, w: Tensor, bias: Optional[Tensor], need_w: bool, need_b: bool,
                       rowscale: Optional[Tensor] = None, rows_per_scale: int = 0):
    """(dw, db) of out = a . w^T + bias for the upstream gradient g (scaled per image by `rowscale` when the Linear sits on a
    StochasticDepth branch), through the shared-gradient registry"""
    sg = shared_grads
    rs = dict(rowscale=rowscale, rows_per_scale=rows_per_scale)
    if need_w and need_b:
        return sg.deliver_pair(token, w, bias, lambda ow, ob, acc, ww, wb: linear_wgrad(g, a, ow, acc, dbias=ob, want_bias=wb, want_weight=ww, **rs))
    dw = sg.deliver(token, w, lambda o, acc: linear_wgrad(g, a, o, acc, **rs)[0]) if need_w else None
    db = None
    if need_b:
        # a bias that trains while its weight is frozen: the rare case without a weight-gradient GEMM to ride in - scale a copy of g
        gb = rowscale_add(g, None, rowscale, rowscale.numel(), rows_per_scale) if rowscale is not None else g
        db = sg.deliver(token, bias, lambda o, acc: colsum(gb, o, acc))
    return dw, db


class LinearFn(torch.autograd.Function):
    """a [M,K] @ w[N,K]^T + bias   (nn.Linear: Wq/Wk/Wv/proj, style_transformer.py:114-116,156).
    With `res` (and `mask` [B], T tokens per image): res + mask[image] * (a @ w^T + bias) - the projection of an attention branch with its
    StochasticDepth row mask and residual add (style_transformer.py:390-391) in the same launch; in the backward the mask is applied
    inside the data- and weight-gradient GEMMs and the upstream gradient goes to `res` untouched."""

    @staticmethod
    def forward(ctx, a: Tensor, w: Tensor, bias: Optional[Tensor], res: Optional[Tensor] = None, mask: Optional[Tensor] = None, T: int = 0):
        a, w = _f32c(a), _f32c(w)
        res_c = _f32c(res) if res is not None else None
        out, _ = linear_act_fwd(a, w, bias, ops.ACT_NONE, False, res=res_c, rowscale=mask, rows_per_scale=T)
        ctx.save_for_backward(a, w, bias, mask)
        ctx.has_bias = bias is not None
        ctx.T = int(T)
        ctx.shared = shared_grads.note(w, bias)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        a, w, bias, mask = ctx.saved_tensors
        g = _f32c(g)
        rs = dict(rowscale=mask, rows_per_scale=ctx.T)
        da = linear_dgrad(g, w, **rs) if ctx.needs_input_grad[0] else None
        dw, db = _weight_bias_grads(ctx.shared, g, a, w, bias if ctx.has_bias else None, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2], **rs)
        n_in = len(ctx.needs_input_grad)                             # apply() may have been called with the three Linear arguments only
        return (da, dw, db, (g if (n_in > 3 and ctx.needs_input_grad[3]) else None), None, None)[:n_in]


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm.  fork: also return x itself - the caller uses THAT alias for the skip connection around the norm
    (style_transformer.py:390-397), so x has this Function as its only consumer and the gradient arriving over the skip connection is
    added inside the backward kernel instead of by one autograd add launch per block."""

    @staticmethod
    def forward(ctx, x: Tensor, gamma: Tensor, beta: Tensor, eps: float, fork: bool = False):
        x = _f32c(x)
        y = ops.layernorm(x, gamma.detach(), beta.detach(), eps)
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        ctx.shared = shared_grads.note(gamma)              # gamma and beta travel together: one [2,C] buffer, counted on gamma
        return (y, x) if fork else y

    @staticmethod
    def backward(ctx, g: Tensor, g_skip: Optional[Tensor] = None):
        x, gamma = ctx.saved_tensors
        box = {}
        skip = _f32c(g_skip) if g_skip is not None else None
        if g is None:                                        # only the alias was used
            return skip, None, None, None, None

        def run(o, acc):
            box["dx"], dgb = layernorm_bwd(x, gamma.detach(), _f32c(g), ctx.eps, o, acc, add=skip)
            return dgb
        dgb = shared_grads.deliver(ctx.shared, gamma, run)
        if dgb is None:
            return box["dx"], None, None, None, None
        return box["dx"], dgb[0], dgb[1], None, None


class MLPFn(torch.autograd.Function):
    """fc2(gelu(fc1(LayerNorm?(x))))  - torchvision MLP (+ the block's norm2), style_transformer.py:366,:392,:839-841,:991.
    The GELU backward is the epilogue of fc2's data-gradient GEMM (gelu'(pre) on the saved pre-activation)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, ln_g, ln_b, eps: float, fork: bool = False, mask: Optional[Tensor] = None, T: int = 0):
        """fork: also return x (the caller adds the skip connection itself).  mask [B] (T tokens per image): return the whole residual
        block  x + mask[image] * mlp(norm(x))  - fc2's epilogue adds the skip connection and applies the StochasticDepth row mask."""
        x, w1, w2 = _f32c(x), _f32c(w1), _f32c(w2)
        xn = ops.layernorm(x, ln_g.detach(), ln_b.detach(), eps) if ln_g is not None else x
        h, pre = linear_act_fwd(xn, w1, b1, ops.ACT_GELU, True)
        if mask is not None:
            y, _ = linear_act_fwd(h, w2, b2, ops.ACT_NONE, False, res=x, rowscale=mask, rows_per_scale=T)
        else:
            y, _ = linear_act_fwd(h, w2, b2, ops.ACT_NONE, False)
        ctx.save_for_backward(x, xn, pre, h, w1, w2, ln_g, b1, b2, mask)
        ctx.eps = eps
        ctx.T = int(T)
        ctx.shared = shared_grads.note(w1, b1, w2, b2, ln_g)
        return (y, x) if (fork and mask is None) else y

    @staticmethod
    def backward(ctx, g: Tensor, g_skip: Optional[Tensor] = None):
        x, xn, pre, h, w1, w2, ln_g, b1, b2, mask = ctx.saved_tensors
        g = _f32c(g)
        rs = dict(rowscale=mask, rows_per_scale=ctx.T)
        # the gradient reaching x over the skip connection: the upstream gradient itself when the block's residual add is fused (mask),
        # the alias' gradient when the caller added the skip connection (fork)
        skip = g if mask is not None else (_f32c(g_skip) if g_skip is not None else None)
        need = ctx.needs_input_grad
        sg, tok = shared_grads, ctx.shared
        dw2, db2 = _weight_bias_grads(tok, g, h, w2, b2, need[3], need[4], **rs)
        dpre = linear_dgrad(g, w2, gelu_pre=pre, **rs)                  # (mask * g . W2) * gelu'(pre)
        dw1, db1 = _weight_bias_grads(tok, dpre, xn, w1, b1, need[1], need[2])
        dx = dg = dbeta = None
        if need[0] or (ln_g is not None and (need[5] or need[6])):
            if ln_g is not None:
                dxn = linear_dgrad(dpre, w1)
                box = {}

                def run(o, acc):
                    box["dx"], dgb = layernorm_bwd(x, ln_g.detach(), dxn, ctx.eps, o, acc, add=skip)
                    return dgb
                dgb = sg.deliver(tok, ln_g, run)
                dx = box["dx"]
                if dgb is not None:
                    dg, dbeta = dgb[0], dgb[1]
            else:
                dx = linear_dgrad(dpre, w1, add=skip)                   # the skip gradient joins in the GEMM's epilogue
        elif skip is not None:
            dx = skip
        return (dx, dw1, db1, dw2, db2, dg, dbeta, None, None, None, None)[:len(ctx.needs_input_grad)]


class InstNormFn(torch.autograd.Function):
    """nn.InstanceNorm2d(affine=False) over the T tokens of each image (+ n_pad copies of pad_row in the statistics,
    style_transformer.py:520-530).  Returns (y, normalised pad rows [B,C] or None)."""

    @staticmethod
    def forward(ctx, x: Tensor, pad_row: Optional[Tensor], B: int, T: int, n_pad: int, eps: float):
        x = _f32c(x)
        pr = _f32c(pad_row) if (pad_row is not None and n_pad > 0) else None
        y, pad_out = ops.instnorm(x, B, T, pad_row=pr.detach() if pr is not None else None, n_pad=n_pad if pr is not None else 0, eps=eps)
        ctx.save_for_backward(x, pr)
        ctx.dims = (B, T, n_pad if pr is not None else 0, eps)
        return y, pad_out

    @staticmethod
    def backward(ctx, gy: Tensor, gpad: Optional[Tensor]):
        x, pr = ctx.saved_tensors
        B, T, n_pad, eps = ctx.dims
        gp = _f32c(gpad) if (gpad is not None and n_pad > 0) else None
        dx, dpad = instnorm_bwd(x, _f32c(gy), B, T, pr, gp, n_pad, eps)
        dpr = dpad.sum(0) if (dpad is not None and ctx.needs_input_grad[1]) else None
        return dx, dpr, None, None, None, None


class ChannelAffineFn(torch.autograd.Function):
    """x * gamma[c] + beta[c] - the affine of nn.InstanceNorm2d(dim, affine=True) (style_transformer.py:983-984)"""

    @staticmethod
    def forward(ctx, x: Tensor, gamma: Tensor, beta: Tensor):
        x = _f32c(x)
        ctx.save_for_backward(x, gamma)
        return ops.channel_affine(x, gamma.detach().float().contiguous(), beta.detach().float().contiguous())

    @staticmethod
    @ops._on_device
    def backward(ctx, g: Tensor):
        x, gamma = ctx.saved_tensors
        g = _f32c(g)
        lib = load_library()
        M, C = x.shape
        dx = ops.channel_affine(g, gamma.detach().float().contiguous(), None) if ctx.needs_input_grad[0] else None
        dgamma = None
        if ctx.needs_input_grad[1]:
            dgamma = torch.empty((C,), dtype=F, device=x.device)
            ws = _ws(lib.master_colsum_workspace_floats(M, C), x.device)
            _check(lib.master_colsum_prod(_p(g), _p(x), _p(dgamma), _p(ws), M, C, C, F32, _stream()), "master_colsum_prod")
        dbeta = colsum(g) if ctx.needs_input_grad[2] else None
        return dx, dgamma, dbeta


class SoftmaxRowsFn(torch.autograd.Function):
    """softmax(scale * x) over rows (the regular-MHA decoder end, style_transformer.py:1100-1106)"""

    @staticmethod
    def forward(ctx, x: Tensor, scale: float, n_valid: int):
        p = ops.softmax_rows(_f32c(x), scale, n_valid)
        ctx.save_for_backward(p)
        ctx.scale, ctx.n_valid = scale, n_valid
        return p

    @staticmethod
    @ops._on_device
    def backward(ctx, g: Tensor):
        (p,) = ctx.saved_tensors
        g = _f32c(g)
        dx = torch.empty_like(p)
        _check(load_library().master_softmax_rows_bwd(_p(p), _p(g), _p(dx), p.shape[0], ctx.n_valid, p.shape[1], float(ctx.scale), F32,
                                                      _stream()), "master_softmax_rows_bwd")
        return dx, None, None


class RowScaleAddFn(torch.autograd.Function):
    """res + scale[b] * x  -  residual add with torchvision StochasticDepth(p, "row") on the branch (style_transformer.py:390-397)"""

    @staticmethod
    def forward(ctx, x: Tensor, res: Tensor, scale: Tensor, B: int, T: int):
        ctx.save_for_backward(scale)
        ctx.dims = (B, T)
        return rowscale_add(_f32c(x), _f32c(res), scale, B, T)

    @staticmethod
    def backward(ctx, g: Tensor):
        (scale,) = ctx.saved_tensors
        B, T = ctx.dims
        g = _f32c(g)
        dx = rowscale_add(g, None, scale, B, T) if ctx.needs_input_grad[0] else None
        return dx, (g if ctx.needs_input_grad[1] else None), None, None, None


class MulAddFn(torch.autograd.Function):
    """Query * sigma + mu (style_transformer.py:1123)"""

    @staticmethod
    @ops._on_device
    def forward(ctx, q: Tensor, sigma: Tensor, mu: Tensor):
        q, sigma, mu = _f32c(q), _f32c(sigma), _f32c(mu)
        out = torch.empty_like(q)
        _check(load_library().master_mul_add_fwd(_p(q), _p(sigma), _p(mu), _p(out), q.numel(), F32, _stream()), "master_mul_add_fwd")
        ctx.save_for_backward(q, sigma)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        q, sigma = ctx.saved_tensors
        g = _f32c(g)
        dq, ds = torch.empty_like(q), torch.empty_like(q)
        with torch.cuda.device(g.device):
            _check(load_library().master_mul_add_bwd(_p(g), _p(q), _p(sigma), _p(dq), _p(ds), q.numel(), F32, _stream()), "master_mul_add_bwd")
        return dq, ds, g


class BiasTileFn(torch.autograd.Function):
    """relative_position_bias_table [entries, heads] -> the [heads,64,64] bias tile of the attention kernels (style_transformer.py:21-28;
    rows >= N zero, key slots >= N at -1e30): one launch each way (master_bias_tile_fwd / _bwd)"""

    @staticmethod
    @ops._on_device
    def forward(ctx, table: Tensor, index: Tensor, N: int):
        table = _f32c(table)
        index = index.reshape(-1).contiguous()
        assert index.dtype == torch.int64 and index.numel() == N * N
        heads = table.shape[1]
        tile = torch.empty((heads, 64, 64), dtype=F, device=table.device)
        _check(load_library().master_bias_tile_fwd(_p(table), _p(index), _p(tile), N, heads, _stream()), "master_bias_tile_fwd")
        ctx.save_for_backward(index)
        ctx.dims = (N, heads, table.shape[0])
        return tile

    @staticmethod
    def backward(ctx, g: Tensor):
        (index,) = ctx.saved_tensors
        N, heads, entries = ctx.dims
        g = _f32c(g)
        dtable = torch.empty((entries, heads), dtype=F, device=g.device)
        with torch.cuda.device(g.device):
            _check(load_library().master_bias_tile_bwd(_p(g), _p(index), _p(dtable), N, heads, entries, _stream()), "master_bias_tile_bwd")
        return dtable, None, None


# ---------------------------------------------------------------------------------------------------------------------------------------
# windowed attention core
# ---------------------------------------------------------------------------------------------------------------------------------------
class WindowAttnFn(torch.autograd.Function):
    """master_window_attn_fwd / master_window_attn_bwd.  q, k, v0 (v1) are [M, C] column blocks of their BASE matrices tq/tk/tv0/tv1
    at column offsets oq/ok/ov0/ov1 (a fused qkv projection is passed as the same base three times): the backward writes each gradient
    block straight into one gradient buffer per distinct base.  k_pad / v_pad: [C] or [B,C] rows of zero-padded slots.
    bias64: [heads,64,64] (differentiable: it is gathered from the relative-position table)."""

    @staticmethod
    def forward(ctx, tq, tk, tv0, tv1, k_pad, v0_pad, v1_pad, bias64, offs, plan, B: int, T: int, heads: int, scale: float):
        C = heads * 32
        oq, ok, ov0, ov1 = offs
        bases = [_f32c(tq), _f32c(tk), _f32c(tv0), _f32c(tv1) if tv1 is not None else None]
        q, k, v0 = bases[0][:, oq:oq + C], bases[1][:, ok:ok + C], bases[2][:, ov0:ov0 + C]
        v1 = bases[3][:, ov1:ov1 + C] if bases[3] is not None else None
        kp, v0p = _f32c(k_pad), _f32c(v0_pad)
        v1p = _f32c(v1_pad) if v1 is not None else None
        o0, o1 = ops.window_attn(q, k, v0, v1, kp.detach(), v0p.detach(), v1p.detach() if v1p is not None else None, bias64.detach(),
                                 plan.src, plan.region, B, T, heads, plan.nW, plan.N, scale)
        ctx.save_for_backward(bases[0], bases[1], bases[2], bases[3], kp, v0p, v1p, bias64)
        ctx.meta = (offs, plan, B, T, heads, scale, None)
        ctx.base_ids = [id(t) if t is not None else None for t in (tq, tk, tv0, tv1)]
        # parameter-like inputs used by several attention calls of one graph (the bias tile and the pad rows = Linear biases of the k
        # weight-sharing layers): their gradients are summed inside this Function's own reduce launch (shared_grads).  The original
        # tensors are kept for that (their ids are the registry keys); a v1 pad row that IS the v0 pad row counts once.
        ctx.pad_src = (k_pad, v0_pad, v1_pad if v1 is not None else None, bias64)
        ctx.same_v = v1 is not None and v1_pad is v0_pad
        pads = [t for t in (k_pad, v0_pad, None if ctx.same_v else ctx.pad_src[2]) if t is not None and t.dim() == 1 and plan.n_pad > 0]
        ctx.shared = shared_grads.note(*pads, bias64)
        return o0, o1

    @staticmethod
    def backward(ctx, g0: Tensor, g1: Optional[Tensor]):
        bq, bk, bv0, bv1, kp, v0p, v1p, bias64 = ctx.saved_tensors
        (oq, ok, ov0, ov1), plan, B, T, heads, scale, _ = ctx.meta
        C = heads * 32
        dev = bq.device
        dual = bv1 is not None
        g0 = _f32c(g0)
        g1 = _f32c(g1) if dual else None
        if dual and g1 is None:
            g1 = torch.zeros_like(g0)
        # one gradient buffer per distinct base matrix; blocks not covered by q/k/v stay zero
        bases, offsets, ids = [bq, bk, bv0, bv1], [oq, ok, ov0, ov1], ctx.base_ids
        bufs = {}
        for t, i in zip(bases, ids):
            if t is not None and i not in bufs:
                cover = sum(C for tt, ii in zip(bases, ids) if tt is not None and ii == i)
                bufs[i] = torch.empty_like(t) if cover == t.shape[1] else torch.zeros_like(t)
        dq = bufs[ids[0]][:, oq:oq + C]; dk = bufs[ids[1]][:, ok:ok + C]; dv0 = bufs[ids[2]][:, ov0:ov0 + C]
        dv1 = bufs[ids[3]][:, ov1:ov1 + C] if dual else None
        n_pad = plan.n_pad
        dkp = torch.empty((B, max(n_pad, 1), C), dtype=F, device=dev) if n_pad > 0 else None
        dvp = torch.empty((2 if dual else 1, B, max(n_pad, 1), C), dtype=F, device=dev) if n_pad > 0 else None      # v0 | v1 pad-row gradients, adjacent
        dv0p = dvp[0] if n_pad > 0 else None
        dv1p = dvp[1] if (n_pad > 0 and dual) else None
        need_bias = ctx.needs_input_grad[7]
        dbws = torch.empty((B * plan.nW, heads, 64, 64), dtype=F, device=dev) if need_bias else None
        q, k, v0 = bq[:, oq:oq + C], bk[:, ok:ok + C], bv0[:, ov0:ov0 + C]
        v1 = bv1[:, ov1:ov1 + C] if dual else None
        kbs = kp.stride(0) if kp.dim() == 2 else 0
        vbs = v0p.stride(0) if v0p.dim() == 2 else 0
        if ops._COUNT_ON:
            nv = 2 if dual else 1
            ops._count("window_attn_bwd", flops=2.0 * B * plan.nW * heads * plan.N * plan.N * 32 * (4 + 3 * nv),
                       bytes=4.0 * B * T * C * (2 + 2 * nv) * 2)
        with torch.cuda.device(dev):
            _check(load_library().master_window_attn_bwd(
                _p(q), _p(k), _p(v0), _p(v1), q.stride(0), k.stride(0), v0.stride(0), _p(kp), _p(v0p), _p(v1p), kbs, vbs, _p(bias64),
                _p(plan.src), _p(plan.region), _p(plan.pad_ord), _p(g0), _p(g1), g0.stride(0), _p(dq), _p(dk), _p(dv0), _p(dv1),
                dq.stride(0), dk.stride(0), dv0.stride(0), _p(dkp), _p(dv0p), _p(dv1p), _p(dbws), B, T, heads, plan.nW, plan.N, n_pad,
                float(scale), F32, _stream()), "master_window_attn_bwd")
        if dual:
            assert dv0.stride(0) == dv1.stride(0) and v0.stride(0) == v1.stride(0)
        # gradients per input position: a base that appears several times gets its buffer once
        outs, seen = [], set()
        for i in ids:
            if i is None or i in seen:
                outs.append(None)
            else:
                seen.add(i); outs.append(bufs[i])

        # ---- pad-row and bias-tile gradients: every reduction of this call in one launch (they were a reduce launch + one torch sum per
        #      pad row, and autograd then added the uses together) ------------------------------------------------------------------------
        k_src, v0_src, v1_src, bias_src = ctx.pad_src
        sg, tok = shared_grads, ctx.shared
        segs = []
        need = ctx.needs_input_grad

        def vec_grad(rows, like, src, nrows):
            """gradient of a [C] pad row = sum of `rows` [nrows, C]; through the registry when the row is shared"""
            out, acc, ret = sg.slot(tok, src, lambda: torch.empty((C,), dtype=F, device=dev))
            segs.append((rows, out, C, C, nrows, acc))
            return ret

        def pad_grad(rows, like, src, nrows):
            if rows is None:
                return None                                     # no padded slots: no gradient through the pad row
            if like.dim() == 1:
                return vec_grad(rows, like, src, nrows)
            return rows.sum(1)                                  # [B,C] pad rows (one per image: the sigma/mu attention's normalised Key row)
        gkp = pad_grad(dkp, kp, k_src, B * max(n_pad, 1)) if need[4] else None
        if dual and ctx.same_v and v0p.dim() == 1 and (need[5] or need[6]):
            gv0p = pad_grad(dvp, v0p, v0_src, 2 * B * max(n_pad, 1)) if n_pad > 0 else None      # one row used for both value streams: one sum over both
            gv1p = None
        else:
            gv0p = pad_grad(dv0p, v0p, v0_src, B * max(n_pad, 1)) if need[5] else None
            gv1p = pad_grad(dv1p, v1p, v1_src, B * max(n_pad, 1)) if (dual and need[6]) else None
        dbias = None
        if need_bias:
            out, acc, dbias = sg.slot(tok, bias_src, lambda: torch.empty((heads, 64, 64), dtype=F, device=dev))
            segs.append((dbws, out, heads * 4096, heads * 4096, B * plan.nW, acc))
        if segs:
            # segments that write the same buffer (one tensor passed for two pad rows) must not share a launch: workgroups of one
            # launch run in any order, and the second segment adds to what the first one writes
            seen_out, batch = set(), []
            for sg_ in segs:
                if sg_[1].data_ptr() in seen_out:
                    reduce_segments(batch)
                    seen_out, batch = set(), []
                seen_out.add(sg_[1].data_ptr()); batch.append(sg_)
            reduce_segments(batch)
        return (outs[0], outs[1], outs[2], outs[3], gkp, gv0p, gv1p, dbias, None, None, None, None, None, None)


# ---------------------------------------------------------------------------------------------------------------------------------------
# decoder convolution (decoder.py:24-54): Upsample? -> ReflectionPad -> Conv3x3 -> ReLU?
# ---------------------------------------------------------------------------------------------------------------------------------------
def _pad32(c: int) -> int:
    return (c + 31) // 32 * 32


@ops._on_device
def conv_backward(g: Tensor, y_relu: Optional[Tensor], x: Tensor, weight: Tensor, upsample: bool, reflect: bool, g_nchw: bool,
                  need_dx: bool, need_dw: bool, need_db: bool, dgrad_relu_src: Optional[Tensor] = None, dgrad_add: Optional[Tensor] = None,
                  premasked: bool = False, wf: Optional[Tensor] = None):
    """shared by the decoder (reflect) and VGG (zero padding) convolutions.  g: gradient of the conv (+ReLU) output, NHWC or NCHW;
    y_relu: the ReLU output (None: no ReLU or g already masked); x: the conv input NHWC [B,Hin,Win,Cin]; weight [Cout,Cin,3,3].
    Returns (dx NHWC like x, dweight, dbias)."""
    lib = load_library()
    B, Hin, Win, Cin = x.shape
    Cout = weight.shape[0]
    up = 1 if upsample else 0
    H, W = Hin << up, Win << up
    Cp = _pad32(Cout)
    dev = x.device
    if Cout == 3 and Cin == 32 and reflect and not upsample and g_nchw and y_relu is None and dgrad_add is None and dgrad_relu_src is None:
        # the decoder's last layer (32 -> 3, no ReLU): direct kernels on the NCHW image gradient - no 32-channel padded copy of it, no
        # GEMMs that spend 29/32 of their work on zero channels
        g = _f32c(g)
        w3 = _f32c(weight.detach())
        dw = db = dx = None
        if need_dw or need_db:
            dw = torch.empty((3, Cin, 3, 3), dtype=F, device=dev)
            db = torch.empty((3,), dtype=F, device=dev) if need_db else None
            ws = _ws(lib.master_conv3x3_cout3_wgrad_workspace_floats(B, H, W), dev)
            if ops._COUNT_ON:
                ops._count("conv_wgrad", flops=2.0 * B * H * W * 3 * 9 * Cin, bytes=4.0 * (B * H * W * 3 + x.numel() + 3 * 9 * Cin))
            _check(lib.master_conv3x3_cout3_wgrad(_p(g), _p(x), _p(dw), _p(db), _p(ws), ws.numel(), B, H, W, Cin, _stream()), "master_conv3x3_cout3_wgrad")
            if not need_dw:
                dw = None
        if need_dx:
            dxp = torch.empty((B, H + 2, W + 2, Cin), dtype=F, device=dev)
            if ops._COUNT_ON:
                ops._count("conv_dgrad", flops=2.0 * B * H * W * 3 * 9 * Cin, bytes=4.0 * (B * H * W * (3 + Cin) + 3 * 9 * Cin))
            _check(lib.master_conv3x3_cout3_dgrad(_p(g), _p(w3), _p(dxp), B, H, W, Cin, _stream()), "master_conv3x3_cout3_dgrad")
            dx = torch.empty((B, H, W, Cin), dtype=F, device=dev)
            _check(lib.master_fold_reflect(_p(dxp), _p(dx), B, H, W, Cin, 0, F32, _stream()), "master_fold_reflect")
        return dx, dw, db
    if premasked and Cp == Cout and not g_nchw:
        gm = g
    else:
        gm = torch.empty((B, H, W, Cp), dtype=F, device=dev)
        _check(lib.master_relu_mask(_p(g), _p(y_relu), _p(gm), B, H * W, Cout, Cp, int(g_nchw), F32, _stream()), "master_relu_mask")
    db = None
    if need_db and not need_dw:
        db = colsum(gm.view(B * H * W, Cp))[:Cout]
    dw = None
    if need_dw:
        dwp = torch.empty((Cp, 3, 3, Cin), dtype=F, device=dev)
        dbp = torch.empty((Cp,), dtype=F, device=dev) if need_db else None            # the bias gradient rides along in the same launch
        ws = _ws(lib.master_gemm_workspace_floats(Cp, 9 * Cin), dev)
        if ops._COUNT_ON:
            ops._count("conv_wgrad", flops=2.0 * B * H * W * Cout * 9 * Cin, bytes=4.0 * (B * H * W * Cout + x.numel() + Cout * 9 * Cin))
        _check(lib.master_conv3x3_wgrad(_p(gm), _p(x), _p(dwp), _p(dbp), B, Hin, Win, Cin, Cp, Cp, up, 0 if reflect else 1, _p(ws), ws.numel(), F32,
                                        _stream()), "master_conv3x3_wgrad")
        if dbp is not None:
            db = dbp[:Cout]
        dw = dwp[:Cout].permute(0, 3, 1, 2).contiguous()                               # [Cout,3,3,Cin] -> [Cout,Cin,3,3]
    dx = None
    if need_dx:
        if wf is None:                                                                 # [Cin][3][3][Cp], taps flipped (the forward packs it when it can)
            wf = conv_weight_pack(weight, want_flipped=True)[1]
        if ops._COUNT_ON:
            ops._count("conv_dgrad", flops=2.0 * B * H * W * Cout * 9 * Cin, bytes=4.0 * (B * H * W * (Cout + Cin) + Cout * 9 * Cin))
        if reflect:
            dxp = torch.empty((B, H + 2, W + 2, Cin), dtype=F, device=dev)
            _check(lib.master_conv3x3_dgrad(_p(gm), _p(wf), _p(dxp), None, None, B, H, W, Cp, Cin, 1, F32, _stream()), "master_conv3x3_dgrad")
            dx = torch.empty((B, Hin, Win, Cin), dtype=F, device=dev)
            _check(lib.master_fold_reflect(_p(dxp), _p(dx), B, H, W, Cin, up, F32, _stream()), "master_fold_reflect")
        else:
            dx = torch.empty((B, H, W, Cin), dtype=F, device=dev)
            _check(lib.master_conv3x3_dgrad(_p(gm), _p(wf), _p(dx), _p(dgrad_relu_src), _p(dgrad_add), B, H, W, Cp, Cin, 0, F32, _stream()),
                   "master_conv3x3_dgrad")
    return dx, dw, db


@ops._on_device
def conv_weight_pack(weight: Tensor, want_flipped: bool):
    """nn.Conv2d weight [Cout,Cin,3,3] -> (wp [Cout,3,3,Cin], wf [Cin,3,3,Cp] or None) in one launch (master_conv_weight_pack)"""
    w = _f32c(weight.detach())
    Cout, Cin = w.shape[:2]
    Cp = _pad32(Cout)
    wp = torch.empty((Cout, 3, 3, Cin), dtype=F, device=w.device)
    wf = torch.empty((Cin, 3, 3, Cp), dtype=F, device=w.device) if want_flipped else None
    _check(load_library().master_conv_weight_pack(_p(w), _p(wp), _p(wf), Cout, Cin, Cp, _stream()), "master_conv_weight_pack")
    return wp, wf


class ConvReflectFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Tensor, upsample: bool, relu: bool, out_nchw: bool):
        x = _f32c(x)
        wp, wf = conv_weight_pack(weight, want_flipped=x.requires_grad)          # the data gradient's layout comes out of the same launch
        y = ops.conv3x3_reflect(x, wp, bias.detach().float().contiguous(), upsample=upsample, relu=relu, out_nchw=out_nchw)
        ctx.save_for_backward(x, weight, y if relu else None, wf)
        ctx.flags = (upsample, relu, out_nchw)
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        x, weight, y, wf = ctx.saved_tensors
        upsample, relu, out_nchw = ctx.flags
        dx, dw, db = conv_backward(_f32c(g), y, x, weight, upsample, True, out_nchw, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                   ctx.needs_input_grad[2], wf=wf)
        return dx, dw, db, None, None, None


# ---------------------------------------------------------------------------------------------------------------------------------------
# perceptual loss (codes/loss.py:201-262): VGG-19 features of [content | style | output] + content / style terms, one Function
# ---------------------------------------------------------------------------------------------------------------------------------------
VGG19_CONVS = (0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28)
VGG19_POOLS = (4, 9, 18, 27)
VGG19_TAPS = (6, 11, 20, 29)
_CPAD = 32
EPS_IN = 1e-5


class PerceptualLossFn(torch.autograd.Function):
    """(content_loss, style_loss) = f(content, style, output; frozen VGG-19 `vgg` = loss.VGG19_custom).  Only `output` receives a gradient.

    forward: the three feature extractions as one batch of 3B (conv outputs kept for the backward), per-(image, channel) statistics,
    IN-distance (content) and mean/std distance (style) at relu2_1/3_1/4_1/5_1.
    backward: per tap the loss gradient w.r.t. the output image's feature map (master_loss_tap_bwd), then the VGG chain in reverse on
    the B output images only: conv data gradients with the ReLU mask of the layer below and the next tap's gradient in the GEMM
    epilogue, max-pool backward fused with the ReLU mask; ends in the fp32 image gradient [B,3,H,W].
    The loss network runs in vgg.compute_dtype: fp32 (parity mode) or bf16 (bf16 MFMA, fp32 accumulation; the network is frozen, so
    there is no bf16 weight gradient anywhere) - statistics, loss terms and the image gradient are fp32 in both."""

    @staticmethod
    def forward(ctx, content, style, output, vgg, squared_c: bool, squared_s: bool):
        B = output.shape[0]
        dev = output.device
        dt = vgg.compute_dtype
        acts = {}                         # conv index -> ReLU output [3B,H,W,C]
        taps = []
        h = None
        for i in range(30):
            if i in VGG19_CONVS:
                if i == 0:                # conv1_1 straight from the NCHW images: one NHWC batch [content | style | output]
                    h = ops.vgg_conv1([content, style, output.detach()], vgg.features[0].weight, vgg.features[0].bias, dt)
                else:
                    w, b = vgg._weights(i)
                    h = ops.conv3x3_zero(h, w, b, relu=True, logical_cin=vgg.features[i].weight.shape[1])
                acts[i] = h
            elif i in VGG19_POOLS:
                h = ops.maxpool2x2(h)
            elif i in VGG19_TAPS:
                taps.append(h)
        # the [B,C] arithmetic of both terms and the style term's gradient coefficients: one single-workgroup launch per tap
        # (master_loss_terms_fwd) - as torch elementwise operators it was ~30 launches per tap, forward and backward each
        losses = torch.empty((2,), dtype=F, device=dev)
        stats = []
        lib = load_library()
        for t, f in enumerate(taps):
            N, H, W, C = f.shape
            mean, var = ops.channel_stats(f)
            part = ops.instnorm_distance(f, mean, var, B, 2 * B, squared_c, EPS_IN, partials_only=True)
            coefA = torch.empty((B, C), dtype=F, device=dev)
            coefB = torch.empty((B, C), dtype=F, device=dev)
            _check(lib.master_loss_terms_fwd(_p(mean), _p(var), _p(part), part.numel(), _p(losses), _p(coefA), _p(coefB), B, H * W, C,
                                             int(squared_s), int(t == 0), _stream()), "master_loss_terms_fwd")
            stats.append((mean, var, coefA, coefB))
        content_loss, style_loss = losses[0], losses[1]
        ctx.B, ctx.flags, ctx.vgg = B, (squared_c, squared_s), vgg
        ctx.acts, ctx.taps, ctx.stats = acts, taps, stats
        return content_loss, style_loss

    @staticmethod
    def backward(ctx, g_content: Tensor, g_style: Tensor):
        B = ctx.B
        squared_c, squared_s = ctx.flags
        lib = load_library()
        acts, taps, stats, vgg = ctx.acts, ctx.taps, ctx.stats, ctx.vgg
        dev = taps[0].device
        dt = taps[0].dtype
        DT = ops._dt(taps[0])
        # upstream gradients of the two scalars stay on the device (a float() here drained the queue at the start of every backward)
        gc = g_content.to(F).contiguous() if g_content is not None else None
        gs = g_style.to(F).contiguous() if g_style is not None else None
        with torch.cuda.device(dev):
            # ---- loss gradient at the four taps, for the output images [2B, 3B) ------------------------------------------------------
            tap_grads = []
            for t, (f, (mean, var, coefA, coefB)) in enumerate(zip(taps, stats)):
                N, H, W, C = f.shape
                HW = H * W
                dx = torch.empty((B, H, W, C), dtype=dt, device=dev)
                ws = _ws(lib.master_loss_tap_workspace_floats(B, HW, C), dev)
                _check(lib.master_loss_tap_bwd(_p(f), _p(mean), _p(var), _p(coefA), _p(coefB), _p(ws), _p(dx), B, HW, C, EPS_IN, _p(gc), _p(gs),
                                               int(squared_c), int(t == 3), DT, _stream()), "master_loss_tap_bwd")
                tap_grads.append(dx)
            # ---- VGG-19 backward on the output images --------------------------------------------------------------------------------
            # layer list: conv0 conv2 pool4 conv5[tap0] conv7 pool9 conv10[tap1] conv12 conv14 conv16 pool18 conv19[tap2] conv21 conv23
            #             conv25 pool27 conv28[tap3].  g = gradient of conv i's ReLU output, already masked by that ReLU.
            sl = slice(2 * B, 3 * B)
            out_of = lambda i: acts[i][sl]
            tap_of = {5: 0, 10: 1, 19: 2}                # conv whose ReLU output is tap t (joined in the epilogue of the conv above)
            g = tap_grads[3]                             # relu5_1 = ReLU(conv28): masked inside master_loss_tap_bwd (relu_gate)
            order = list(VGG19_CONVS)
            for pos in range(len(order) - 1, -1, -1):
                i = order[pos]
                below = order[pos - 1] if pos > 0 else None
                pooled = (i - 1) in VGG19_POOLS
                if below is None or pooled:
                    relu_src, add = None, None           # (a pool's backward applies the ReLU mask of `below`)
                else:
                    relu_src = out_of(below)
                    add = tap_grads[tap_of[below]] if below in tap_of else None
                wf = vgg._flipped(i)                                                   # [Cx][3][3][Cg], cached (the network is frozen)
                Bn, H, W, Cg = g.shape
                Cx = wf.shape[0]
                dx = torch.empty((Bn, H, W, Cx), dtype=dt, device=dev)
                if ops._COUNT_ON:      # algorithmic: the convolution's real channel counts (the 3-channel image is computed padded to 32)
                    cx_real = vgg.features[i].weight.shape[1]
                    ops._count("conv_dgrad", flops=2.0 * Bn * H * W * Cg * 9 * cx_real,
                               bytes=g.element_size() * (Bn * H * W * (Cg + cx_real * (1 + (relu_src is not None) + (add is not None))) + Cg * 9 * cx_real))
                _check(lib.master_conv3x3_dgrad(_p(g), _p(wf), _p(dx), _p(relu_src), _p(add), Bn, H, W, Cg, Cx, 0,
                                                ops.x3_code(wf) if ops.is_split(wf) else DT, _stream()), "master_conv3x3_dgrad")
                g = dx
                if pooled:
                    y = out_of(below)                     # the pool's input = ReLU output of the conv below
                    Np, Hp, Wp, Cpp = y.shape
                    dpool = torch.empty_like(y)
                    _check(lib.master_maxpool2x2_bwd(_p(g), _p(y), _p(dpool), Np, Hp, Wp, Cpp, DT, _stream()), "master_maxpool2x2_bwd")
                    g = dpool
            Bn, H, W, Cp = g.shape
            dimg = torch.empty((Bn, 3, H, W), dtype=F, device=dev)
            _check(lib.master_nhwc_to_image(_p(g), _p(dimg), Bn, H * W, Cp, DT, _stream()), "master_nhwc_to_image")
        ctx.acts = ctx.taps = ctx.stats = ctx.vgg = None
        return None, None, dimg, None, None, None
