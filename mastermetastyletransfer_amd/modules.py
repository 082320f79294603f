"""Host-side mirror of the reference's module tree for the stylization path.

Same class names, constructor arguments, sub-module names and state-dict keys as
codes/style_transformer.py, codes/decoder.py and torchvision's swin_b.features[:4], so checkpoints
written by the reference's trainers load unchanged.  The modules only HOLD
parameters; every forward enqueues hand-written HIP kernels through the C ABI (ops.py).  There is no
torch-op fallback: CPU tensors raise.

Layout: activations are token-major [B*H*W, C] matrices in the compute dtype (fp32 parity mode, or
bf16 throughput mode); the public forwards take/return the reference's [B,H,W,C] (or NCHW) fp32.
Linear layers run on the un-padded token matrix; pad/roll/window-partition are index tables consumed
by the attention kernel, and a zero-padded slot's key/value row is the projection bias.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
from torch import nn, Tensor

from . import ops
from .window_plan import relative_position_index, window_plan

HEAD_DIM = 32
EPS = 1e-5


def _inference_only(m: nn.Module):
    """inference entry points of modules that have no train-mode form of their own (the window attention modules are trained through
    the blocks that own them)"""
    if m.training:
        raise NotImplementedError(
            "train-mode forward of this module alone is not on the HIP path: train through StyleTransformer / Decoder / "
            "MasterStyleTransferModel (train_forward.py), or call .eval()")


def _train_mode(m: nn.Module) -> bool:
    """True when the call must build an autograd graph on the HIP backward kernels (train_forward.py): the module is in train mode,
    like the reference's trainer (`.train()`, train_only_inner_loop.py:239).  An eval-mode forward never records a graph (inference
    kernels only), whatever torch.is_grad_enabled() says: set the stochastic-depth probabilities to 0 to train without the masks."""
    if not m.training:
        return False
    if getattr(m, "compute_dtype", torch.float32) != torch.float32 or getattr(m, "split_weights", False):
        raise NotImplementedError("the train step runs in the exact-fp32 mode: call set_compute_dtype(torch.float32) before .train()")
    if any(getattr(sub, "_inference_only_arch", False) for sub in m.modules()):
        raise NotImplementedError("bias-free projections and a ReLU MLP activation are implemented for inference only (the backward kernels "
                                  "cover the reference's defaults: qkv_bias = proj_bias = True, nn.GELU)")
    return True


class _HipModule(nn.Module):
    """nn.Module with a cache of tensors derived from parameters (fused / repacked / bf16 copies)."""
    compute_dtype = torch.float32
    split_weights = False          # "bf16" / "f16" (or True = "bf16"): the split fast parity modes (ops.F32X3 / ops.F16X3) - fp32 activations, GEMM / conv weights as 16-bit hi + lo

    def __getstate__(self):
        """pickling / copy.deepcopy / torch.save(module): the caches of derived device tensors (and their events) are rebuilt on demand"""
        d = dict(self.__dict__)
        d.pop("_derived_cache", None); d.pop("_zero_bias", None)
        return d

    def _derived(self, name: str, srcs: Sequence[Tensor], build: Callable[[], Tensor]) -> Tensor:
        cache = self.__dict__.setdefault("_derived_cache", {})
        key = tuple((s.data_ptr(), s._version, s.dtype, str(s.device)) for s in srcs) + (self.compute_dtype, self.split_weights)
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                t = build()
            # the tensor is built on the CURRENT stream; a forward issued later on another stream (multi-stream serving) must not read it
            # before that work is done: remember where and when it was built (r05)
            ev = None
            if t.is_cuda:
                ev = torch.cuda.Event()
                ev.record()
            hit = (key, t, ev, torch.cuda.current_stream().cuda_stream if t.is_cuda else None)
            cache[name] = hit
        elif hit[2] is not None:
            cs = torch.cuda.current_stream()
            if cs.cuda_stream != hit[3] and not torch.cuda.is_current_stream_capturing():
                cs.wait_event(hit[2])
        return hit[1]

    def _w(self, name: str, *params: Tensor) -> Tensor:
        """weight(s) concatenated along dim 0, contiguous, in the compute dtype"""
        def build():
            w = params[0] if len(params) == 1 else torch.cat(list(params), dim=0)
            if self.split_weights:
                return ops.split_weights(w, fmt=self.split_weights)
            return ops.convert(w.detach().contiguous(), self.compute_dtype)
        return self._derived("w:" + name, params, build)

    def _b(self, name: str, *params: Tensor) -> Tensor:
        """bias(es) concatenated, fp32 (GEMM epilogues are fp32 in both modes)"""
        def build():
            b = params[0] if len(params) == 1 else torch.cat(list(params), dim=0)
            return b.detach().float().contiguous()
        return self._derived("b:" + name, params, build)

    def _ln_linear(self, name: str, x: Tensor, norm: nn.LayerNorm, weights: Sequence[Tensor], biases: Sequence[Tensor]) -> Tensor:
        """LayerNorm(x) @ cat(weights).T + cat(biases).  bf16 mode: one fused launch; fp32 modes: LayerNorm, then Linear (ops.linear_ln)"""
        return ops.linear_ln(x, self._b("n1w", norm.weight), self._b("n1b", norm.bias), self._w(name, *weights), self._b(name, *biases), eps=norm.eps)

    def _bias_of(self, lin: nn.Linear) -> Tensor:
        """the bias of a projection, or - bias=False (qkv_bias / proj_bias of full_model.py:39-42) - a zero row of its width: F.linear
        without a bias adds nothing, and a zero-padded window slot's key / value row (the reference pads BEFORE the Linear) is then zero"""
        if lin.bias is not None:
            return lin.bias
        cache = self.__dict__.setdefault("_zero_bias", {})
        key = (lin.out_features, str(lin.weight.device))
        if key not in cache:
            cache[key] = torch.zeros(lin.out_features, dtype=torch.float32, device=lin.weight.device)
        return cache[key]

    def _row(self, name: str, param: Tensor) -> Tensor:
        """a bias used as the K/V row of zero-padded window slots: compute dtype"""
        return self._derived("row:" + name, [param], lambda: ops.convert(param.detach().contiguous(), self.compute_dtype))


def _tokens(x: Tensor, dtype) -> Tensor:
    """[B,H,W,C] fp32 -> [B*H*W, C] in the compute dtype"""
    B, H, W, C = x.shape
    return ops.convert(x.reshape(B * H * W, C).contiguous(), dtype)


def _image(x2d: Tensor, B: int, H: int, W: int) -> Tensor:
    return ops.convert(x2d, torch.float32).view(B, H, W, -1)


def _make_mlp(dim: int, hidden: int, dropout: float, act_layer: Optional[Callable[..., nn.Module]] = nn.GELU) -> nn.Sequential:
    """same children/keys as torchvision.ops.MLP: 0 Linear, 1 activation (torchvision: activation_layer(), nn.GELU or nn.ReLU here), 2 Dropout,
    3 Linear, 4 Dropout"""
    return nn.Sequential(nn.Linear(dim, hidden), (act_layer or nn.GELU)(), nn.Dropout(dropout), nn.Linear(hidden, dim), nn.Dropout(dropout))


def _init_mlp(mlp: nn.Sequential):
    for m in mlp.modules():                      # reference style_transformer.py:368-372
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.normal_(m.bias, std=1e-6)


def _mlp_fwd(owner: _HipModule, name: str, mlp: nn.Sequential, x: Tensor, residual: Optional[Tensor],
             norm: Optional[nn.LayerNorm] = None) -> Tensor:
    """residual + fc2(gelu(fc1(norm?(x))))   (torchvision MLP, call sites style_transformer.py:366,839-841,991).
    bf16 mode: one fused launch (hidden activations stay in registers, LayerNorm folded in);
    fp32 mode: LayerNorm, GEMM+GELU, GEMM+residual."""
    pad = (-mlp[0].weight.shape[0]) % 64         # the bf16 GEMM steps K in 64s, the fp32 / split ones in 32s
    if pad:
        # a hidden width that is not a multiple of 64 (mlp_ratio 1.1 at dim 256: 281; the reference takes any width): zero rows of fc1 (and of
        # its bias) and zero columns of fc2 - gelu(0) = relu(0) = 0 meets a zero weight, so the padded units contribute nothing (r05; inference)
        Fp = torch.nn.functional.pad

        def conv(t: Tensor) -> Tensor:
            return ops.split_weights(t, fmt=owner.split_weights) if owner.split_weights else ops.convert(t.contiguous(), owner.compute_dtype)
        w1 = owner._derived("w:" + name + ".0:pad", [mlp[0].weight], lambda: conv(Fp(mlp[0].weight.detach(), (0, 0, 0, pad))))
        b1 = owner._derived("b:" + name + ".0:pad", [mlp[0].bias], lambda: Fp(mlp[0].bias.detach().float(), (0, pad)).contiguous())
        w2 = owner._derived("w:" + name + ".3:pad", [mlp[3].weight], lambda: conv(Fp(mlp[3].weight.detach(), (0, pad))))
        b2 = owner._b(name + ".3", mlp[3].bias)
        if norm is not None:
            x = ops.layernorm(x, owner._b(name + ".ln_w", norm.weight), owner._b(name + ".ln_b", norm.bias), norm.eps)
        h = ops.linear(x, w1, b1, act=ops.ACT_RELU if isinstance(mlp[1], nn.ReLU) else ops.ACT_GELU)
        return ops.linear(h, w2, b2, r2=residual)
    w1, b1 = owner._w(name + ".0", mlp[0].weight), owner._b(name + ".0", mlp[0].bias)
    b2 = owner._b(name + ".3", mlp[3].bias)
    g = owner._b(name + ".ln_w", norm.weight) if norm is not None else None
    b = owner._b(name + ".ln_b", norm.bias) if norm is not None else None
    relu = isinstance(mlp[1], nn.ReLU)          # MLP_activation_layer = nn.ReLU (full_model.py:47-50): the fused kernels are GELU-only
    if not relu and ops.mlp_fused_supported(x, w1.shape[0], owner.split_weights):
        # split mode: the fused kernel reads W2 in the k order of a packed accumulator tile (ops.split_weights layout 1)
        w2f = owner._derived("w:" + name + ".3:acc_k", [mlp[3].weight], lambda: ops.split_weights(mlp[3].weight, layout=1, fmt=owner.split_weights)) \
            if owner.split_weights else owner._w(name + ".3", mlp[3].weight)
        try:
            return ops.mlp(x, w1, b1, w2f, b2, res=residual, ln_gamma=g, ln_beta=b, eps=norm.eps if norm is not None else EPS)
        except NotImplementedError:
            pass          # MASTER_ERR_UNSUPPORTED (e.g. a hidden width whose bias vectors do not fit next to the weight ring): unfused path
    w2 = owner._w(name + ".3", mlp[3].weight)
    if norm is not None:
        x = ops.layernorm(x, g, b, norm.eps)
    h = ops.linear(x, w1, b1, act=ops.ACT_RELU if relu else ops.ACT_GELU)
    return ops.linear(h, w2, b2, r2=residual)


def _check_arch(dropout, attention_dropout, qkv_bias, proj_bias, act_layer, dim, heads):
    if dropout != 0.0 or attention_dropout != 0.0:
        raise NotImplementedError("dropout > 0 is not implemented on the HIP path (reference default 0.0)")
    if act_layer is not None and act_layer not in (nn.GELU, nn.ReLU):
        raise NotImplementedError("MLP activations on the HIP path: nn.GELU (reference default) and nn.ReLU")
    if dim % heads != 0 or dim // heads != HEAD_DIM:
        raise NotImplementedError(f"head_dim must be {HEAD_DIM} (dim {dim}, heads {heads}); reference default is 256/8")


class _WindowAttentionBase(_HipModule):
    """relative-position table/index + window geometry shared by the three attention variants"""

    def _init_rel_pos(self, num_heads: int):
        ws = self.window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws[0] - 1) * (2 * ws[1] - 1), num_heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)        # style_transformer.py:220-225
        self.register_buffer("relative_position_index", relative_position_index(ws))

    def _bias64(self) -> Tensor:
        """fp32 [heads,64,64] bias tile; key slots >= N hold -1e30 (they mask the tile padding)"""
        def build():
            N = self.window_size[0] * self.window_size[1]
            t = self.relative_position_bias_table.detach().float()
            b = t[self.relative_position_index].view(N, N, -1).permute(2, 0, 1)
            out = torch.zeros((b.shape[0], 64, 64), dtype=torch.float32, device=t.device)
            out[:, :, N:] = -1e30
            out[:, :N, :N] = b
            return out.contiguous()
        return self._derived("bias64", [self.relative_position_bias_table, self.relative_position_index], build)

    def _core(self, q, k, vs, k_pad, v_pads, B, H, W):
        plan = window_plan(H, W, self.window_size, self.shift_size, q.device)
        v1 = vs[1] if len(vs) > 1 else None
        o0, o1 = ops.window_attn(q, k, vs[0], v1, k_pad, v_pads[0], v_pads[1] if v1 is not None else None,
                                 self._bias64(), plan.src, plan.region, B, H * W, self.num_heads, plan.nW, plan.N,
                                 HEAD_DIM ** -0.5, x3=self.split_weights)
        return [o0] if v1 is None else [o0, o1]


class ShiftedWindowAttention(_WindowAttentionBase):
    """Reference codes/style_transformer.py:175-292: window attention whose Q, K, V come from three inputs."""

    def __init__(self, dim: int, num_heads: int, window_size: List[int], shift_size: List[int], dropout: float = 0.0,
                 attention_dropout: float = 0.0, qkv_bias: bool = True, proj_bias: bool = True):
        super().__init__()
        if len(window_size) != 2 or len(shift_size) != 2:
            raise ValueError("window_size and shift_size must be of length 2")
        _check_arch(dropout, attention_dropout, qkv_bias, proj_bias, nn.GELU, dim, num_heads)
        self.window_size, self.shift_size, self.num_heads = list(window_size), list(shift_size), num_heads
        self.attention_dropout, self.dropout = attention_dropout, dropout
        self.Wq = nn.Linear(dim, dim, bias=qkv_bias)
        self.Wk = nn.Linear(dim, dim, bias=qkv_bias)
        self.Wv = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)
        self._inference_only_arch = not (qkv_bias and proj_bias)
        self._init_rel_pos(num_heads)

    def attend(self, xq: Tensor, xk: Tensor, xvs: Sequence[Tensor], residuals: Sequence[Optional[Tensor]],
               B: int, H: int, W: int, norm: Optional[nn.LayerNorm] = None) -> List[Tensor]:
        """token-matrix form: returns [residual_i + proj(attn(q, k, v_i))]; identical inputs (same tensor
        object) share their projection, and identical value streams share everything.  `norm` (the block's
        norm1) is applied to q, k and v; for q = k = v it is folded into the fused qkv projection."""
        C = xq.shape[1]
        if xq is xk and all(v is xq for v in xvs):
            if norm is not None:
                qkv = self._ln_linear("qkv", xq, norm, [self.Wq.weight, self.Wk.weight, self.Wv.weight],
                                      [self._bias_of(self.Wq), self._bias_of(self.Wk), self._bias_of(self.Wv)])
            else:
                wqkv = self._w("qkv", self.Wq.weight, self.Wk.weight, self.Wv.weight)
                bqkv = self._b("qkv", self._bias_of(self.Wq), self._bias_of(self.Wk), self._bias_of(self.Wv))
                qkv = ops.linear(xq, wqkv, bqkv)
            q, k, vs = qkv[:, :C], qkv[:, C:2 * C], [qkv[:, 2 * C:]]
            uniq = [0] * len(xvs)
        else:
            if norm is not None:
                memo: List[Tuple[Tensor, Tensor]] = []

                def ln(t):
                    for src, o in memo:
                        if src is t:
                            return o
                    o = ops.layernorm(t, self._b("n1w", norm.weight), self._b("n1b", norm.bias), norm.eps)
                    memo.append((t, o))
                    return o
                xq, xk, xvs = ln(xq), ln(xk), [ln(v) for v in xvs]
            uniq, seen = [], []
            for v in xvs:
                for i, s in enumerate(seen):
                    if s is v:
                        uniq.append(i)
                        break
                else:
                    uniq.append(len(seen))
                    seen.append(v)
            wv, bv = self._w("v", self.Wv.weight), self._b("v", self._bias_of(self.Wv))
            # the projections of this call read different inputs but are independent: one grouped launch (q|k + v, or q, k, v) when
            # there are at most three of them
            if xq is xk:
                probs = [(xq, self._w("qk", self.Wq.weight, self.Wk.weight), self._b("qk", self._bias_of(self.Wq), self._bias_of(self.Wk)))]
            else:
                probs = [(xq, self._w("q", self.Wq.weight), self._b("q", self._bias_of(self.Wq))),
                         (xk, self._w("k", self.Wk.weight), self._b("k", self._bias_of(self.Wk)))]
            probs += [(v, wv, bv) for v in seen]
            if len(probs) <= 3:
                outs_ = ops.linear_group([p_[0] for p_ in probs], [p_[1] for p_ in probs], [p_[2] for p_ in probs])
            else:
                outs_ = [ops.linear(*p_) for p_ in probs]
            if xq is xk:
                q, k, vs = outs_[0][:, :C], outs_[0][:, C:], outs_[1:]
            else:
                q, k, vs = outs_[0], outs_[1], outs_[2:]
        if len(vs) > 2:
            raise NotImplementedError("at most two value streams per attention call")
        kp, vp = self._row("k", self._bias_of(self.Wk)), self._row("v", self._bias_of(self.Wv))
        pw, pb = self._w("proj", self.proj.weight), self._b("proj", self._bias_of(self.proj))
        plan = window_plan(H, W, self.window_size, self.shift_size, q.device)
        if len(vs) == 1 and len({None if r_ is None else id(r_) for r_ in residuals}) == 1 and \
                ops.window_attn_proj_supported(q, self.num_heads, plan.nW, plan.N):
            # one value stream, one residual (the self-attention calls, and Scale is Shift in the first layer): attention core,
            # proj and the residual add in ONE launch
            o = ops.window_attn_proj(q, k, vs[0], kp, vp, self._bias64(), plan.src, plan.region, pw, pb, residuals[0], B, H * W,
                                     self.num_heads, plan.nW, plan.N, HEAD_DIM ** -0.5)
            return [o for _ in residuals]
        cores = self._core(q, k, vs, kp, [vp, vp], B, H, W)
        outs, done = [], {}
        for i, res in zip(uniq, residuals):
            key = (i, None if res is None else id(res))
            if key not in done:
                done[key] = ops.linear(cores[i], pw, pb, r2=res)
            outs.append(done[key])
        return outs

    def forward(self, input_q: Tensor, input_k: Tensor, input_v: Tensor) -> Tensor:
        _inference_only(self)
        B, H, W, C = input_q.shape
        cache: Dict[int, Tensor] = {}

        def tok(t):
            if id(t) not in cache:
                cache[id(t)] = _tokens(t, self.compute_dtype)
            return cache[id(t)]
        out = self.attend(tok(input_q), tok(input_k), [tok(input_v)], [None], B, H, W)[0]
        return _image(out, B, H, W)


class StyleSwinTransformerBlock(_HipModule):
    """Reference codes/style_transformer.py:303-398."""

    def __init__(self, dim: int, num_heads: int, window_size: List[int] = [8, 8], shift_size: List[int] = [4, 4],
                 dropout: float = 0.0, attention_dropout: float = 0.0, qkv_bias: bool = True, proj_bias: bool = True,
                 mlp_ratio: float = 4.0, stochastic_depth_prob: float = 0.0,
                 norm_layer: Optional[Callable[..., nn.Module]] = nn.LayerNorm,
                 MLP_activation_layer: Optional[Callable[..., nn.Module]] = nn.GELU, exclude_MLP_after: bool = False):
        super().__init__()
        if norm_layer is not None and norm_layer is not nn.LayerNorm:
            raise NotImplementedError("only nn.LayerNorm / None norm layers are implemented on the HIP path")
        _check_arch(dropout, attention_dropout, qkv_bias, proj_bias, MLP_activation_layer, dim, num_heads)
        self.exclude_MLP_after = exclude_MLP_after
        self.use_norm = norm_layer is not None
        if self.use_norm:
            self.norm1 = nn.LayerNorm(dim)
            if not exclude_MLP_after:
                self.norm2 = nn.LayerNorm(dim)
        self.attn = ShiftedWindowAttention(dim, num_heads, window_size, shift_size, dropout, attention_dropout,
                                           qkv_bias, proj_bias)
        self.stochastic_depth_prob = stochastic_depth_prob       # identity in eval mode (torchvision StochasticDepth)
        if not exclude_MLP_after:
            self.mlp = _make_mlp(dim, int(dim * mlp_ratio), dropout, MLP_activation_layer)
            _init_mlp(self.mlp)
            self._inference_only_arch = MLP_activation_layer is nn.ReLU

    def _ln(self, which: str, x: Tensor) -> Tensor:
        n = getattr(self, which)
        return ops.layernorm(x, self._b(which + ".w", n.weight), self._b(which + ".b", n.bias), n.eps)

    def run(self, xq: Tensor, xk: Tensor, xvs: Sequence[Tensor], residuals: Sequence[Tensor], B, H, W) -> List[Tensor]:
        """token-matrix form for one q/k pair and one or two value streams (shared softmax)"""
        outs = self.attn.attend(xq, xk, xvs, residuals, B, H, W, norm=self.norm1 if self.use_norm else None)
        if not self.exclude_MLP_after:
            outs = [_mlp_fwd(self, "mlp", self.mlp, o, o, norm=self.norm2 if self.use_norm else None) for o in outs]
        return outs

    def forward(self, input_q: Tensor, input_k: Tensor, input_v: Tensor, calculating_Key_in_encoder: bool = None) -> Tensor:
        B, H, W, C = input_q.shape
        if _train_mode(self):
            from . import train_forward as TF
            memo: Dict[int, Tensor] = {}

            def tk(t):
                if id(t) not in memo:
                    memo[id(t)] = t.reshape(B * H * W, C).float()
                return memo[id(t)]
            res = tk(input_q) if (calculating_Key_in_encoder is True or not self.exclude_MLP_after) else tk(input_v)
            return TF.block(self, tk(input_q), tk(input_k), [tk(input_v)], [res], B, H, W, TF.sd_masks)[0].view(B, H, W, C)
        cache: Dict[int, Tensor] = {}

        def tok(t):
            if id(t) not in cache:
                cache[id(t)] = _tokens(t, self.compute_dtype)
            return cache[id(t)]
        # residual source, style_transformer.py:383-386
        res = tok(input_q) if (calculating_Key_in_encoder is True or not self.exclude_MLP_after) else tok(input_v)
        return _image(self.run(tok(input_q), tok(input_k), [tok(input_v)], [res], B, H, W)[0], B, H, W)


class ShiftedWindowAttention_for_decoder_last_MHA(_WindowAttentionBase):
    """Reference codes/style_transformer.py:414-764: sigma/mu cross attention (un-projected, instance-
    normalised Query; Key instance-normalised after Wk over the zero-padded map; two value streams, one
    softmax, shared proj)."""

    def __init__(self, dim: int, num_heads: int, window_size: List[int], shift_size: List[int],
                 instance_norm_q: Optional[nn.Module] = None, instance_norm_k: Optional[nn.Module] = None,
                 dropout: float = 0.0, attention_dropout: float = 0.0, qkv_bias: bool = True, proj_bias: bool = True,
                 use_q_proj: bool = False, use_Key_instance_norm_after_linear_transformation: bool = True):
        super().__init__()
        if len(window_size) != 2 or len(shift_size) != 2:
            raise ValueError("window_size and shift_size must be of length 2")
        _check_arch(dropout, attention_dropout, qkv_bias, proj_bias, nn.GELU, dim, num_heads)
        if use_q_proj:
            raise NotImplementedError("use_q_proj=True is not implemented on the HIP path (StyleDecoder hard-codes False, style_transformer.py:1022)")
        # the reference registers the norm modules it is handed as children (:645-646): with affine=True their weight/bias appear in the
        # state dict under this module too (shared with StyleDecoder.instance_norm_Query / _Key)
        if instance_norm_q is not None and getattr(instance_norm_q, "affine", False):
            self.instance_norm_q = instance_norm_q
        if instance_norm_k is not None and getattr(instance_norm_k, "affine", False):
            self.instance_norm_k = instance_norm_k
        self.window_size, self.shift_size, self.num_heads = list(window_size), list(shift_size), num_heads
        self.use_q_proj = use_q_proj
        self.use_Key_instance_norm_after_linear_transformation = use_Key_instance_norm_after_linear_transformation
        self.Wk = nn.Linear(dim, dim, bias=qkv_bias)
        self.Wv_scale = nn.Linear(dim, dim, bias=qkv_bias)
        self.Wv_shift = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)
        self._inference_only_arch = not (qkv_bias and proj_bias)
        self._init_rel_pos(num_heads)

    def sigma_mu(self, xq: Tensor, xk: Tensor, xs: Tensor, xsh: Tensor, B: int, H: int, W: int,
                 fuse_query: Optional[Tensor] = None, q_norm_repeat: int = 1):
        """token-matrix form.  Returns (sigma, mu), or Query*sigma+mu when fuse_query is given
        (style_transformer.py:1123 folded into the second proj's epilogue)."""
        T = H * W
        nq, nk = getattr(self, "instance_norm_q", None), getattr(self, "instance_norm_k", None)      # affine norms, or None
        aff = lambda t, n, tag: t if n is None else ops.channel_affine(t, self._b(tag + ".w", n.weight), self._b(tag + ".b", n.bias))
        q, _ = ops.instnorm(xq, B, T, repeat=q_norm_repeat)                         # :468 (repeat=2: also the caller's :1056)
        q = aff(q, nq, "inq")
        if not self.use_Key_instance_norm_after_linear_transformation:
            xk, _ = ops.instnorm(xk, B, T)                                          # :470-472
            xk = aff(xk, nk, "ink")
        # Wk . Key | Wv_scale . Scale | Wv_shift . Shift  (:515-517): three inputs, three weights, one grouped launch
        k, vs, vsh = ops.linear_group([xk, xs, xsh],
                                      [self._w("k", self.Wk.weight), self._w("vs", self.Wv_scale.weight), self._w("vsh", self.Wv_shift.weight)],
                                      [self._b("k", self._bias_of(self.Wk)), self._b("vs", self._bias_of(self.Wv_scale)),
                                       self._b("vsh", self._bias_of(self.Wv_shift))])
        k_pad = self._row("k", self._bias_of(self.Wk))
        if self.use_Key_instance_norm_after_linear_transformation:
            # :520-530  statistics over the zero-padded pH x pW map: its padded slots hold Wk.bias
            plan = window_plan(H, W, self.window_size, self.shift_size, xq.device)
            k, k_pad_b = ops.instnorm(k, B, T, pad_row=k_pad, n_pad=plan.n_pad)
            k = aff(k, nk, "ink")
            if k_pad_b is not None:
                k_pad = k_pad_b if nk is None else aff(k_pad_b.contiguous(), nk, "ink")
        o_s, o_sh = self._core(q, k, [vs, vsh], k_pad, [self._row("vs", self._bias_of(self.Wv_scale)), self._row("vsh", self._bias_of(self.Wv_shift))],
                               B, H, W)
        pw, pb = self._w("proj", self.proj.weight), self._b("proj", self._bias_of(self.proj))
        mu = ops.linear(o_sh, pw, pb)                                               # :594-595
        if fuse_query is not None:
            return ops.linear(o_s, pw, pb, r1=fuse_query, r2=mu)                    # Query * sigma + mu
        return ops.linear(o_s, pw, pb), mu                                          # :575-576

    def forward(self, input_q: Tensor, input_k: Tensor, input_v_scale: Tensor, input_v_shift: Tensor):
        _inference_only(self)
        B, H, W, C = input_q.shape
        dt = self.compute_dtype
        s, m = self.sigma_mu(_tokens(input_q, dt), _tokens(input_k, dt), _tokens(input_v_scale, dt),
                             _tokens(input_v_shift, dt), B, H, W)
        return _image(s, B, H, W), _image(m, B, H, W)


class StyleEncoder(_HipModule):
    """Reference codes/style_transformer.py:777-912."""

    def __init__(self, encoder_dim: int, encoder_num_heads: int, encoder_window_size: List[int],
                 encoder_shift_size: List[int], encoder_mlp_ratio: float = 4.0, encoder_dropout: float = 0.0,
                 encoder_attention_dropout: float = 0.0, encoder_qkv_bias: bool = True, encoder_proj_bias: bool = True,
                 encoder_stochastic_depth_prob: float = 0.1, encoder_norm_layer: Callable[..., nn.Module] = None,
                 encoder_MLP_activation_layer: Optional[Callable[..., nn.Module]] = nn.GELU,
                 encoder_if_use_processed_Key_in_Scale_and_Shift_calculation: bool = True):
        super().__init__()
        self.if_use_processed_Key_in_Scale_and_Shift_calculation = encoder_if_use_processed_Key_in_Scale_and_Shift_calculation
        self.encoder_stochastic_depth_prob = encoder_stochastic_depth_prob
        self.shared_MHA_without_MLP = StyleSwinTransformerBlock(
            dim=encoder_dim, num_heads=encoder_num_heads, window_size=encoder_window_size, shift_size=encoder_shift_size,
            dropout=encoder_dropout, attention_dropout=encoder_attention_dropout, qkv_bias=encoder_qkv_bias,
            proj_bias=encoder_proj_bias, mlp_ratio=encoder_mlp_ratio, stochastic_depth_prob=encoder_stochastic_depth_prob,
            norm_layer=encoder_norm_layer, MLP_activation_layer=encoder_MLP_activation_layer, exclude_MLP_after=True)
        hid = int(encoder_dim * encoder_mlp_ratio)
        if encoder_MLP_activation_layer is not None and encoder_MLP_activation_layer not in (nn.GELU, nn.ReLU):
            raise NotImplementedError("MLP activations on the HIP path: nn.GELU (reference default) and nn.ReLU")
        self.encoder_MLP_Key = _make_mlp(encoder_dim, hid, encoder_dropout, encoder_MLP_activation_layer)
        self.encoder_MLP_Scale = _make_mlp(encoder_dim, hid, encoder_dropout, encoder_MLP_activation_layer)
        self.encoder_MLP_Shift = _make_mlp(encoder_dim, hid, encoder_dropout, encoder_MLP_activation_layer)
        self._inference_only_arch = encoder_MLP_activation_layer is nn.ReLU
        # (the reference's init loop at :847-851 never matches an nn.Linear, so these keep nn.Linear's default init)

    def run(self, key: Tensor, scale: Tensor, shift: Tensor, B, H, W):
        blk = self.shared_MHA_without_MLP

        def do_key(key):
            key = blk.run(key, key, [key], [key], B, H, W)[0]                                   # :859-864
            return _mlp_fwd(self, "mlp_key", self.encoder_MLP_Key, key, key)                    # :865

        def do_scale_shift(key, scale, shift):
            # :867-882  both calls use q = k = Key -> one softmax, two value streams; when Scale and Shift
            # are the same tensor (first layer: Scale = Shift = Fs) the attention output is shared too
            s_att, sh_att = blk.run(key, key, [scale, shift], [scale, shift], B, H, W)
            return (_mlp_fwd(self, "mlp_scale", self.encoder_MLP_Scale, s_att, s_att),
                    _mlp_fwd(self, "mlp_shift", self.encoder_MLP_Shift, sh_att, sh_att))

        if self.if_use_processed_Key_in_Scale_and_Shift_calculation:
            key = do_key(key)
            scale, shift = do_scale_shift(key, scale, shift)
        else:
            scale, shift = do_scale_shift(key, scale, shift)
            key = do_key(key)
        return key, scale, shift

    def forward(self, Key: Tensor, Scale: Tensor, Shift: Tensor):
        B, H, W, C = Key.shape
        if _train_mode(self):
            from . import train_forward as TF
            k = Key.reshape(B * H * W, C).float()
            s = k if Scale is Key else Scale.reshape(B * H * W, C).float()
            sh = s if Shift is Scale else (k if Shift is Key else Shift.reshape(B * H * W, C).float())
            return tuple(t.view(B, H, W, C) for t in TF.style_encoder(self, k, s, sh, B, H, W, TF.sd_masks))
        dt = self.compute_dtype
        k = _tokens(Key, dt)
        s = k if Scale is Key else _tokens(Scale, dt)
        sh = s if Shift is Scale else (k if Shift is Key else _tokens(Shift, dt))
        return tuple(_image(t, B, H, W) for t in self.run(k, s, sh, B, H, W))


class StyleDecoder(_HipModule):
    """Reference codes/style_transformer.py:918-1128 (Swin sigma/mu branch)."""

    def __init__(self, decoder_dim: int, decoder_num_heads: int, decoder_window_size: List[int],
                 decoder_shift_size: List[int], decoder_mlp_ratio: float = 4.0, decoder_dropout: float = 0.0,
                 decoder_attention_dropout: float = 0.0, decoder_qkv_bias: bool = True, decoder_proj_bias: bool = True,
                 decoder_stochastic_depth_prob: float = 0.1, decoder_norm_layer: Callable[..., nn.Module] = nn.LayerNorm,
                 decoder_MLP_activation_layer: Optional[Callable[..., nn.Module]] = nn.GELU,
                 decoder_use_instance_norm_with_affine: bool = False,
                 decoder_use_regular_MHA_instead_of_Swin_at_the_end: bool = False,
                 decoder_use_Key_instance_norm_after_linear_transformation: bool = True,
                 decoder_exclude_MLP_after_Fcs_self_MHA: bool = False):
        super().__init__()
        self.decoder_dim, self.decoder_num_heads, self.decoder_mlp_ratio = decoder_dim, decoder_num_heads, decoder_mlp_ratio
        self.decoder_use_instance_norm_with_affine = decoder_use_instance_norm_with_affine
        self.decoder_use_regular_MHA_instead_of_Swin_at_the_end = decoder_use_regular_MHA_instead_of_Swin_at_the_end
        self.decoder_use_Key_instance_norm_after_linear_transformation = decoder_use_Key_instance_norm_after_linear_transformation
        self.decoder_stochastic_depth_prob = decoder_stochastic_depth_prob
        self.MHA_self_attn = StyleSwinTransformerBlock(
            dim=decoder_dim, num_heads=decoder_num_heads, window_size=decoder_window_size, shift_size=decoder_shift_size,
            dropout=decoder_dropout, attention_dropout=decoder_attention_dropout, qkv_bias=decoder_qkv_bias,
            proj_bias=decoder_proj_bias, mlp_ratio=decoder_mlp_ratio, stochastic_depth_prob=decoder_stochastic_depth_prob,
            norm_layer=decoder_norm_layer, MLP_activation_layer=decoder_MLP_activation_layer,
            exclude_MLP_after=decoder_exclude_MLP_after_Fcs_self_MHA)
        if decoder_use_instance_norm_with_affine:                                  # :983-984
            self.instance_norm_Query = nn.InstanceNorm2d(decoder_dim, affine=True)
            self.instance_norm_Key = nn.InstanceNorm2d(decoder_dim, affine=True)
            nq, nk = self.instance_norm_Query, self.instance_norm_Key
        else:
            self.instance_norm = nn.InstanceNorm2d(decoder_dim, affine=False)      # parameter-free, kept for parity of the tree
            nq = nk = self.instance_norm
        if decoder_MLP_activation_layer is not None and decoder_MLP_activation_layer not in (nn.GELU, nn.ReLU):
            raise NotImplementedError("MLP activations on the HIP path: nn.GELU (reference default) and nn.ReLU")
        self.last_MLP = _make_mlp(decoder_dim, int(decoder_dim * decoder_mlp_ratio), decoder_dropout, decoder_MLP_activation_layer)
        self._inference_only_arch = decoder_MLP_activation_layer is nn.ReLU
        if not decoder_use_regular_MHA_instead_of_Swin_at_the_end:
            self.decoder_MHA_for_sigma_and_mu = ShiftedWindowAttention_for_decoder_last_MHA(
                dim=decoder_dim, num_heads=decoder_num_heads, window_size=decoder_window_size, shift_size=decoder_shift_size,
                instance_norm_q=nq, instance_norm_k=nk, dropout=decoder_dropout,
                attention_dropout=decoder_attention_dropout, qkv_bias=decoder_qkv_bias, proj_bias=decoder_proj_bias,
                use_q_proj=False,
                use_Key_instance_norm_after_linear_transformation=decoder_use_Key_instance_norm_after_linear_transformation)
        else:                                                                      # :1034-1041  global single-head attention at the end
            self.linear_transformation_Key = nn.Linear(decoder_dim, decoder_dim)
            self.linear_transformation_Scale = nn.Linear(decoder_dim, decoder_dim)
            self.linear_transformation_Shift = nn.Linear(decoder_dim, decoder_dim)
            self.proj_sigma = nn.Linear(decoder_dim, decoder_dim)
            self.proj_mu = nn.Linear(decoder_dim, decoder_dim)

    REGULAR_MHA_AFFINE_ERROR = ("expected input's size at dim=0 to match num_features ({dim}), but got: {B}.  (the reference hands "
                                "InstanceNorm2d(affine=True) a 3-D [B, C, H*W] tensor in the regular-MHA branch, style_transformer.py:1073-1090)")

    def _image_norm(self, x: Tensor, B: int, H: int, W: int) -> Tensor:
        """The regular-MHA branch hands nn.InstanceNorm2d a 3-D [B, C, H*W] tensor (style_transformer.py:1073-1090): torch treats it as
        one un-batched sample with B channels, i.e. every image is normalised over ALL its channels and tokens jointly (biased
        variance, eps 1e-5).  Per-channel statistics (master_channel_stats_fwd) are combined on the host side of the device ([B,C]
        arithmetic), the normalisation itself is one per-channel affine launch per image."""
        C = x.shape[1]
        T = H * W
        mean, var = ops.channel_stats(x.view(B, H, W, C))
        mu = mean.mean(dim=1)
        v = (var + (mean - mu[:, None]) ** 2).mean(dim=1)
        rstd = torch.rsqrt(v + EPS)
        g = rstd[:, None].expand(B, C).contiguous()
        b = (-mu * rstd)[:, None].expand(B, C).contiguous()
        return torch.cat([ops.channel_affine(x[i * T:(i + 1) * T], g[i], b[i]) for i in range(B)], dim=0)

    def _regular_mha(self, query: Tensor, key: Tensor, scale: Tensor, shift: Tensor, B: int, H: int, W: int) -> Tensor:
        """:1060-1119  sigma, mu = proj(softmax(IN(Query) K^T / sqrt(C)) . V) with ONE head over all H*W tokens of the image; returns
        Query * sigma + mu.  Q.K^T and P.V run on master_linear_fwd per image (K rows / V^T columns zero-padded to a multiple of 64
        tokens), the softmax on master_softmax_rows_fwd."""
        T, C = H * W, query.shape[1]
        if self.decoder_use_instance_norm_with_affine:
            raise ValueError(self.REGULAR_MHA_AFFINE_ERROR.format(dim=C, B=B)) if B != C else NotImplementedError(
                "regular MHA with affine InstanceNorm and batch == dim is not implemented")
        if T % 4:
            raise NotImplementedError("the regular-MHA decoder end needs H*W % 4 == 0 tokens per image")
        lin = lambda t, m, n: ops.linear(t, self._w(n, m.weight), self._b(n, m.bias))
        q_in = self._image_norm(query, B, H, W)
        if self.decoder_use_Key_instance_norm_after_linear_transformation:
            k = self._image_norm(lin(key, self.linear_transformation_Key, "ltk"), B, H, W)
        else:
            k = lin(self._image_norm(key, B, H, W), self.linear_transformation_Key, "ltk")
        vs, vsh = lin(scale, self.linear_transformation_Scale, "lts"), lin(shift, self.linear_transformation_Shift, "ltsh")
        Tp = -(-T // 64) * 64
        kp = torch.zeros((B, Tp, C), dtype=k.dtype, device=k.device); kp[:, :T] = k.view(B, T, C)
        vT = torch.zeros((2, B, C, Tp), dtype=k.dtype, device=k.device)
        vT[0, :, :, :T] = vs.view(B, T, C).transpose(1, 2); vT[1, :, :, :T] = vsh.view(B, T, C).transpose(1, 2)
        att = torch.empty((2, B * T, C), dtype=k.dtype, device=k.device)
        for i in range(B):
            s = ops.linear(q_in[i * T:(i + 1) * T], kp[i])                                     # [T, Tp] logits (padding columns: 0, ignored)
            p = ops.softmax_rows(s, C ** -0.5, n_valid=T)
            ops.linear(p, vT[0, i], out=att[0, i * T:(i + 1) * T]); ops.linear(p, vT[1, i], out=att[1, i * T:(i + 1) * T])
        mu = lin(att[1], self.proj_mu, "pmu")
        return ops.linear(att[0], self._w("psg", self.proj_sigma.weight), self._b("psg", self.proj_sigma.bias), r1=query, r2=mu)

    def run(self, fcs: Tensor, key: Tensor, scale: Tensor, shift: Tensor, B, H, W) -> Tensor:
        T = H * W
        query = self.MHA_self_attn.run(fcs, fcs, [fcs], [fcs], B, H, W)[0]                     # :1047
        if self.decoder_use_regular_MHA_instead_of_Swin_at_the_end:
            query = self._regular_mha(query, key, scale, shift, B, H, W)                       # :1060-1123
        elif self.decoder_use_instance_norm_with_affine:
            nq, nk = self.instance_norm_Query, self.instance_norm_Key
            q_in = ops.channel_affine(ops.instnorm(query, B, T)[0], self._b("inq.w", nq.weight), self._b("inq.b", nq.bias))   # :1056
            k_in = ops.channel_affine(ops.instnorm(key, B, T)[0], self._b("ink.w", nk.weight), self._b("ink.b", nk.bias))     # :1057
            query = self.decoder_MHA_for_sigma_and_mu.sigma_mu(q_in, k_in, scale, shift, B, H, W, fuse_query=query)           # :1059,:1123
        else:
            k_in, _ = ops.instnorm(key, B, T)                                                  # :1057
            # Query is instance-normalised here (:1056) and once more inside the sigma/mu attention (:468): one launch
            query = self.decoder_MHA_for_sigma_and_mu.sigma_mu(query, k_in, scale, shift, B, H, W, fuse_query=query,
                                                               q_norm_repeat=2)                # :1056,:1059,:1123
        return _mlp_fwd(self, "last_mlp", self.last_MLP, query, query)                         # :1125

    def forward(self, Fcs: Tensor, Key: Tensor, Scale: Tensor, Shift: Tensor) -> Tensor:
        B, H, W, C = Fcs.shape
        if _train_mode(self):
            from . import train_forward as TF
            f = lambda t: t.reshape(B * H * W, C).float()
            return TF.style_decoder(self, f(Fcs), f(Key), f(Scale), f(Shift), B, H, W, TF.sd_masks).view(B, H, W, C)
        dt = self.compute_dtype
        return _image(self.run(_tokens(Fcs, dt), _tokens(Key, dt), _tokens(Scale, dt), _tokens(Shift, dt), B, H, W), B, H, W)


class StyleTransformer(_HipModule):
    """Reference codes/style_transformer.py:1133-1245."""

    def __init__(self, encoder_dim: int, decoder_dim: int, encoder_num_heads: int, decoder_num_heads: int,
                 encoder_window_size: List[int], decoder_window_size: List[int], encoder_shift_size: List[int],
                 decoder_shift_size: List[int], encoder_mlp_ratio: float = 4.0, decoder_mlp_ratio: float = 4.0,
                 encoder_dropout: float = 0.0, decoder_dropout: float = 0.0, encoder_attention_dropout: float = 0.0,
                 decoder_attention_dropout: float = 0.0, encoder_qkv_bias: bool = True, decoder_qkv_bias: bool = True,
                 encoder_proj_bias: bool = True, decoder_proj_bias: bool = True, encoder_stochastic_depth_prob: float = 0.1,
                 decoder_stochastic_depth_prob: float = 0.1, encoder_norm_layer: Callable[..., nn.Module] = None,
                 decoder_norm_layer: Callable[..., nn.Module] = nn.LayerNorm,
                 encoder_MLP_activation_layer: Optional[Callable[..., nn.Module]] = nn.GELU,
                 decoder_MLP_activation_layer: Optional[Callable[..., nn.Module]] = nn.GELU,
                 encoder_if_use_processed_Key_in_Scale_and_Shift_calculation: bool = True,
                 decoder_use_instance_norm_with_affine: bool = False,
                 decoder_use_regular_MHA_instead_of_Swin_at_the_end: bool = False,
                 decoder_use_Key_instance_norm_after_linear_transformation: bool = True,
                 decoder_exclude_MLP_after_Fcs_self_MHA: bool = False):
        super().__init__()
        self.encoder = StyleEncoder(
            encoder_dim=encoder_dim, encoder_num_heads=encoder_num_heads, encoder_window_size=encoder_window_size,
            encoder_shift_size=encoder_shift_size, encoder_mlp_ratio=encoder_mlp_ratio, encoder_dropout=encoder_dropout,
            encoder_attention_dropout=encoder_attention_dropout, encoder_qkv_bias=encoder_qkv_bias,
            encoder_proj_bias=encoder_proj_bias, encoder_stochastic_depth_prob=encoder_stochastic_depth_prob,
            encoder_norm_layer=encoder_norm_layer, encoder_MLP_activation_layer=encoder_MLP_activation_layer,
            encoder_if_use_processed_Key_in_Scale_and_Shift_calculation=encoder_if_use_processed_Key_in_Scale_and_Shift_calculation)
        self.decoder = StyleDecoder(
            decoder_dim=decoder_dim, decoder_num_heads=decoder_num_heads, decoder_window_size=decoder_window_size,
            decoder_shift_size=decoder_shift_size, decoder_mlp_ratio=decoder_mlp_ratio, decoder_dropout=decoder_dropout,
            decoder_attention_dropout=decoder_attention_dropout, decoder_qkv_bias=decoder_qkv_bias,
            decoder_proj_bias=decoder_proj_bias, decoder_stochastic_depth_prob=decoder_stochastic_depth_prob,
            decoder_norm_layer=decoder_norm_layer, decoder_MLP_activation_layer=decoder_MLP_activation_layer,
            decoder_use_instance_norm_with_affine=decoder_use_instance_norm_with_affine,
            decoder_use_regular_MHA_instead_of_Swin_at_the_end=decoder_use_regular_MHA_instead_of_Swin_at_the_end,
            decoder_use_Key_instance_norm_after_linear_transformation=decoder_use_Key_instance_norm_after_linear_transformation,
            decoder_exclude_MLP_after_Fcs_self_MHA=decoder_exclude_MLP_after_Fcs_self_MHA)

    def run(self, fc: Tensor, fs: Tensor, k: int, B, H, W) -> Tensor:
        scale = shift = fs                                                     # :1235-1236 (same tensor object)
        for _ in range(k):                                                     # :1239, shared weights
            fs, scale, shift = self.encoder.run(fs, scale, shift, B, H, W)
            fc = self.decoder.run(fc, fs, scale, shift, B, H, W)
        return fc

    def forward(self, Fc: Tensor, Fs: Tensor, k: int = 1) -> Tensor:
        B, H, W, C = Fc.shape
        if _train_mode(self):
            from . import train_forward as TF
            return TF.style_transformer(self, Fc.reshape(B * H * W, C).float(), Fs.reshape(B * H * W, C).float(), k, B, H, W,
                                        TF.sd_masks).view(B, H, W, C)
        dt = self.compute_dtype
        return _image(self.run(_tokens(Fc, dt), _tokens(Fs, dt), k, B, H, W), B, H, W)


# ---------------------------------------------------------------------------------------------------
# image decoder (reference codes/decoder.py)
# ---------------------------------------------------------------------------------------------------
class Decoder(_HipModule):
    CONV_IDX = (0, 3, 5, 7, 9, 12, 14, 17, 19)
    UPSAMPLE_BEFORE = (3, 12, 17)

    def __init__(self, channel_dim: int = 256, initializer: str = "kaiming_normal_"):
        super().__init__()
        assert initializer in ["default", "kaiming_normal_", "kaiming_uniform_", "xavier_normal_", "xavier_uniform_",
                               "orthogonal_"], \
            "Invalid initializer. Please choose one of the following: default, kaiming_normal_, kaiming_uniform_, xavier_normal_, xavier_uniform_, orthogonal_"
        c = channel_dim

        def conv(ci, co):
            return nn.Conv2d(ci, co, (3, 3), padding=(1, 1), padding_mode="reflect")
        up = lambda: nn.Upsample(scale_factor=2, mode="nearest")
        self.decoder = nn.Sequential(                                    # same indices as decoder.py:23-55
            conv(c, c // 2), nn.ReLU(), up(),
            conv(c // 2, c // 2), nn.ReLU(), conv(c // 2, c // 2), nn.ReLU(), conv(c // 2, c // 2), nn.ReLU(),
            conv(c // 2, c // 4), nn.ReLU(), up(),
            conv(c // 4, c // 4), nn.ReLU(), conv(c // 4, c // 8), nn.ReLU(), up(),
            conv(c // 8, c // 8), nn.ReLU(), conv(c // 8, 3))
        init = {"kaiming_normal_": lambda w: nn.init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu"),
                "kaiming_uniform_": lambda w: nn.init.kaiming_uniform_(w, mode="fan_out", nonlinearity="relu"),
                "xavier_normal_": nn.init.xavier_normal_, "xavier_uniform_": nn.init.xavier_uniform_,
                "orthogonal_": nn.init.orthogonal_}.get(initializer)
        for m in self.decoder.modules():                                 # decoder.py:58-73
            if isinstance(m, nn.Conv2d):
                if init is not None:
                    init(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def padded_channels(self, i: int) -> Tuple[int, int]:
        """(Cout, Cin) of conv i as the kernels see them: channel counts that are not multiples of 32 (channel_dim = 192: the 48- and
        24-channel layers) are carried zero-padded to the next multiple - zero weights and biases, so the padding channels stay 0
        through every ReLU and contribute nothing to the next layer.  The last convolution keeps its 3 outputs."""
        conv = self.decoder[i]
        pad = lambda n: (n + 31) // 32 * 32
        return (conv.out_channels if i == self.CONV_IDX[-1] else pad(conv.out_channels)), pad(conv.in_channels)

    def run_nhwc(self, x: Tensor) -> Tensor:
        """x: NHWC [B,H,W,C] in the compute dtype -> fp32 NCHW image [B,3,8H,8W]"""
        last, prev = self.CONV_IDX[-1], self.CONV_IDX[-2]

        def packed(i):
            conv = self.decoder[i]
            co, ci = self.padded_channels(i)

            def build():
                w = conv.weight.detach()
                if (co, ci) != tuple(w.shape[:2]):
                    w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, ci - w.shape[1], 0, co - w.shape[0]))
                if self.split_weights and co > 4:          # the 3-channel last layer runs on the direct fp32 kernel in both fp32 modes
                    return ops.split_weights(w.permute(0, 2, 3, 1).contiguous(), fmt=self.split_weights)
                return ops.convert(w.permute(0, 2, 3, 1).contiguous(), self.compute_dtype)
            return self._derived(f"conv{i}", [conv.weight], build)

        def bias(i):
            conv = self.decoder[i]
            co = self.padded_channels(i)[0]
            if co == conv.out_channels:
                return self._b(f"conv{i}", conv.bias)
            return self._derived(f"b:conv{i}:padded", [conv.bias],
                                 lambda: torch.nn.functional.pad(conv.bias.detach().float(), (0, co - conv.out_channels)).contiguous())
        for i in self.CONV_IDX:
            conv = self.decoder[i]
            if i == prev and prev in self.UPSAMPLE_BEFORE and ops.decoder_tail_supported(x, conv.out_channels, self.split_weights):
                # bf16 and split modes: Upsample + Conv(32->32) + ReLU + Conv(32->3) in one launch, intermediate kept in LDS
                w_last = packed(last) if not self.split_weights else self._derived(
                    f"conv{last}:split", [self.decoder[last].weight],
                    lambda: ops.split_weights(self.decoder[last].weight.detach().permute(0, 2, 3, 1).contiguous(), fmt=self.split_weights))
                return ops.decoder_tail(x, packed(prev), self._b(f"conv{prev}", conv.bias), w_last,
                                        self._b(f"conv{last}", self.decoder[last].bias))
            # split mode: the maps between two x3 convolutions travel in the split layout (the consumer's LDS-DMA pieces copy them as they
            # lie: no split arithmetic, no LDS stores in its loop); the layer in front of the 3-channel last one writes fp32 for the direct kernel
            nxt = self.CONV_IDX[self.CONV_IDX.index(i) + 1] if i != last else None
            out_split = bool(self.split_weights and nxt is not None and nxt != last and os.environ.get("MASTER_X3_SPLIT_ACT", "1") != "0")
            x = ops.conv3x3_reflect(x, packed(i), bias(i), upsample=i in self.UPSAMPLE_BEFORE,
                                    relu=i != last, out_nchw=i == last, out_split=out_split)
        return x

    def forward(self, x: Tensor) -> Tensor:
        """NCHW in / NCHW out like the reference (decoder.py:75-76)"""
        if _train_mode(self):
            from . import train_forward as TF
            return TF.image_decoder(self, x.permute(0, 2, 3, 1).contiguous().float())
        return self.run_nhwc(ops.convert(x.permute(0, 2, 3, 1).contiguous(), self.compute_dtype))


# ---------------------------------------------------------------------------------------------------
# Swin-B features[:4] (torchvision key names: 0.0 conv, 0.2 LN, 1.{0,1} blocks, 2 merge, 3.{0,1} blocks)
# ---------------------------------------------------------------------------------------------------
class _SwinAttention(_WindowAttentionBase):
    def __init__(self, dim: int, num_heads: int, window_size, shift_size):
        super().__init__()
        _check_arch(0.0, 0.0, True, True, nn.GELU, dim, num_heads)
        self.window_size, self.shift_size, self.num_heads = list(window_size), list(shift_size), num_heads
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        self._init_rel_pos(num_heads)

    def attend(self, x: Tensor, residual: Tensor, B, H, W, norm: nn.LayerNorm) -> Tensor:
        """residual + proj(attention(qkv(norm(x))))"""
        C = x.shape[1]
        if residual is x and ops.attn_block_supported(x, self.num_heads):
            plan = window_plan(H, W, self.window_size, self.shift_size, x.device)
            return ops.attn_block(x, self._b("n1w", norm.weight), self._b("n1b", norm.bias), self._w("qkv", self.qkv.weight),
                                  self._b("qkv", self.qkv.bias), self._w("proj", self.proj.weight), self._b("proj", self.proj.bias),
                                  self._bias64(), plan.src, plan.region, B, H * W, self.num_heads, plan.nW, plan.N,
                                  HEAD_DIM ** -0.5, norm.eps)
        qkv = self._ln_linear("qkv", x, norm, [self.qkv.weight], [self.qkv.bias])
        rows = self._derived("rows", [self.qkv.bias], lambda: ops.convert(self.qkv.bias.detach().contiguous(), self.compute_dtype))
        plan = window_plan(H, W, self.window_size, self.shift_size, x.device)
        if ops.window_attn_proj_supported(qkv, self.num_heads, plan.nW, plan.N):
            return ops.window_attn_proj(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], rows[C:2 * C], rows[2 * C:], self._bias64(), plan.src,
                                        plan.region, self._w("proj", self.proj.weight), self._b("proj", self.proj.bias), residual, B, H * W,
                                        self.num_heads, plan.nW, plan.N, HEAD_DIM ** -0.5)
        o = self._core(qkv[:, :C], qkv[:, C:2 * C], [qkv[:, 2 * C:]], rows[C:2 * C], [rows[2 * C:], None], B, H, W)[0]
        return ops.linear(o, self._w("proj", self.proj.weight), self._b("proj", self.proj.bias), r2=residual)


class SwinTransformerBlock(_HipModule):
    def __init__(self, dim: int, num_heads: int, window_size, shift_size, mlp_ratio: float = 4.0, stochastic_depth_prob: float = 0.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=EPS)
        self.attn = _SwinAttention(dim, num_heads, window_size, shift_size)
        self.norm2 = nn.LayerNorm(dim, eps=EPS)
        self.mlp = _make_mlp(dim, int(dim * mlp_ratio), 0.0)
        self.stochastic_depth_prob = stochastic_depth_prob      # torchvision: StochasticDepth(p, "row") on both branches
        _init_mlp(self.mlp)

    def run(self, x: Tensor, B, H, W, masks: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
        """x + SD(attn(norm1 x)); x + SD(mlp(norm2 x)).  `masks` = (attention, mlp) row masks [B] of a train-mode call (the frozen
        encoder still runs its StochasticDepth when the model is in train mode, train_only_inner_loop.py:239); None = identity."""
        if masks is None:
            x = self.attn.attend(x, x, B, H, W, self.norm1)
            return _mlp_fwd(self, "mlp", self.mlp, x, x, norm=self.norm2)
        from .autograd import rowscale_add
        if x.dtype != torch.float32:
            raise NotImplementedError("train-mode StochasticDepth in the encoder runs in the fp32 mode")
        T = H * W
        x = rowscale_add(self.attn.attend(x, None, B, H, W, self.norm1), x, masks[0], B, T)
        return rowscale_add(_mlp_fwd(self, "mlp", self.mlp, x, None, norm=self.norm2), x, masks[1], B, T)


class PatchMerging(_HipModule):
    def __init__(self, dim: int):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim, eps=EPS)

    def run(self, x2d: Tensor, B, H, W) -> Tensor:
        C = x2d.shape[1]
        x = x2d.view(B, H, W, C)
        if H % 2 or W % 2:
            # torchvision PatchMerging pads the map with zeros to even sides before the 2x2 gather (image sides that are multiples of 4
            # but not of 8); the padded copy is plumbing, the gather + LayerNorm + Linear run on it unchanged
            x = torch.nn.functional.pad(x, (0, 0, 0, W % 2, 0, H % 2)).contiguous()
        return ops.patch_merge_linear(x, self._b("nw", self.norm.weight), self._b("nb", self.norm.bias),
                                      self._w("red", self.reduction.weight), self.norm.eps)


class SwinEncoder(nn.Sequential):
    """swin_b.features[:4] with torchvision's child indices, so `torch.load`-ed torchvision state dicts of
    the reference's cut encoder (codes/utils.py:59-102) load by key.  NCHW image -> NHWC [B,H/8,W/8,2E]."""
    compute_dtype = torch.float32

    def __init__(self, embed_dim: int = 128, depths=(2, 2), num_heads=(4, 8), window_size=(7, 7), mlp_ratio: float = 4.0,
                 stochastic_depth_prob: float = 0.5, total_blocks: int = 24):
        """stochastic_depth_prob / total_blocks: torchvision's schedule for the FULL backbone (swin_b: 0.5 over 24 blocks):
        block number n (counted over all stages) gets p = stochastic_depth_prob * n / (total_blocks - 1)."""
        patch = nn.Sequential(nn.Conv2d(3, embed_dim, kernel_size=4, stride=4), nn.Identity(), nn.LayerNorm(embed_dim, eps=EPS))
        counter = [0]

        def stage(dim, nh, depth):
            blocks = []
            for i in range(depth):
                p = stochastic_depth_prob * float(counter[0]) / float(max(total_blocks - 1, 1))
                counter[0] += 1
                blocks.append(SwinTransformerBlock(dim, nh, window_size, [0, 0] if i % 2 == 0 else [w // 2 for w in window_size],
                                                   mlp_ratio, stochastic_depth_prob=p))
            return nn.Sequential(*blocks)
        super().__init__(patch, stage(embed_dim, num_heads[0], depths[0]), PatchMerging(embed_dim),
                         stage(2 * embed_dim, num_heads[1], depths[1]))
        for m in self.modules():                                  # torchvision SwinTransformer init
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def run(self, *imgs: Tensor) -> Tuple[Tensor, int, int, int]:
        """fp32 NCHW image batch(es) -> (token matrix [B*H/8*W/8, 2E] in the compute dtype, B, H/8, W/8).  Several
        batches (content, style) are embedded into consecutive slices of one buffer and encoded as one batch."""
        _, _, H, W = imgs[0].shape
        if H < 4 or W < 4:
            raise RuntimeError(f"Kernel size can't be greater than actual input size: the 4x4 patch embedding needs images of at least 4x4 (got {H}x{W})")
        if H % 4 or W % 4:
            # Conv2d(kernel 4, stride 4, no padding) never reads the trailing H % 4 rows / W % 4 columns (torchvision's patch embedding; the
            # reference accepts such images): crop them - a copy, on this rare path only
            for img in imgs:
                if img.shape[1:] != imgs[0].shape[1:]:
                    raise ValueError("all image batches must have the same size")
            H, W = H - H % 4, W - W % 4
            imgs = tuple(img[:, :, :H, :W] for img in imgs)
        conv, ln = self[0][0], self[0][2]
        B = sum(int(i.shape[0]) for i in imgs)
        dt1 = getattr(self, "stage1_dtype", None) or self.compute_dtype          # fp32 for embed 96 in the bf16 mode (model.set_compute_dtype)
        x = torch.empty((B, H // 4, W // 4, conv.weight.shape[0]), dtype=dt1, device=imgs[0].device)
        b0 = 0
        for img in imgs:
            if img.shape[1:] != imgs[0].shape[1:]:
                raise ValueError("all image batches must have the same size")
            ops.patch_embed(img.contiguous().float(), conv.weight.detach().reshape(conv.weight.shape[0], -1).float().contiguous(),
                            conv.bias.detach().float(), ln.weight.detach().float(), ln.bias.detach().float(),
                            dt1, ln.eps, out=x[b0:b0 + img.shape[0]])
            b0 += img.shape[0]
        # train mode: torchvision's blocks apply StochasticDepth; each image batch (one reference encoder call) draws its masks
        # in call order - all of content's, then all of style's - and the per-block masks of the 2B batch are their concatenation
        masks = {}
        if self.training:
            from . import train_forward as TF
            blocks = [b for b in list(self[1]) + list(self[3]) if b.stochastic_depth_prob > 0.0]
            if len({int(img.shape[0]) for img in imgs}) == 1:          # all masks of this call in two launches instead of two per mask
                TF.sd_masks.prefetch([b.stochastic_depth_prob for _ in imgs for b in blocks for _ in range(2)], int(imgs[0].shape[0]), imgs[0].device)
            per_call = [[(TF.sd_masks.draw(int(img.shape[0]), b.stochastic_depth_prob, img.device),
                          TF.sd_masks.draw(int(img.shape[0]), b.stochastic_depth_prob, img.device)) for b in blocks] for img in imgs]
            for j, b in enumerate(blocks):
                masks[id(b)] = (torch.cat([pc[j][0] for pc in per_call]), torch.cat([pc[j][1] for pc in per_call]))
        h, w = H // 4, W // 4
        x = x.view(B * h * w, -1)
        for blk in self[1]:
            x = blk.run(x, B, h, w, masks.get(id(blk)))
        x = self[2].run(x, B, h, w)
        if x.dtype != self.compute_dtype:
            x = ops.convert(x, self.compute_dtype)
        h, w = (h + 1) // 2, (w + 1) // 2
        for blk in self[3]:
            x = blk.run(x, B, h, w, masks.get(id(blk)))
        return x, B, h, w

    def forward(self, img: Tensor) -> Tensor:
        if any(p.requires_grad for p in self.parameters()) and torch.is_grad_enabled() and self.training:
            raise NotImplementedError("fine-tuning the Swin encoder is not on the HIP path: freeze it (the reference's default, "
                                      "train_only_inner_loop.py:254-257)")
        with torch.no_grad():
            x, B, h, w = self.run(img)
            return _image(x, B, h, w)
