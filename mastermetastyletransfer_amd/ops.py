"""ctypes bindings of libmaster_hip.so (include/master_hip.h) on torch device tensors.

PyTorch is plumbing here: it owns device memory and the stream; every function below hands raw device
pointers and the current HIP stream to the C ABI.  There is NO CPU or torch fallback: a missing
library or a non-GPU tensor raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MASTER_HIP_LIB") or os.path.join(_HERE, "csrc", "libmaster_hip.so")   # override: A/B runs of two builds

F32, BF16, F32X3, F16X3 = 0, 1, 2, 3   # F32X3 / F16X3: fp32 storage, three 16-bit MFMAs per product on bf16 / IEEE-half operand halves
SPLIT_F16 = 0x10                        # OR-ed into the layout argument of master_split_weights: IEEE-half halves
A_SPLIT, OUT_SPLIT = 0x200, 0x400        # OR-ed into F32X3: activation operand / output in the split layout (include/master_hip.h)

ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
K_FAMILIES = ("linear", "conv3x3", "window_attn", "layernorm", "instnorm", "patch_embed", "patch_merge", "convert", "mlp", "decoder_tail", "attn_block",
              "conv_dgrad", "conv_wgrad", "linear_bwd", "window_attn_bwd", "train_misc")

EXPORTS = (
    "master_version", "master_last_error_string", "master_split_weights", "master_linear_fwd", "master_conv3x3_reflect_fwd",
    "master_window_attn_fwd", "master_layernorm_fwd", "master_instnorm_fwd", "master_patch_embed_fwd",
    "master_patch_merge_ln_fwd", "master_patch_merge_linear_fwd", "master_resize_plan", "master_preprocess_fwd", "master_postprocess_fwd", "master_image_to_nhwc_fwd", "master_vgg_conv1_fwd", "master_conv3x3_zero_fwd", "master_maxpool2x2_fwd", "master_channel_stats_fwd", "master_stats_slices", "master_instnorm_distance_fwd", "master_convert", "master_mlp_fwd", "master_decoder_tail_fwd", "master_linear_ln_fwd", "master_attn_block_fwd", "master_window_attn_proj_fwd", "master_linear_group_fwd", "master_prof_enable", "master_prof_reset", "master_prof_read",
    # train step
    "master_linear_act_fwd", "master_linear_res_fwd", "master_linear_dgrad", "master_linear_wgrad", "master_gemm_workspace_floats", "master_colsum",
    "master_colsum_workspace_floats", "master_reduce_slabs", "master_reduce_segments", "master_layernorm_bwd", "master_layernorm_bwd_workspace_floats",
    "master_instnorm_bwd", "master_window_attn_bwd", "master_conv3x3_dgrad", "master_conv3x3_wgrad", "master_relu_mask",
    "master_fold_reflect", "master_maxpool2x2_bwd", "master_nhwc_to_image", "master_rowscale_add", "master_mul_add_fwd",
    "master_mul_add_bwd", "master_colsum_prod", "master_channel_affine_fwd", "master_softmax_rows_fwd", "master_softmax_rows_bwd", "master_loss_tap_bwd", "master_loss_tap_workspace_floats", "master_loss_terms_fwd", "master_conv_weight_pack", "master_conv3x3_cout3_dgrad", "master_conv3x3_cout3_wgrad", "master_conv3x3_cout3_wgrad_workspace_floats", "master_bias_tile_fwd", "master_bias_tile_bwd", "master_adam_chunk", "master_adam_step", "master_adam_step_dev", "master_lerp",
)

_lib = None

# optional algorithmic work counters (bench.py roofline leg): flops / compulsory bytes per kernel family
_COUNT_ON = False
_COUNTS = {}


def count_enable(on: bool):
    global _COUNT_ON
    _COUNT_ON = bool(on)


def count_reset():
    _COUNTS.clear()


def count_read():
    return {k: dict(v) for k, v in _COUNTS.items()}


def _count(family: str, **kw):
    d = _COUNTS.setdefault(family, {})
    for k, v in kw.items():
        d[k] = d.get(k, 0.0) + float(v)


class MasterHipError(RuntimeError):
    pass


def load_library() -> ctypes.CDLL:
    """dlopen the in-tree library; raise (never fall back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MasterHipError(
            f"{LIB_PATH} not found: build it with `python -m mastermetastyletransfer_amd.build` "
            "(hipcc, gfx950). There is no CPU fallback for the stylization path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise MasterHipError(f"{LIB_PATH} does not export {name}")
    lib.master_last_error_string.restype = ctypes.c_char_p
    vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    lib.master_linear_fwd.argtypes = [vp] * 6 + [ci] * 8 + [vp]
    lib.master_split_weights.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_size_t, ci, vp]
    lib.master_conv3x3_reflect_fwd.argtypes = [vp] * 4 + [ci] * 9 + [vp]
    lib.master_window_attn_fwd.argtypes = [vp] * 4 + [ci] * 3 + [vp] * 3 + [ci] * 2 + [vp] * 3 + [vp] * 2 + [ci] * 6 + [cf, ci, vp]
    lib.master_layernorm_fwd.argtypes = [vp] * 4 + [ci, ci, cf, ci, vp]
    lib.master_instnorm_fwd.argtypes = [vp] * 4 + [ci] * 4 + [cf, ci, ci, vp]
    lib.master_patch_embed_fwd.argtypes = [vp] * 6 + [ci] * 4 + [cf, ci, vp]
    lib.master_patch_merge_ln_fwd.argtypes = [vp] * 4 + [ci] * 4 + [cf, ci, vp]
    lib.master_patch_merge_linear_fwd.argtypes = [vp] * 5 + [ci] * 5 + [cf, ci, vp]
    lib.master_convert.argtypes = [vp, vp, ctypes.c_size_t, ci, ci, vp]
    lib.master_resize_plan.argtypes = [ci, ci, vp, vp, ctypes.POINTER(ci)]
    lib.master_preprocess_fwd.argtypes = [vp, ci, ci, ci, vp, ci, ci, vp, vp, ci, vp, vp, ci, vp, vp, vp, vp]
    lib.master_postprocess_fwd.argtypes = [vp, vp, ci, ci, ci, vp]
    lib.master_image_to_nhwc_fwd.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
    lib.master_vgg_conv1_fwd.argtypes = [vp] * 4 + [ci] * 5 + [vp]
    lib.master_conv3x3_zero_fwd.argtypes = [vp] * 4 + [ci] * 7 + [vp]
    lib.master_maxpool2x2_fwd.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
    lib.master_channel_stats_fwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp]
    lib.master_stats_slices.argtypes = [ci]
    lib.master_instnorm_distance_fwd.argtypes = [vp] * 4 + [ci] * 4 + [cf, ci, ci, vp]
    lib.master_mlp_fwd.argtypes = [vp] * 9 + [ci] * 6 + [cf, ci, vp]
    lib.master_linear_ln_fwd.argtypes = [vp] * 8 + [ci] * 6 + [cf, ci, vp]
    lib.master_attn_block_fwd.argtypes = [vp] * 11 + [ci] * 6 + [cf, cf, ci, vp]
    lib.master_decoder_tail_fwd.argtypes = [vp] * 6 + [ci] * 4 + [vp]
    lib.master_linear_group_fwd.argtypes = [ci] + [vp] * 9 + [ci, ci, cf, ci, vp]
    lib.master_window_attn_proj_fwd.argtypes = [vp] * 3 + [ci] * 3 + [vp] * 2 + [ci] * 2 + [vp] * 6 + [ci, vp, ci] + [ci] * 5 + [cf, ci, vp]
    lib.master_prof_read.argtypes = [ci, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double)]
    # ---- train step ----
    cz, cll = ctypes.c_size_t, ctypes.c_longlong
    lib.master_linear_act_fwd.argtypes = [vp] * 5 + [ci] * 7 + [vp]
    lib.master_linear_dgrad.argtypes = [vp] * 6 + [ci] * 9 + [vp]
    lib.master_linear_res_fwd.argtypes = [vp] * 5 + [ci] + [vp, vp] + [ci] * 8 + [vp]
    lib.master_gemm_workspace_floats.argtypes = [ci, ci]; lib.master_gemm_workspace_floats.restype = cz
    lib.master_linear_wgrad.argtypes = [vp] * 5 + [ci] * 6 + [vp, cz, ci, vp]
    lib.master_colsum_workspace_floats.argtypes = [ci, ci]; lib.master_colsum_workspace_floats.restype = cz
    lib.master_colsum.argtypes = [vp] * 3 + [ci] * 4 + [vp]
    lib.master_reduce_slabs.argtypes = [vp, vp, cz, cz, ci, vp]
    lib.master_reduce_segments.argtypes = [vp, ci, vp]
    lib.master_conv_weight_pack.argtypes = [vp] * 3 + [ci] * 3 + [vp]
    lib.master_conv3x3_cout3_dgrad.argtypes = [vp] * 3 + [ci] * 4 + [vp]
    lib.master_conv3x3_cout3_wgrad.argtypes = [vp] * 5 + [cz] + [ci] * 4 + [vp]
    lib.master_conv3x3_cout3_wgrad_workspace_floats.argtypes = [ci] * 3
    lib.master_conv3x3_cout3_wgrad_workspace_floats.restype = cz
    lib.master_bias_tile_fwd.argtypes = [vp] * 3 + [ci, ci, vp]
    lib.master_bias_tile_bwd.argtypes = [vp] * 3 + [ci] * 3 + [vp]
    lib.master_colsum_prod.argtypes = [vp] * 4 + [ci] * 4 + [vp]
    lib.master_channel_affine_fwd.argtypes = [vp] * 4 + [ci] * 3 + [vp]
    lib.master_softmax_rows_fwd.argtypes = [vp] * 2 + [ci] * 3 + [cf, ci, vp]
    lib.master_softmax_rows_bwd.argtypes = [vp] * 3 + [ci] * 3 + [cf, ci, vp]
    lib.master_layernorm_bwd_workspace_floats.argtypes = [ci, ci]; lib.master_layernorm_bwd_workspace_floats.restype = cz
    lib.master_layernorm_bwd.argtypes = [vp] * 7 + [ci, ci, cf, ci, vp]
    lib.master_instnorm_bwd.argtypes = [vp] * 6 + [ci] * 4 + [cf, ci, vp]
    lib.master_window_attn_bwd.argtypes = ([vp] * 4 + [ci] * 3 + [vp] * 3 + [ci] * 2 + [vp] * 4 + [vp] * 2 + [ci] + [vp] * 4 + [ci] * 3
                                           + [vp] * 4 + [ci] * 6 + [cf, ci, vp])
    lib.master_conv3x3_dgrad.argtypes = [vp] * 5 + [ci] * 7 + [vp]
    lib.master_conv3x3_wgrad.argtypes = [vp] * 4 + [ci] * 8 + [vp, cz, ci, vp]
    lib.master_relu_mask.argtypes = [vp] * 3 + [ci] * 6 + [vp]
    lib.master_fold_reflect.argtypes = [vp] * 2 + [ci] * 6 + [vp]
    lib.master_maxpool2x2_bwd.argtypes = [vp] * 3 + [ci] * 5 + [vp]
    lib.master_nhwc_to_image.argtypes = [vp] * 2 + [ci] * 4 + [vp]
    lib.master_rowscale_add.argtypes = [vp] * 4 + [ci] * 4 + [vp]
    lib.master_mul_add_fwd.argtypes = [vp] * 4 + [cz, ci, vp]
    lib.master_mul_add_bwd.argtypes = [vp] * 5 + [cz, ci, vp]
    lib.master_loss_tap_workspace_floats.argtypes = [ci, ci, ci]; lib.master_loss_tap_workspace_floats.restype = cz
    lib.master_loss_tap_bwd.argtypes = [vp] * 7 + [ci] * 3 + [cf, vp, vp, ci, ci, ci, vp]
    lib.master_loss_terms_fwd.argtypes = [vp] * 3 + [ci] + [vp] * 3 + [ci] * 5 + [vp]
    lib.master_adam_step.argtypes = [vp, ci, cll] + [cf] * 5 + [ci, vp]
    lib.master_adam_step_dev.argtypes = [vp, ci, cll, vp, vp, vp]
    lib.master_lerp.argtypes = [vp, ci, cll, cf, vp]
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = load_library().master_last_error_string().decode()
        if rc == -2:
            raise NotImplementedError(f"{what}: {msg}")
        raise MasterHipError(f"{what} failed (status {rc}): {msg}")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    # every op runs under @_on_device, so the current device is the operands' device here
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_device(fn):
    """Run `fn` with the operands' GPU as the current device: the launch goes to THAT device's current stream (not device 0's)
    and the per-device kernel attributes are set on the right device.  Operands on different GPUs are rejected."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        dev = None
        flat = [t for a in args for t in (a if isinstance(a, (list, tuple)) else (a,))]
        for a in flat:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if dev is None:
                    dev = a.device
                elif a.device != dev:
                    raise MasterHipError(f"{fn.__name__}: operands live on different GPUs ({dev} and {a.device})")
        for a in kwargs.values():
            if isinstance(a, torch.Tensor) and a.is_cuda and dev is not None and a.device != dev:
                raise MasterHipError(f"{fn.__name__}: operands live on different GPUs ({dev} and {a.device})")
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapper


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise MasterHipError(f"unsupported element type {t.dtype}")


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise MasterHipError("the stylization path runs on the GPU only (got a CPU tensor); there is no CPU fallback")


_SPLIT_DTYPES = (torch.bfloat16, torch.float16)


def is_split(w: torch.Tensor) -> bool:
    """True for a weight in the split layout of the F32X3 / F16X3 modes: [rows, K/32, 2, 32] bf16 or half (split_weights)"""
    return w.dtype in _SPLIT_DTYPES and w.dim() == 4 and w.shape[2] == 2 and w.shape[3] == 32


def is_split_act(x: torch.Tensor) -> bool:
    """True for an NHWC activation map in the split layout [B, H, W, C/32, 2, 32] bf16 or half (written by a convolution with out_split=True)"""
    return x.dtype in _SPLIT_DTYPES and x.dim() == 6 and x.shape[4] == 2 and x.shape[5] == 32


def x3_code(w: torch.Tensor) -> int:
    """the C ABI's dtype code of a split operand: bf16 halves -> MASTER_F32X3, IEEE-half halves -> MASTER_F16X3"""
    return F16X3 if w.dtype == torch.float16 else F32X3


def split_torch_dtype(fmt) -> torch.dtype:
    """"bf16" (or True) / "f16" -> the torch dtype that tags a split tensor of that format"""
    if fmt in (True, "bf16"):
        return torch.bfloat16
    if fmt == "f16":
        return torch.float16
    raise ValueError(f"split format must be 'bf16' or 'f16' (got {fmt!r})")


@_on_device
def split_weights(w: torch.Tensor, layout: int = 0, fmt="bf16") -> torch.Tensor:
    """fp32 [rows, K] (or a packed conv weight [Cout,3,3,Cin], flattened) -> bf16 [rows, K/32, 2, 32]: per 32-wide k chunk the 32 bf16
    hi = bf16(w) then the 32 bf16 lo = bf16(w - hi) - what master_linear_fwd / master_conv3x3_reflect_fwd take as `w` with MASTER_F32X3.
    fmt="f16": IEEE-half halves (torch.float16 tensor of the same shape), the weight operand of MASTER_F16X3.
    layout=1: k order 0-3, 8-11, 4-7, 12-15 inside every group of 16 (the second Linear of the fused MLP, master_mlp_fwd)"""
    _dev(w)
    w = w.detach().float().contiguous()
    rows = w.shape[0]
    K = w.numel() // rows
    if K % 32:
        raise MasterHipError(f"split_weights: K = {K} is not a multiple of 32")
    dt = split_torch_dtype(fmt)
    if dt == torch.float16 and not torch.cuda.is_current_stream_capturing():
        # weights are split once per parameter version (derived-weight cache), outside any graph capture: the one place where a range
        # check costs nothing.  A weight beyond half's range would become inf in every product it takes part in.
        wmax = float(w.abs().max()) if w.numel() else 0.0
        if not wmax <= 65504.0:
            raise MasterHipError(f"split_weights(fmt='f16'): max |w| = {wmax:g} is outside IEEE half's range (65 504); use the bf16 halves "
                                 "(\"bf16x3\") for this tensor")
    out = torch.empty((rows, K // 32, 2, 32), dtype=dt, device=w.device)
    _check(load_library().master_split_weights(_p(w), _p(out), rows, K, layout | (SPLIT_F16 if dt == torch.float16 else 0), _stream()),
           "master_split_weights")
    return out


@_on_device
def linear(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, r1: Optional[torch.Tensor] = None,
           r2: Optional[torch.Tensor] = None, act: int = ACT_NONE, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M,N] = r1 * act(a @ w.T + bias) + r2 ; a is [M,K] (last dim contiguous, row stride free)."""
    _dev(a, w)
    M, K = a.shape
    N = w.shape[0]
    x3 = is_split(w)
    assert (w.shape[1] * 32 == K and a.dtype == torch.float32 if x3 else w.shape[1] == K) and a.stride(1) == 1 and w.is_contiguous()
    if out is None:
        out = torch.empty((M, N), dtype=a.dtype, device=a.device)
    ldr = 0
    for r in (r1, r2):
        if r is not None:
            assert r.shape == (M, N) and r.stride(1) == 1 and r.dtype == a.dtype
            ldr = r.stride(0)
    if r1 is not None and r2 is not None:
        assert r1.stride(0) == r2.stride(0)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
    if _COUNT_ON:
        es = a.element_size()
        _count("linear", flops=2.0 * M * N * K,
               bytes=es * (M * K + N * K + M * N * (1 + (r1 is not None) + (r2 is not None))))
    _check(load_library().master_linear_fwd(_p(a), _p(w), _p(bias), _p(r1), _p(r2), _p(out), M, N, K, a.stride(0),
                                             out.stride(0), ldr, act, x3_code(w) if x3 else _dt(a), _stream()), "master_linear_fwd")
    return out


@_on_device
def linear_ln(x: torch.Tensor, ln_gamma: torch.Tensor, ln_beta: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
              r2: Optional[torch.Tensor] = None, eps: float = 1e-5) -> torch.Tensor:
    """(LayerNorm(x) @ w.T + bias) + r2.  bf16 mode: one launch when the shape is covered; otherwise LayerNorm, then Linear."""
    M, K = x.shape
    N = w.shape[0]
    if x.dtype == torch.bfloat16 and K in (128, 256, 512) and N % 128 == 0 and N <= 1024 and (r2 is None or N <= 256) and x.stride(1) == 1:
        _dev(x, w)
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
        if _COUNT_ON:
            _count("linear", flops=2.0 * M * N * K, bytes=x.element_size() * (M * K + N * K + M * N * (1 + (r2 is not None))))
        _check(load_library().master_linear_ln_fwd(_p(x), _p(ln_gamma), _p(ln_beta), _p(w), _p(bias), None, _p(r2), _p(out), M, N, K,
                                                    x.stride(0), out.stride(0), r2.stride(0) if r2 is not None else 0, eps,
                                                    _dt(x), _stream()), "master_linear_ln_fwd")
        return out
    return linear(layernorm(x, ln_gamma, ln_beta, eps), w, bias, r2=r2)


@_on_device
def linear_group(xs: Sequence[torch.Tensor], ws: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]]) -> list:
    """[x_i @ w_i.T + bias_i] for up to three problems with the same M and K in ONE launch (bf16, token-stationary kernel); falls back to
    one linear() per problem when the shapes are outside it."""
    n = len(xs)
    M, K = xs[0].shape
    grouped = (1 < n <= 3 and xs[0].dtype == torch.bfloat16 and K in (128, 256) and os.environ.get("MASTER_LINEAR_GROUP", "1") != "0"
               and all(x.shape == (M, K) and x.stride(1) == 1 and x.dtype == torch.bfloat16 for x in xs)
               and all(w.shape[1] == K and w.shape[0] % 128 == 0 and w.shape[0] <= 1024 and w.is_contiguous() for w in ws))
    if not grouped:
        return [linear(x, w, b) for x, w, b in zip(xs, ws, biases)]
    _dev(*xs, *ws)
    outs = [torch.empty((M, w.shape[0]), dtype=xs[0].dtype, device=xs[0].device) for w in ws]
    arr = lambda ts: (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])
    ints = lambda vs: (ctypes.c_int * n)(*vs)
    if _COUNT_ON:
        for x, w in zip(xs, ws):
            _count("linear", flops=2.0 * M * w.shape[0] * K, bytes=x.element_size() * (M * K + w.numel() + M * w.shape[0]))
    _check(load_library().master_linear_group_fwd(n, arr(xs), arr(ws), arr(biases), arr(outs), ints([w.shape[0] for w in ws]),
                                                   ints([x.stride(0) for x in xs]), ints([o.stride(0) for o in outs]), None, None, M, K,
                                                   1e-5, _dt(xs[0]), _stream()), "master_linear_group_fwd")
    return outs


def mlp_fused_supported(x: torch.Tensor, hidden: int, split: bool = False) -> bool:
    if split:          # F32X3: fp32 tokens, split weights (mlp_x3.hip); MASTER_MLP_X3=0 falls back to two GEMM launches (A/B runs)
        return x.dtype == torch.float32 and x.shape[1] in (128, 256) and hidden % 64 == 0 and os.environ.get("MASTER_MLP_X3", "1") != "0"
    return x.dtype == torch.bfloat16 and x.shape[1] in (128, 256) and hidden % 128 == 0


@_on_device
def mlp(x: torch.Tensor, w1, b1, w2, b2, res: Optional[torch.Tensor] = None, ln_gamma=None, ln_beta=None,
        eps: float = 1e-5) -> torch.Tensor:
    """res + fc2(gelu(fc1(LN?(x)))) in one launch (bf16 mode; F32X3 mode with w1 = split_weights(W1), w2 = split_weights(W2, layout=1))."""
    _dev(x, w1, w2)
    M, C = x.shape
    HID = w1.shape[0]
    x3 = is_split(w1)
    assert x.stride(1) == 1 and w1.is_contiguous() and w2.is_contiguous()
    assert (is_split(w2) and w1.shape[1] * 32 == C and w2.shape[:2] == (C, HID // 32) and x.dtype == torch.float32) if x3 else \
        (w1.shape == (HID, C) and w2.shape == (C, HID))
    assert b1.dtype == torch.float32 and b2.dtype == torch.float32
    out = torch.empty((M, C), dtype=x.dtype, device=x.device)
    if _COUNT_ON:
        # compulsory bytes: x once, out once, the weights once; a residual that IS x (every call site of the path) is not a second read
        extra_res = res is not None and res.data_ptr() != x.data_ptr()
        _count("mlp", flops=4.0 * M * C * HID, bytes=x.element_size() * (M * C * (2 + extra_res) + 2 * C * HID))
    _check(load_library().master_mlp_fwd(_p(x), _p(ln_gamma), _p(ln_beta), _p(w1), _p(b1), _p(w2), _p(b2), _p(res), _p(out),
                                          M, C, HID, x.stride(0), res.stride(0) if res is not None else 0, out.stride(0),
                                          eps, x3_code(w1) if x3 else _dt(x), _stream()), "master_mlp_fwd")
    return out


@_on_device
def conv3x3_reflect(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, upsample: bool, relu: bool,
                    out_nchw: bool = False, out_split: bool = False) -> torch.Tensor:
    """x NHWC [B,H,W,Cin]; w_packed [Cout,3,3,Cin]; returns NHWC, or fp32 NCHW when out_nchw.
    F32X3 mode (w_packed = split_weights(...)): x may be an activation map in the split layout (is_split_act: the output of a previous
    call with out_split=True), and out_split=True writes this call's output in it ([B,Ho,Wo,Cout/32,2,32] bf16, Cout % 32 == 0)."""
    _dev(x, w_packed)
    a_split = is_split_act(x)
    B, H, W = x.shape[:3]
    Cin = x.shape[3] * 32 if a_split else x.shape[3]
    Cout = w_packed.shape[0]
    x3 = is_split(w_packed)
    assert x.is_contiguous() and w_packed.is_contiguous() and bias.dtype == torch.float32
    assert not x3 or ((x.dtype == w_packed.dtype if a_split else x.dtype == torch.float32) and w_packed.shape[1] * 32 == 9 * Cin)
    assert x3 or not (a_split or out_split), "split activations belong to the F32X3 mode"
    Ho, Wo = (H * 2, W * 2) if upsample else (H, W)
    if out_nchw:
        out = torch.empty((B, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    elif out_split:
        assert Cout % 32 == 0
        out = torch.empty((B, Ho, Wo, Cout // 32, 2, 32), dtype=w_packed.dtype, device=x.device)      # halves in the weights' format
    else:
        out = torch.empty((B, Ho, Wo, Cout), dtype=torch.float32 if a_split else x.dtype, device=x.device)
    if _COUNT_ON:
        _count("conv3x3", flops=2.0 * B * Ho * Wo * Cout * 9 * Cin,
               bytes=x.element_size() * (x.numel() + w_packed.numel()) + out.element_size() * out.numel())
    dt = (x3_code(w_packed) | (A_SPLIT if a_split else 0) | (OUT_SPLIT if out_split else 0)) if x3 else _dt(x)
    _check(load_library().master_conv3x3_reflect_fwd(_p(x), _p(w_packed), _p(bias), _p(out), B, H, W, Cin, Cout,
                                                      int(upsample), int(relu), int(out_nchw), dt, _stream()),
           "master_conv3x3_reflect_fwd")
    return out


def attn_block_supported(x: torch.Tensor, heads: int) -> bool:
    return x.dtype == torch.bfloat16 and x.shape[1] == 128 and heads == 4 and x.is_contiguous()


@_on_device
def attn_block(x, ln_gamma, ln_beta, wqkv, bqkv, wproj, bproj, bias64, src, region, B: int, T: int, heads: int, nW: int, N: int,
               scale: float, eps: float = 1e-5) -> torch.Tensor:
    """x + proj(window_attention(qkv(LN?(x)))) in one launch (bf16, C = 128, 4 heads)."""
    _dev(x, wqkv, wproj)
    M, C = x.shape
    assert M == B * T and wqkv.shape == (3 * C, C) and wproj.shape == (C, C) and wqkv.is_contiguous() and wproj.is_contiguous()
    out = torch.empty_like(x)
    if _COUNT_ON:
        # algorithmic work = what the reference computes: the projections on the N (= 49) tokens of a window (642 MFLOP per
        # image and block at C = 256 / 25 windows), not on the kernel's 64-slot tiles
        _count("attn_block", flops=2.0 * B * nW * N * C * 4 * C + 4.0 * B * nW * heads * N * N * 32, bytes=4.0 * x.numel())
    _check(load_library().master_attn_block_fwd(_p(x), _p(ln_gamma), _p(ln_beta), _p(wqkv), _p(bqkv), _p(wproj), _p(bproj),
                                                 _p(bias64), _p(src), _p(region), _p(out), B, T, C, heads, nW, N, eps,
                                                 float(scale), _dt(x), _stream()), "master_attn_block_fwd")
    return out


def decoder_tail_supported(x: torch.Tensor, cout_mid: int, split: bool = False) -> bool:
    if split:                     # F32X3 mode: fp32 NHWC [B,H,W,32] or a map in the split layout [B,H,W,1,2,32]; 8 x 32 output tiles
        ok = (is_split_act(x) and x.shape[3] == 1) or (x.dim() == 4 and x.dtype == torch.float32 and x.shape[3] == 32)
        return ok and cout_mid == 32 and (2 * x.shape[1]) % 8 == 0 and (2 * x.shape[2]) % 32 == 0 and os.environ.get("MASTER_X3_TAIL", "1") != "0"
    if x.dim() != 4:
        return False
    B, H, W, C = x.shape
    return x.dtype == torch.bfloat16 and C == 32 and cout_mid == 32 and (2 * H) % 16 == 0 and (2 * W) % 32 == 0


@_on_device
def decoder_tail(x: torch.Tensor, w8, b8, w9, b9) -> torch.Tensor:
    """x NHWC [B,Hs,Ws,32] -> fp32 NCHW [B,3,2Hs,2Ws]: upsample, conv 32->32 + ReLU, conv 32->3 in one launch.  bf16 mode: x, w8, w9 bf16;
    F32X3 mode: x fp32 or in the split layout, w8 / w9 = split_weights of the packed [32,3,3,32] / [3,3,3,32] weights."""
    _dev(x, w8, w9)
    B, H, W = x.shape[:3]
    x3 = is_split(w8)
    assert x.is_contiguous() and w8.is_contiguous() and w9.is_contiguous()
    if x3:
        assert is_split(w9) and w8.shape[:2] == (32, 9) and w9.shape[:2] == (3, 9)
        dt = x3_code(w8) | (A_SPLIT if is_split_act(x) else 0)
    else:
        assert w8.shape == (32, 3, 3, 32) and w9.shape == (3, 3, 3, 32)
        dt = _dt(x)
    out = torch.empty((B, 3, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    if _COUNT_ON:
        _count("decoder_tail", flops=2.0 * B * 4 * H * W * 9 * 32 * (32 + 3), bytes=(4.0 if x3 else 2.0) * B * H * W * 32 + 4.0 * out.numel())
    _check(load_library().master_decoder_tail_fwd(_p(x), _p(w8), _p(b8), _p(w9), _p(b9), _p(out), B, H, W, dt, _stream()),
           "master_decoder_tail_fwd")
    return out


@_on_device
def window_attn(q, k, v0, v1, k_pad, v0_pad, v1_pad, bias64, src, region, B: int, T: int, heads: int, nW: int, N: int,
                scale: float, x3=False):
    """q/k/v*: [B*T, heads*32] views (row stride free).  Returns out0 (and out1 when v1 is given), [B*T, heads*32].
    x3 (fp32 tensors; True / "bf16" / "f16"): both products as three 16-bit MFMAs on operands split in registers into bf16 or IEEE-half
    halves (the F32X3 / F16X3 fast parity modes)."""
    _dev(q, k, v0)
    C = heads * 32
    out0 = torch.empty((B * T, C), dtype=q.dtype, device=q.device)
    out1 = torch.empty_like(out0) if v1 is not None else None
    assert v0.stride(0) == (v1.stride(0) if v1 is not None else v0.stride(0))
    kbs = k_pad.stride(0) if k_pad.dim() == 2 else 0
    vbs = v0_pad.stride(0) if v0_pad.dim() == 2 else 0
    if _COUNT_ON:
        nv = 1 if v1 is None else 2
        _count("window_attn", flops=2.0 * B * nW * heads * N * N * 32 * (1 + nv), qk_flops=2.0 * B * nW * heads * N * N * 32,
               bytes=q.element_size() * B * T * C * (2 + 2 * nv))
    _check(load_library().master_window_attn_fwd(
        _p(q), _p(k), _p(v0), _p(v1), q.stride(0), k.stride(0), v0.stride(0), _p(k_pad), _p(v0_pad), _p(v1_pad),
        kbs, vbs, _p(bias64), _p(src), _p(region), _p(out0), _p(out1), C, B, T, heads, nW, N, float(scale),
        (F16X3 if x3 == "f16" else F32X3) if (x3 and q.dtype == torch.float32 and os.environ.get("MASTER_ATTN_X3", "1") != "0") else _dt(q),
        _stream()), "master_window_attn_fwd")
    return out0, out1


def window_attn_proj_supported(q: torch.Tensor, heads: int, nW: int, N: int = 49) -> bool:
    """the fused attention + proj + residual launch: bf16, 8 heads (C = 256), windows of up to 49 tokens, tables of the geometry next
    to three 32-KB tiles and the 50-KB output stage in LDS (up to 45 windows per image: 32x32 maps have 25).  Measured no faster than
    attention + Linear as two launches (window_attn_proj_bf16.hip, "MEASURED OUTCOME"), so it is opt-in: MASTER_ATTN_PROJ=1."""
    return (q.dtype == torch.bfloat16 and heads == 8 and N <= 49 and 3 * 32768 + 51200 + nW * 64 * 5 <= 160 * 1024
            and os.environ.get("MASTER_ATTN_PROJ", "0") == "1")


@_on_device
def window_attn_proj(q, k, v, k_pad, v_pad, bias64, src, region, wproj, bproj, res, B: int, T: int, heads: int, nW: int, N: int,
                     scale: float) -> torch.Tensor:
    """res + proj(window_attention(q, k, v)) in one launch (bf16, 8 heads).  q/k/v: [B*T, 256] views (row stride free)."""
    _dev(q, k, v, wproj)
    C = heads * 32
    out = torch.empty((B * T, C), dtype=q.dtype, device=q.device)
    assert wproj.shape == (C, C) and wproj.is_contiguous() and bproj.dtype == torch.float32
    assert res is None or (res.shape == (B * T, C) and res.stride(1) == 1 and res.dtype == q.dtype)
    kbs = k_pad.stride(0) if k_pad.dim() == 2 else 0
    vbs = v_pad.stride(0) if v_pad.dim() == 2 else 0
    if _COUNT_ON:
        _count("window_attn", flops=2.0 * B * nW * heads * N * N * 32 * 2 + 2.0 * B * T * C * C, qk_flops=2.0 * B * nW * heads * N * N * 32,
               bytes=q.element_size() * (B * T * C * (4 + (res is not None)) + C * C))
    _check(load_library().master_window_attn_proj_fwd(
        _p(q), _p(k), _p(v), q.stride(0), k.stride(0), v.stride(0), _p(k_pad), _p(v_pad), kbs, vbs, _p(bias64), _p(src), _p(region),
        _p(wproj), _p(bproj), _p(res), res.stride(0) if res is not None else 0, _p(out), out.stride(0), B, T, heads, nW, N,
        float(scale), _dt(q), _stream()), "master_window_attn_proj_fwd")
    return out


@_on_device
def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    _dev(x)
    M, C = x.shape
    assert x.is_contiguous()
    out = torch.empty_like(x)
    if _COUNT_ON:
        _count("layernorm", bytes=2.0 * x.element_size() * x.numel())
    _check(load_library().master_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(out), M, C, eps, _dt(x), _stream()),
           "master_layernorm_fwd")
    return out


@_on_device
def instnorm(x: torch.Tensor, B: int, T: int, pad_row: Optional[torch.Tensor] = None, n_pad: int = 0, eps: float = 1e-5,
             repeat: int = 1):
    """x [B*T, C].  Returns (normalised x, normalised pad rows [B,C] or None).  repeat=2: InstanceNorm applied twice."""
    _dev(x)
    C = x.shape[1]
    assert x.is_contiguous() and x.shape[0] == B * T
    out = torch.empty_like(x)
    pad_out = torch.empty((B, C), dtype=x.dtype, device=x.device) if n_pad > 0 else None
    if _COUNT_ON:
        _count("instnorm", bytes=2.0 * x.element_size() * x.numel())
    _check(load_library().master_instnorm_fwd(_p(x), _p(out), _p(pad_row), _p(pad_out), B, T, C, n_pad, eps, repeat, _dt(x),
                                               _stream()), "master_instnorm_fwd")
    return out, pad_out


@_on_device
def channel_affine(x: torch.Tensor, gamma: torch.Tensor, beta: Optional[torch.Tensor]) -> torch.Tensor:
    """x[m,c] * gamma[c] + beta[c]: the affine half of nn.InstanceNorm2d(dim, affine=True); gamma/beta fp32 [C]"""
    _dev(x)
    M, C = x.shape
    assert x.is_contiguous() and gamma.dtype == torch.float32
    out = torch.empty_like(x)
    _check(load_library().master_channel_affine_fwd(_p(x), _p(gamma), _p(beta), _p(out), M, C, _dt(x), _stream()), "master_channel_affine_fwd")
    return out


@_on_device
def softmax_rows(x: torch.Tensor, scale: float = 1.0, n_valid: Optional[int] = None) -> torch.Tensor:
    """softmax(scale * x) over the first n_valid columns of every row of a [M, ld] matrix; the other columns (row padding) become 0"""
    _dev(x)
    M, ld = x.shape
    assert x.is_contiguous()
    out = torch.empty_like(x)
    _check(load_library().master_softmax_rows_fwd(_p(x), _p(out), M, ld if n_valid is None else n_valid, ld, float(scale), _dt(x), _stream()),
           "master_softmax_rows_fwd")
    return out


@_on_device
def patch_embed(img: torch.Tensor, w: torch.Tensor, bias, gamma, beta, dtype: torch.dtype, eps: float = 1e-5,
                out: Optional[torch.Tensor] = None):
    """img fp32 NCHW [B,3,H,W] -> NHWC [B,H/4,W/4,E] with LayerNorm(E) applied (into `out` when given)."""
    _dev(img, w)
    B, _, H, W = img.shape
    E = w.shape[0]
    assert img.dtype == torch.float32 and img.is_contiguous()
    if out is None:
        out = torch.empty((B, H // 4, W // 4, E), dtype=dtype, device=img.device)
    assert out.shape == (B, H // 4, W // 4, E) and out.dtype == dtype and out.is_contiguous()
    _check(load_library().master_patch_embed_fwd(_p(img), _p(w), _p(bias), _p(gamma), _p(beta), _p(out), B, H, W, E, eps,
                                                  F32 if dtype == torch.float32 else BF16, _stream()),
           "master_patch_embed_fwd")
    return out


@_on_device
def patch_merge_ln(x: torch.Tensor, gamma, beta, eps: float = 1e-5) -> torch.Tensor:
    _dev(x)
    B, H, W, C = x.shape
    assert x.is_contiguous()
    out = torch.empty((B, H // 2, W // 2, 4 * C), dtype=x.dtype, device=x.device)
    _check(load_library().master_patch_merge_ln_fwd(_p(x), _p(gamma), _p(beta), _p(out), B, H, W, C, eps, _dt(x),
                                                     _stream()), "master_patch_merge_ln_fwd")
    return out


@_on_device
def patch_merge_linear(x: torch.Tensor, gamma, beta, w: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """torchvision PatchMerging: Linear(4C, N, no bias)(LayerNorm(2x2 gather(x))); x NHWC [B,H,W,C] -> [B*H/2*W/2, N].
    bf16 mode: one launch (gather + LN in the GEMM prologue); otherwise gather+LN, then Linear."""
    B, H, W, C = x.shape
    N = w.shape[0]
    if x.dtype == torch.bfloat16 and 4 * C in (128, 256, 512) and N % 128 == 0 and N <= 1024 and x.is_contiguous():
        _dev(x, w)
        M = B * (H // 2) * (W // 2)
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
        if _COUNT_ON:
            _count("linear", flops=2.0 * M * N * 4 * C, bytes=x.element_size() * (x.numel() + w.numel() + M * N))
        _check(load_library().master_patch_merge_linear_fwd(_p(x), _p(gamma), _p(beta), _p(w), _p(out), B, H, W, C, N, eps, _dt(x),
                                                            _stream()), "master_patch_merge_linear_fwd")
        return out
    g = patch_merge_ln(x, gamma, beta, eps)
    return linear(g.view(B * (H // 2) * (W // 2), 4 * C), w)


@_on_device
def image_to_nhwc(img, cpad: int, dtype: torch.dtype) -> torch.Tensor:
    """fp32 NCHW [N,3,H,W] -> dtype NHWC [N,H,W,cpad] with zero channels 3..cpad-1. A list/tuple of image batches of one H x W is
    written into consecutive batch slices of ONE output (no concatenated fp32 copy of the inputs is made)."""
    imgs = list(img) if isinstance(img, (list, tuple)) else [img]
    _dev(*imgs)
    imgs = [i.contiguous().float() for i in imgs]
    _, _, H, W = imgs[0].shape
    if any(tuple(i.shape[1:]) != (3, H, W) for i in imgs):
        raise ValueError(f"image_to_nhwc: image batches must all be [N,3,{H},{W}], got {[tuple(i.shape) for i in imgs]}")
    out = torch.empty((sum(i.shape[0] for i in imgs), H, W, cpad), dtype=dtype, device=imgs[0].device)
    lo = 0
    for i in imgs:
        n = i.shape[0]
        _check(load_library().master_image_to_nhwc_fwd(_p(i), _p(out[lo:lo + n]), n, H, W, cpad, _dt(out), _stream()), "master_image_to_nhwc_fwd")
        lo += n
    return out


@_on_device
def conv3x3_zero(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, relu: bool = True,
                 logical_cin: Optional[int] = None) -> torch.Tensor:
    """nn.Conv2d(Cin, Cout, 3, padding=1) (+ReLU) on NHWC x; w_packed = [Cout,3,3,Cin].  logical_cin: the reference layer's input
    channels when x carries zero-padded ones (VGG conv1_1: 3 of 32) - only the work counters use it."""
    _dev(x, w_packed)
    B, H, W, Cin = x.shape
    Cout = w_packed.shape[0]
    out = torch.empty((B, H, W, Cout), dtype=x.dtype, device=x.device)
    if _COUNT_ON:
        lc = Cin if logical_cin is None else logical_cin
        _count("conv3x3", flops=2.0 * B * H * W * Cout * 9 * lc,
               bytes=x.element_size() * (B * H * W * lc + Cout * 9 * lc + out.numel()))
    dt = x3_code(w_packed) if is_split(w_packed) else _dt(x)
    _check(load_library().master_conv3x3_zero_fwd(_p(x), _p(w_packed), _p(bias), _p(out), B, H, W, Cin, Cout, int(relu), dt, _stream()),
           "master_conv3x3_zero_fwd")
    return out


@_on_device
def vgg_conv1(img, weight: torch.Tensor, bias: torch.Tensor, dtype: torch.dtype, relu: bool = True) -> torch.Tensor:
    """Conv2d(3, 64, 3, padding=1) (+ ReLU) straight from fp32 NCHW images -> NHWC [N,H,W,64] in `dtype` (master_vgg_conv1_fwd).  A list of
    image batches of one H x W is written into consecutive batch slices of one output.  weight: nn.Conv2d's [64,3,3,3]."""
    imgs = list(img) if isinstance(img, (list, tuple)) else [img]
    _dev(*imgs)
    imgs = [i.contiguous().float() for i in imgs]
    _, _, H, W = imgs[0].shape
    if any(tuple(i.shape[1:]) != (3, H, W) for i in imgs) or tuple(weight.shape) != (64, 3, 3, 3):
        raise ValueError("vgg_conv1: images [N,3,H,W] of one size and a [64,3,3,3] weight")
    w = weight.detach().float().contiguous()
    b = bias.detach().float().contiguous()
    out = torch.empty((sum(i.shape[0] for i in imgs), H, W, 64), dtype=dtype, device=imgs[0].device)
    lo = 0
    for i in imgs:
        n = i.shape[0]
        if _COUNT_ON:
            _count("conv3x3", flops=2.0 * n * H * W * 27 * 64, bytes=4.0 * n * 3 * H * W + out.element_size() * n * H * W * 64)
        _check(load_library().master_vgg_conv1_fwd(_p(i), _p(w), _p(b), _p(out[lo:lo + n]), n, H, W, int(relu), _dt(out), _stream()), "master_vgg_conv1_fwd")
        lo += n
    return out


@_on_device
def maxpool2x2(x: torch.Tensor) -> torch.Tensor:
    _dev(x)
    N, H, W, C = x.shape
    out = torch.empty((N, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
    _check(load_library().master_maxpool2x2_fwd(_p(x), _p(out), N, H, W, C, _dt(x), _stream()), "master_maxpool2x2_fwd")
    return out


@_on_device
def channel_stats(x: torch.Tensor):
    """NHWC x -> (mean [N,C], biased variance [N,C]) over the positions, fp32"""
    _dev(x)
    N, H, W, C = x.shape
    mean = torch.empty((N, C), dtype=torch.float32, device=x.device)
    var = torch.empty_like(mean)
    lib = load_library()
    work = torch.empty((2 * N * lib.master_stats_slices(H * W) * C,), dtype=torch.float32, device=x.device)
    _check(lib.master_channel_stats_fwd(_p(x), _p(mean), _p(var), _p(work), N, H * W, C, _dt(x), _stream()), "master_channel_stats_fwd")
    return mean, var


@_on_device
def instnorm_distance(x: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, pairs: int, pair_offset: int, squared: bool,
                      eps: float = 1e-5, partials_only: bool = False) -> torch.Tensor:
    """mean over pairs, channels and positions of |IN(x[i]) - IN(x[i + pair_offset])| (or its square): a 0-d fp32 tensor
    (partials_only: the per-workgroup partial sums, for master_loss_terms_fwd to fold)"""
    _dev(x)
    N, H, W, C = x.shape
    partial = torch.empty((pairs * load_library().master_stats_slices(H * W),), dtype=torch.float32, device=x.device)
    _check(load_library().master_instnorm_distance_fwd(_p(x), _p(mean), _p(var), _p(partial), pairs, pair_offset, H * W, C, eps, int(squared),
                                                       _dt(x), _stream()), "master_instnorm_distance_fwd")
    return partial if partials_only else partial.sum() / float(pairs * C * H * W)


_RESIZE_PLANS = {}


def resize_plan(in_size: int, out_size: int, device) -> tuple:
    """Pillow bilinear resample tables of one axis as device tensors (bounds int32 [out,2], coeffs int32 [out,ksize]); cached."""
    key = (in_size, out_size, str(device))
    hit = _RESIZE_PLANS.get(key)
    if hit is None:
        import ctypes
        lib = load_library()
        ks = ctypes.c_int(0)
        _check(lib.master_resize_plan(in_size, out_size, None, None, ctypes.byref(ks)), "master_resize_plan")
        b = torch.empty((out_size, 2), dtype=torch.int32)
        k = torch.empty((out_size, ks.value), dtype=torch.int32)
        _check(lib.master_resize_plan(in_size, out_size, ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(k.data_ptr()), ctypes.byref(ks)),
               "master_resize_plan")
        hit = _RESIZE_PLANS[key] = (b.to(device), k.to(device))
    return hit


@_on_device
def preprocess(img: torch.Tensor, out_h: int, out_w: int, mean=None, std=None) -> torch.Tensor:
    """uint8 [B,H,W,3] RGB -> fp32 [B,3,out_h,out_w]: Pillow-exact bilinear resize, /255, optional (x - mean) / std"""
    import ctypes
    _dev(img)
    assert img.dtype == torch.uint8 and img.dim() == 4 and img.shape[3] == 3 and img.is_contiguous()
    B, H, W, _ = img.shape
    hb, hk = resize_plan(W, out_w, img.device)
    vb, vk = resize_plan(H, out_h, img.device)
    out = torch.empty((B, 3, out_h, out_w), dtype=torch.float32, device=img.device)
    tmp = torch.empty((B, H, out_w, 3), dtype=torch.uint8, device=img.device) if out_w != W else None
    m = (ctypes.c_float * 3)(*mean) if mean is not None else None
    d = (ctypes.c_float * 3)(*std) if std is not None else None
    _check(load_library().master_preprocess_fwd(_p(img), B, H, W, _p(out), out_h, out_w, _p(hb), _p(hk), hk.shape[1], _p(vb), _p(vk),
                                                vk.shape[1], _p(tmp), m, d, _stream()), "master_preprocess_fwd")
    return out


@_on_device
def postprocess(img: torch.Tensor) -> torch.Tensor:
    """fp32 [B,3,H,W] -> uint8 [B,H,W,3] = clip(x * 255, 0, 255) truncated"""
    _dev(img)
    assert img.dtype == torch.float32 and img.dim() == 4 and img.shape[1] == 3
    img = img.contiguous()
    B, _, H, W = img.shape
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=img.device)
    _check(load_library().master_postprocess_fwd(_p(img), _p(out), B, H, W, _stream()), "master_postprocess_fwd")
    return out


@_on_device
def convert(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    _dev(x)
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _check(load_library().master_convert(_p(x), _p(out), x.numel(), _dt(x), F32 if dtype == torch.float32 else BF16,
                                          _stream()), "master_convert")
    return out


# ---- profiling (bench.py roofline leg) ------------------------------------------------------------
def prof_enable(on: bool):
    _check(load_library().master_prof_enable(int(on)), "master_prof_enable")


def prof_reset():
    _check(load_library().master_prof_reset(), "master_prof_reset")


def prof_read():
    lib = load_library()
    res = {}
    for i, name in enumerate(K_FAMILIES):
        n, ms = ctypes.c_int64(0), ctypes.c_double(0.0)
        _check(lib.master_prof_read(i, ctypes.byref(n), ctypes.byref(ms)), "master_prof_read")
        res[name] = (n.value, ms.value)
    return res
