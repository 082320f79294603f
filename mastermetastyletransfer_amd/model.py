"""Drop-in for the reference's ``codes/full_model.py::MasterStyleTransferModel`` on MI355X.

Same constructor keywords (all 37 of full_model.py:22-60), same attributes (``swin_encoder``,
``style_transformer``, ``decoder``), same ``forward(content_image, style_image,
transformer_layer_count=1)`` contract (fp32 NCHW in, fp32 NCHW out, full_model.py:214-226), same
state-dict keys.  The forward enqueues only the HIP kernels of libmaster_hip.so.
"""
from __future__ import annotations

import os
import warnings
from typing import Callable, List, Optional

import torch
from torch import nn, Tensor

from . import ops
from .modules import Decoder, StyleTransformer, SwinEncoder, _HipModule, _train_mode

project_absolute_path = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

_SWIN_VARIANTS = {  # embed_dim, heads and depths of the first two stages, (stochastic_depth_prob, total blocks) of the full torchvision backbone
    "swin_T": (96, (3, 6), (2, 2), (0.2, 12)), "swin_S": (96, (3, 6), (2, 2), (0.3, 24)), "swin_B": (128, (4, 8), (2, 2), (0.5, 24)),
}


def _load_swin_encoder(path: Optional[str], variant: str, allow_random: bool = False) -> SwinEncoder:
    if variant not in _SWIN_VARIANTS:        # same message as codes/utils.py:71-72
        raise ValueError("Invalid Swin Transformer variant. Please choose one of the following: swin_T, swin_S, swin_B")
    embed, heads, depths, (sd_prob, total) = _SWIN_VARIANTS[variant]
    enc = SwinEncoder(embed_dim=embed, depths=depths, num_heads=heads, stochastic_depth_prob=sd_prob, total_blocks=total)
    if path and os.path.exists(path):
        # the reference pickles the whole torchvision nn.Sequential (utils.py:93-102); a plain state dict works too
        obj = torch.load(path, map_location="cpu", weights_only=False)
        sd = obj.state_dict() if isinstance(obj, nn.Module) else obj
        enc.load_state_dict(sd)
    elif allow_random:
        warnings.warn("no Swin encoder weights found at %r: the encoder keeps its random initialisation" % (path,))
    else:
        raise FileNotFoundError(
            "no Swin encoder weights at %r.  The reference downloads ImageNet weights through torchvision here (codes/utils.py:82-102); "
            "this machine has no network.  Save swin_b().features[:4] (module or state dict) there, or pass "
            "allow_random_swin_encoder=True to run with a randomly initialised encoder (benchmarks, tests)." % (path,))
    return enc


class MasterStyleTransferModel(nn.Module):
    def __init__(self,
                 project_absolute_path: str = project_absolute_path,
                 swin_model_relative_path: str = None,
                 swin_variant: str = "swin_B",
                 style_encoder_dim: int = 256,
                 style_decoder_dim: int = 256,
                 style_encoder_num_heads: int = 8,
                 style_decoder_num_heads: int = 8,
                 style_encoder_window_size: List[int] = [8, 8],
                 style_decoder_window_size: List[int] = [8, 8],
                 style_encoder_shift_size: List[int] = [4, 4],
                 style_decoder_shift_size: List[int] = [4, 4],
                 style_encoder_mlp_ratio: float = 4.0,
                 style_decoder_mlp_ratio: float = 4.0,
                 style_encoder_dropout: float = 0.0,
                 style_decoder_dropout: float = 0.0,
                 style_encoder_attention_dropout: float = 0.0,
                 style_decoder_attention_dropout: float = 0.0,
                 style_encoder_qkv_bias: bool = True,
                 style_decoder_qkv_bias: bool = True,
                 style_encoder_proj_bias: bool = True,
                 style_decoder_proj_bias: bool = True,
                 style_encoder_stochastic_depth_prob: float = 0.1,
                 style_decoder_stochastic_depth_prob: float = 0.1,
                 style_encoder_norm_layer: Callable[..., nn.Module] = None,
                 style_decoder_norm_layer: Callable[..., nn.Module] = nn.LayerNorm,
                 style_encoder_MLP_activation_layer: Optional[Callable[..., nn.Module]] = nn.GELU,
                 style_decoder_MLP_activation_layer: Optional[Callable[..., nn.Module]] = nn.GELU,
                 style_encoder_if_use_processed_Key_in_Scale_and_Shift_calculation: bool = True,
                 style_decoder_use_instance_norm_with_affine: bool = False,
                 style_decoder_use_regular_MHA_instead_of_Swin_at_the_end: bool = False,
                 style_decoder_use_Key_instance_norm_after_linear_transformation: bool = True,
                 style_decoder_exclude_MLP_after_Fcs_self_MHA: bool = False,
                 style_transformer_load_pretrained_weights: bool = False,
                 style_transformer_pretrained_weights_path: str = None,
                 decoder_initializer: str = "kaiming_normal_",
                 direct_pretrained_style_transformer_path: str = '',
                 direct_pretrained_decoder_path: str = '',
                 compute_dtype: torch.dtype = torch.float32,
                 allow_random_swin_encoder: bool = False):
        super().__init__()
        for k, v in list(locals().items()):          # the reference mirrors every ctor argument as an attribute
            if k.startswith(("style_", "decoder_initializer", "direct_pretrained")):
                setattr(self, k, v)

        swin_path = None
        if swin_model_relative_path is not None:
            swin_path = os.path.join(project_absolute_path, swin_model_relative_path)
        else:
            swin_path = os.path.join(project_absolute_path, "models", f"{swin_variant}_first_2_stages.pt")
        self.swin_encoder = _load_swin_encoder(swin_path, swin_variant, allow_random_swin_encoder)
        self.swin_variant = swin_variant

        self.style_transformer = StyleTransformer(
            encoder_dim=style_encoder_dim, decoder_dim=style_decoder_dim,
            encoder_num_heads=style_encoder_num_heads, decoder_num_heads=style_decoder_num_heads,
            encoder_window_size=style_encoder_window_size, decoder_window_size=style_decoder_window_size,
            encoder_shift_size=style_encoder_shift_size, decoder_shift_size=style_decoder_shift_size,
            encoder_mlp_ratio=style_encoder_mlp_ratio, decoder_mlp_ratio=style_decoder_mlp_ratio,
            encoder_dropout=style_encoder_dropout, decoder_dropout=style_decoder_dropout,
            encoder_attention_dropout=style_encoder_attention_dropout, decoder_attention_dropout=style_decoder_attention_dropout,
            encoder_qkv_bias=style_encoder_qkv_bias, decoder_qkv_bias=style_decoder_qkv_bias,
            encoder_proj_bias=style_encoder_proj_bias, decoder_proj_bias=style_decoder_proj_bias,
            encoder_stochastic_depth_prob=style_encoder_stochastic_depth_prob,
            decoder_stochastic_depth_prob=style_decoder_stochastic_depth_prob,
            encoder_norm_layer=style_encoder_norm_layer, decoder_norm_layer=style_decoder_norm_layer,
            encoder_MLP_activation_layer=style_encoder_MLP_activation_layer,
            decoder_MLP_activation_layer=style_decoder_MLP_activation_layer,
            encoder_if_use_processed_Key_in_Scale_and_Shift_calculation=style_encoder_if_use_processed_Key_in_Scale_and_Shift_calculation,
            decoder_use_instance_norm_with_affine=style_decoder_use_instance_norm_with_affine,
            decoder_use_regular_MHA_instead_of_Swin_at_the_end=style_decoder_use_regular_MHA_instead_of_Swin_at_the_end,
            decoder_use_Key_instance_norm_after_linear_transformation=style_decoder_use_Key_instance_norm_after_linear_transformation,
            decoder_exclude_MLP_after_Fcs_self_MHA=style_decoder_exclude_MLP_after_Fcs_self_MHA)
        self.decoder = Decoder(channel_dim=style_decoder_dim, initializer=decoder_initializer)

        if style_transformer_load_pretrained_weights and not direct_pretrained_style_transformer_path:
            self.load_pretained_weights_to_style_transformer(style_transformer_pretrained_weights_path)
        if direct_pretrained_style_transformer_path != '':           # full_model.py:147-150
            self.style_transformer.load_state_dict(torch.load(direct_pretrained_style_transformer_path, map_location="cpu"))
        if direct_pretrained_decoder_path != '':                     # full_model.py:152-155
            self.decoder.load_state_dict(torch.load(direct_pretrained_decoder_path, map_location="cpu"))
        self.set_compute_dtype(compute_dtype)

    def load_pretained_weights_to_style_transformer(self, pretrained_weights_path: str):
        if pretrained_weights_path is None:                          # full_model.py:160-161
            raise ValueError("Please provide the path of the pretrained weights")
        # full_model.py:163-193 -> load_pretrained_weights_to_style_transformer.py:688-733: one MS-Swin block copied into every
        # attention / MLP of the StyleTransformer (host-side state-dict slicing, weights.py)
        from .weights import remap_ms_swin_block
        path = pretrained_weights_path
        if not os.path.isabs(path) and not os.path.exists(path):
            path = os.path.join(project_absolute_path, path)
        if not os.path.exists(path):
            raise ValueError(f"The path of the pretrained weights does not exist: {path}")
        block_sd = torch.load(path, map_location="cpu")
        sd = remap_ms_swin_block(
            block_sd, self.style_transformer.state_dict(), encoder_dim=self.style_encoder_dim, decoder_dim=self.style_decoder_dim,
            encoder_mlp_ratio=self.style_encoder_mlp_ratio, decoder_mlp_ratio=self.style_decoder_mlp_ratio,
            encoder_window_size=list(self.style_encoder_window_size), decoder_window_size=list(self.style_decoder_window_size),
            encoder_qkv_bias=self.style_encoder_qkv_bias, decoder_qkv_bias=self.style_decoder_qkv_bias,
            encoder_proj_bias=self.style_encoder_proj_bias, decoder_proj_bias=self.style_decoder_proj_bias,
            encoder_norm_layer=self.style_encoder_norm_layer, decoder_norm_layer=self.style_decoder_norm_layer,
            decoder_exclude_MLP_after_Fcs_self_MHA=self.style_decoder_exclude_MLP_after_Fcs_self_MHA)
        self.style_transformer.load_state_dict(sd)

    def set_compute_dtype(self, dtype, split_bf16: bool = False, split: Optional[str] = None) -> "MasterStyleTransferModel":
        """torch.float32: exact-fp32 matrix cores, parity mode.  torch.bfloat16: bf16 MFMA, throughput mode.
        "f16x3" / "bf16x3" (= torch.float32 with split="f16" / "bf16"; split_bf16=True is "bf16"): the FAST parity modes - fp32 storage,
        norms, softmax and epilogues as in the fp32 mode, every Linear / convolution / attention product as three 16-bit MFMAs on operands
        split into hi + lo halves (csrc/gemm_x3.hip).  "f16x3" (r05) uses IEEE-half halves: hi + lo carries 22 significant bits, the stylized
        image sits at the exact mode's distance from the CPU path (~1e-4 max-abs on outputs of max-abs 4 - 17; bar 1e-3) as long as every
        activation stays inside half's range (|x| <= 65 504, as under fp16 autocast; beyond it the output is inf / NaN, never silently
        wrong).  "bf16x3" (r04) uses bf16 halves: fp32's exponent range, 17 significant bits, 5e-4 ... 1e-3 on the same pairs - inside the
        bar on the benchmark pair (1.1e-4) but without margin on others (profiles/r05_x3_split_sim.txt).  Same throughput.  Inference only."""
        if dtype in ("bf16x3", "f16x3"):
            dtype, split = torch.float32, dtype[:-2]
        if split is None and split_bf16:
            split = "bf16"
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute_dtype must be torch.float32, torch.bfloat16, \"f16x3\" or \"bf16x3\"")
        if split not in (None, "bf16", "f16"):
            raise ValueError("split must be None, \"bf16\" or \"f16\"")
        if split and dtype != torch.float32:
            raise ValueError("the split modes are forms of the fp32 mode: set_compute_dtype(torch.float32, split=\"f16\")")
        self.compute_dtype = dtype
        self.split_weights = split or False
        for m in self.modules():
            if isinstance(m, (_HipModule, SwinEncoder)):
                m.compute_dtype = dtype
                m.split_weights = split or False
        # swin_T / swin_S: stage 1 is 96 channels wide and the bf16 GEMMs step K in 64s, so the patch embedding, the two stage-1 blocks
        # and the patch merging stay in fp32 (SwinEncoder.run converts once, after the merge); stage 2, the StyleTransformer and the
        # decoder (192 / 768 / 384 wide) run on the generic bf16 kernels - the fused ones are specialised for swin_B's 128 / 256.
        self.swin_encoder.stage1_dtype = dtype
        if dtype == torch.bfloat16 and self.swin_encoder[0][0].out_channels % 64 != 0:
            self.swin_encoder.stage1_dtype = torch.float32
            for part in (self.swin_encoder[1], self.swin_encoder[2]):
                for m in part.modules():
                    if isinstance(m, _HipModule):
                        m.compute_dtype = torch.float32
        return self

    def set_encoder_compute_dtype(self, dtype) -> "MasterStyleTransferModel":
        """The FROZEN Swin encoder alone in torch.float32 (exact kernels) or in a split mode ("f16x3" / "bf16x3"), whatever the rest of the
        model runs in - meant for the train step (r05): the encoder is frozen there by default (train_only_inner_loop.py:254-257) and runs
        without a graph, so its 15 GFLOP per pair need no backward kernels; with IEEE-half halves its features sit at the exact kernels'
        distance from the CPU path and every parameter gradient of the step stays inside the bound the all-fp32 step is held to
        (tests/test_gpu_f16x3.py::test_train_step_with_f16x3_loss_network_and_encoder).  (With bf16 halves they do not: 7.8e-3 of a tensor's
        max against the 5e-3 bound, r04.)  A later set_compute_dtype() overrides it."""
        split = None
        if dtype in ("f16x3", "bf16x3"):
            dtype, split = torch.float32, dtype[:-2]
        if dtype != torch.float32:
            raise ValueError('the encoder-only switch takes torch.float32, "f16x3" or "bf16x3"')
        for m in self.swin_encoder.modules():
            if isinstance(m, (_HipModule, SwinEncoder)):
                m.compute_dtype = torch.float32
                m.split_weights = split or False
        self.swin_encoder.stage1_dtype = torch.float32
        return self

    def forward(self, content_image: Tensor, style_image: Tensor, transformer_layer_count: int = 1) -> Tensor:
        if content_image.shape != style_image.shape:
            raise ValueError("content and style batches must have the same shape")
        B = content_image.shape[0]
        if _train_mode(self):
            return self._forward_train(content_image, style_image, transformer_layer_count)
        # the two encoder passes of full_model.py:219-220 are independent -> one batch of 2B images
        x, _, h, w = self.swin_encoder.run(content_image, style_image)
        self._check_widths(x.shape[1])
        if x.dtype != self.compute_dtype:            # set_encoder_compute_dtype() left the encoder in fp32 storage next to a bf16 model
            x = ops.convert(x, self.compute_dtype)
        fc, fs = x[: B * h * w], x[B * h * w:]
        fcs = self.style_transformer.run(fc, fs, transformer_layer_count, B, h, w)
        # the reference permutes NHWC->NCHW here (:222); the decoder kernels read NHWC directly and the
        # last convolution writes the NCHW image
        return self.decoder.run_nhwc(fcs.view(B, h, w, -1))

    def _check_widths(self, c: int) -> None:
        """the encoder's output width (256 for swin_B, 192 for swin_T / swin_S) feeds both halves of the StyleTransformer and the
        decoder; the reference fails in its first nn.Linear with the same kind of error"""
        if c != self.style_encoder_dim or c != self.style_decoder_dim:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied: the {self.swin_variant} encoder produces {c}-channel tokens, "
                               f"style_encoder_dim = {self.style_encoder_dim}, style_decoder_dim = {self.style_decoder_dim}")

    def _forward_train(self, content_image: Tensor, style_image: Tensor, k: int) -> Tensor:
        """train-mode forward (train_only_inner_loop.py:523): autograd graph over the HIP forward/backward kernels.  The Swin encoder
        is frozen like the reference's default (:254-257): it runs without a graph (its StochasticDepth is still active in train mode);
        gradients flow into the StyleTransformer and the image decoder."""
        from . import train_forward as TF
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.swin_encoder.parameters()):
            raise NotImplementedError("fine-tuning the Swin encoder is not on the HIP path: freeze it "
                                      "(for p in model.swin_encoder.parameters(): p.requires_grad = False - the reference's default)")
        B = content_image.shape[0]
        with torch.no_grad():
            x, _, h, w = self.swin_encoder.run(content_image, style_image)
        self._check_widths(x.shape[1])
        fc, fs = x[: B * h * w], x[B * h * w:]
        fcs = TF.style_transformer(self.style_transformer, fc, fs, k, B, h, w, TF.sd_masks)
        return TF.image_decoder(self.decoder, fcs.view(B, h, w, -1))
