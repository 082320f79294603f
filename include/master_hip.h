/*
 * master_hip.h - C ABI of libmaster_hip.so: the MI355X (gfx950) kernels behind the Master
 * stylization hot path  MasterStyleTransferModel.forward  (reference codes/full_model.py:214-226).
 *
 * The reference has no FFI of its own (it is pure PyTorch); its boundary is the nn.Module surface
 * of codes/full_model.py.  These entry points are what a maintainer binds (ctypes stub shown in
 * INTEGRATION.md) to replace the torch op sequences cited on each function.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed for the duration of the call; nothing is retained,
 *     freed or allocated; work is enqueued on `stream` (a hipStream_t) and the call returns at once.
 *   - activations are token-major ("NHWC"): a feature map [B,H,W,C] is a row-major matrix
 *     [B*H*W, C]; images at the two ends are NCHW like the reference's.
 *   - `dtype`: MASTER_F32 = fp32 storage + exact-fp32 MFMA (parity mode, <=1e-3 vs the CPU path);
 *              MASTER_BF16 = bf16 storage + bf16 MFMA with fp32 accumulation (throughput mode).
 *     Norm statistics, softmax and all epilogues are fp32 in both modes.
 *   - return value: MASTER_OK or a negative master_status; master_last_error_string() describes
 *     the last failure on the calling thread.  Nothing throws, nothing synchronises the device.
 *   - threads / devices: entry points may be called from several host threads and for several GPUs of one process; a call
 *     works on the CURRENT HIP device (callers make the operands' device current).  The only state kept between calls is a
 *     per-(kernel, device) atomic remembering the dynamic-LDS limit already raised there, the mutex-protected timing records of
 *     master_prof_*, and the thread-local error string.
 */
#ifndef MASTER_HIP_H
#define MASTER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* master_stream_t; /* hipStream_t */

enum master_status {
    MASTER_OK = 0,
    MASTER_ERR_INVALID = -1,     /* bad argument / shape the kernels cannot take */
    MASTER_ERR_UNSUPPORTED = -2, /* valid in the reference, not implemented here */
    MASTER_ERR_LAUNCH = -3       /* hipLaunch reported an error */
};

enum master_dtype {
    MASTER_F32 = 0, MASTER_BF16 = 1,
    /* fp32 storage, contractions as three bf16 MFMAs per product on split operands (x = hi + lo, a*w = a_lo*w_hi + a_hi*w_lo +
     * a_hi*w_hi, fp32 accumulation): the r04 fast parity mode - 1.1e-4 max-abs on the stylized image against the fp32 CPU path on
     * the benchmark pair (outputs of max-abs 3.5; bar 1e-3), 6e-4 ... 1.1e-3 on pairs whose outputs reach 12 - 24: at the bar, without
     * margin (MASTER_F16X3 below holds it with margin) - at 5.3x the matrix-core rate of the exact-fp32 form; fp32's exponent range.  Accepted by master_linear_fwd, master_conv3x3_reflect_fwd and
     * master_mlp_fwd (activations, bias, residuals and outputs fp32 exactly as with MASTER_F32, the WEIGHT operand must have gone through
     * master_split_weights) and by master_window_attn_fwd (all operands fp32: Q, K, V and the probabilities are split in registers).
     * Every other kernel of this mode runs with MASTER_F32. */
    MASTER_F32X3 = 2,
    /* The same three-product form with IEEE-HALF halves (hi = half(x), lo = half(x - hi); v_mfma_f32_32x32x16_f16): 11 significant bits per
     * half instead of 8, so hi + lo carries x to 2^-22 instead of 2^-17 and the stylized image sits at the exact-fp32 mode's distance from
     * the CPU path (r05, 36 pairs on MI355X: worst 1.3e-4 max-abs; exact-fp32 kernels 1.2e-4; bf16 halves 1.1e-3), at the same matrix-core rate.  What it
     * gives up is fp32's exponent range: an operand beyond +-65 504 becomes inf (as under fp16 autocast), values below 2^-14 lose their
     * low half gradually (the matrix core takes subnormal halves).  Accepted wherever MASTER_F32X3 is, except master_conv3x3_dgrad (gradients
     * need fp32's range); weights = master_split_weights(..., layout | MASTER_SPLIT_F16). */
    MASTER_F16X3 = 3
};
/* OR-ed into the `dtype` argument of master_linear_wgrad / master_colsum / master_layernorm_bwd: the parameter gradient is ADDED to
 * the output buffer (dw and dbias / out / dgb) instead of overwriting it - the gradient of a tensor used several times in one graph (the k
 * StyleTransformer layers share their weights, style_transformer.py:1239) is summed by the kernel's own slab reduce instead of by one
 * autograd add launch per use.  Needs the workspace (at least one output-sized slab). */
enum master_flags {
    MASTER_ACCUMULATE = 0x100,
    /* OR-ed into MASTER_F32X3 for master_linear_fwd / master_conv3x3_reflect_fwd: the activation operand (a / x) is already in the SPLIT
     * layout - per row (token / pixel) and 32-wide channel chunk 32 bf16 hi | 32 bf16 lo, the layout master_split_weights produces, the
     * same bytes and row stride as the fp32 matrix - and / or the OUTPUT is written in it (N, ldo % 32 == 0; NHWC).  A chain of
     * convolutions (decoder.py:24-54) then hands its maps on in the form the next layer's LDS-DMA pieces copy: no split arithmetic and
     * no LDS stores in the consumer's loop.  hi + lo reproduces the fp32 value to 2^-17 relative. */
    MASTER_A_SPLIT = 0x200, MASTER_OUT_SPLIT = 0x400,
    /* OR-ed into the `layout` argument of master_split_weights: IEEE-half halves (for MASTER_F16X3) instead of bf16 halves */
    MASTER_SPLIT_F16 = 0x10
};
enum master_act { MASTER_ACT_NONE = 0, MASTER_ACT_GELU = 1 /* exact erf */, MASTER_ACT_RELU = 2 };

int master_version(void);
const char* master_last_error_string(void);

/* out[M,N] = r1 (.) act(a[M,K] . w[N,K]^T + bias[N]) + r2       (r1, r2, bias optional = NULL)
 * Replaces F.linear / nn.Linear + GELU + residual adds:
 *   style_transformer.py:114-116,156 (Wq/Wk/Wv/proj), :366,:839-841,:991 (torchvision MLP),
 *   :390-397,:865-882,:1125 (residual adds), :1123 (Query*sigma+mu via r1=Query, r2=mu).
 * Because a Linear acts per token, the projections run on the un-padded, un-rolled token matrix;
 * padding/roll/window partition are index math inside master_window_attn_fwd.
 * lda/ldo/ldr are row strides in elements (r1 and r2 share ldr).  K % 32 == 0.
 * dtype = MASTER_F32X3: a, r1, r2, out fp32, w = master_split_weights(fp32 weight); lda % 4 == 0. */
int master_linear_fwd(const void* a, const void* w, const void* bias, const void* r1, const void* r2,
                      void* out, int M, int N, int K, int lda, int ldo, int ldr, int act, int dtype,
                      master_stream_t stream);

/* Weight preparation of the MASTER_F32X3 mode: w fp32 [rows, K] (K % 32 == 0; a conv weight as [Cout, 9*Cin]) -> out, the same number
 * of bytes: per row and 32-wide k chunk 32 bf16 hi = bf16(w) followed by 32 bf16 lo = bf16(w - hi) (round to nearest even) - the
 * 128-byte row segments the GEMM's LDS-DMA pieces copy.  Done once per weight (the host mirror caches it per parameter version).
 * layout 0: k in natural order (master_linear_fwd, master_conv3x3_reflect_fwd, w1 of master_mlp_fwd).  layout 1: inside every group of
 * 16 k the order 0-3, 8-11, 4-7, 12-15 - the order in which a packed 32x32 MFMA accumulator tile presents its rows as an operand; w2 of
 * master_mlp_fwd (MASTER_F32X3) must be in this layout, so that the GELU'd hidden tile feeds the second product without a lane exchange.
 * layout | MASTER_SPLIT_F16: the halves are IEEE half instead of bf16 (the weight operand of MASTER_F16X3). */
int master_split_weights(const float* w, void* out, size_t rows, size_t K, int layout, master_stream_t stream);

/* master_linear_fwd with the block's LayerNorm folded in:  out = r1 (.) (LayerNorm(x) . w^T + bias) + r2, one launch
 * (norm1 + the fused qkv projection of a Swin / StyleSwinTransformerBlock: style_transformer.py:390, :114-116).
 * ln_gamma/ln_beta fp32 [K].  MASTER_BF16 only, K in {128,256,512}, N % 128 == 0, residuals only for N <= 256
 * (MASTER_ERR_UNSUPPORTED otherwise: callers then run master_layernorm_fwd + master_linear_fwd). */
int master_linear_ln_fwd(const void* x, const float* ln_gamma, const float* ln_beta, const void* w, const void* bias,
                         const void* r1, const void* r2, void* out, int M, int N, int K, int ldx, int ldo, int ldr,
                         float eps, int dtype, master_stream_t stream);

/* Fused MLP block:  out[M,C] = res + W2 . gelu(W1 . LN?(x) + b1) + b2   in ONE launch; the hidden
 * activations stay in registers.  x, res, out: [M,C] token matrices (row strides ldx/ldr/ldo); w1 [HID,C],
 * w2 [C,HID] as nn.Linear stores them; b1, b2, ln_gamma, ln_beta fp32; ln_gamma == NULL skips the LayerNorm;
 * res == NULL skips the residual.  Replaces torchvision MLP + residual (+ the preceding norm2):
 * style_transformer.py:366,:392,:839-841,:865-882,:991,:1125.  MASTER_BF16: C in {128,256}, HID % 128 == 0.  MASTER_F32X3: x, res, out
 * fp32, w1 = master_split_weights(W1, layout 0), w2 = master_split_weights(W2, layout 1), exact-erf GELU, C in {128,256}, HID % 64 == 0.
 * (MASTER_ERR_UNSUPPORTED otherwise, and for MASTER_F32: callers then use two master_linear_fwd calls). */
int master_mlp_fwd(const void* x, const float* ln_gamma, const float* ln_beta, const void* w1, const float* b1,
                   const void* w2, const float* b2, const void* res, void* out, int M, int C, int HID, int ldx,
                   int ldr, int ldo, float eps, int dtype, master_stream_t stream);

/* 3x3 convolution, reflect padding 1, optional nearest x2 upsample of the input fused into the
 * gather, optional ReLU.  x is NHWC [B,Hin,Win,Cin]; w is repacked [Cout][3][3][Cin] (= weight
 * .permute(0,2,3,1)); out is NHWC [B,Hout,Wout,Cout] or, with out_nchw, NCHW [B,Cout,Hout,Wout]
 * (always fp32 when out_nchw: it is the model output), Hout = Hin << upsample.
 * Replaces nn.Upsample + nn.Conv2d(padding_mode='reflect') + nn.ReLU, decoder.py:23-55.
 * Cin % 32 == 0.  dtype = MASTER_F32X3: x and out fp32, w = master_split_weights of the [Cout, 9*Cin] matrix. */
int master_conv3x3_reflect_fwd(const void* x, const void* w, const void* bias, void* out, int B, int Hin,
                               int Win, int Cin, int Cout, int upsample, int relu, int out_nchw, int dtype,
                               master_stream_t stream);

/* Decoder tail in one launch: nearest x2 upsample -> Conv3x3(32->32, reflect) + ReLU -> Conv3x3(32->3, reflect).
 * x: NHWC [B,Hs,Ws,32]; w8 [32][3][3][32], w9 [3][3][3][32] repacked like master_conv3x3_reflect_fwd; b8, b9 fp32;
 * out: fp32 NCHW [B,3,2Hs,2Ws] (the model output).  The 32-channel full-resolution intermediate never leaves LDS.
 * Replaces the last four layers of decoder.py:47-54.  MASTER_BF16: 2Hs % 16 == 0 and 2Ws % 32 == 0.  MASTER_F32X3: x fp32 NHWC or, with
 * MASTER_A_SPLIT OR-ed in, in the split layout; w8, w9 = master_split_weights of the [32, 288] / [3, 288] matrices; 2Hs % 8 == 0 and
 * 2Ws % 32 == 0.  (MASTER_ERR_UNSUPPORTED otherwise, and for MASTER_F32: callers then use two master_conv3x3_reflect_fwd calls). */
int master_decoder_tail_fwd(const void* x, const void* w8, const float* b8, const void* w9, const float* b9, float* out,
                            int B, int Hs, int Ws, int dtype, master_stream_t stream);

/* Windowed multi-head attention core for one or two value streams sharing one softmax.
 * For batch b, window w, head h:   S = scale * Q K^T + bias[h] + mask(w);  P = softmax(S);
 * out_s = P V_s  (s = 0, and 1 when v1 != NULL), written for real (non-padded) query tokens only.
 *   q,k,v0,v1 : token matrices [B*T, *] with row strides ldq/ldk/ldv (elements); head h uses
 *               columns [h*32, h*32+32) from the given base pointer (so a fused [M,3C] QKV buffer is
 *               passed as three offset pointers with ld = 3C).  head_dim is 32.
 *   src       : int32 [nW, 64]; src[w][t] = token index (h*W+w) feeding slot t of window w after
 *               pad -> roll -> partition, -1 for a zero-padded slot, slots >= N unused.
 *   k_pad, v0_pad, v1_pad : the key/value row a padded slot holds (= the projection bias, because
 *               the reference pads BEFORE the Linear); [C] or, with *_pad_bstride = C, [B, C].
 *   bias      : fp32 [heads, 64, 64]; bias[h][i][j] = rel-pos bias for query i, key j; entries with
 *               j >= N must be -1e30 (they mask the tile padding).
 *   region    : uint8 [nW, 64] shift-mask region id per slot, or NULL when no shift is active;
 *               mask = -100 where region[i] != region[j].
 * Replaces style_transformer.py:120-155 and :536-575,594 (heads split, scale, QK^T, bias, mask,
 * softmax, AV) plus the gathers of :83-111 and scatters of :160-168. */
int master_window_attn_fwd(const void* q, const void* k, const void* v0, const void* v1, int ldq, int ldk,
                           int ldv, const void* k_pad, const void* v0_pad, const void* v1_pad,
                           int k_pad_bstride, int v_pad_bstride, const float* bias, const int32_t* src,
                           const uint8_t* region, void* out0, void* out1, int ldo, int B, int T, int heads,
                           int nW, int N, float scale, int dtype, master_stream_t stream);

/* master_window_attn_fwd (one value stream) with the output projection and the block's residual folded into the same launch:
 *     out[token, :] = res[token, :] + wproj . concat_heads(P V) + bproj         (res optional = NULL)
 * wproj [C, C] row-major as nn.Linear stores it, bproj fp32 [C]; res / out token matrices [B*T, C] with row strides ldr / ldo.
 * A workgroup is a whole window (wave = head); the attention output never goes to HBM.  Replaces style_transformer.py:120-168
 * (attention core, proj :156, window reverse / un-pad) and the residual add of :391 for the calls with ONE value stream.
 * MASTER_BF16 with heads == 8 (C == 256) and nW <= ~100 windows per image only; MASTER_ERR_UNSUPPORTED otherwise (callers then use
 * master_window_attn_fwd + master_linear_fwd). */
int master_window_attn_proj_fwd(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, const void* k_pad,
                                const void* v_pad, int k_pad_bstride, int v_pad_bstride, const float* bias, const int32_t* src,
                                const uint8_t* region, const void* wproj, const float* bproj, const void* res, int ldr, void* out,
                                int ldo, int B, int T, int heads, int nW, int N, float scale, int dtype, master_stream_t stream);

/* n (1..3) independent Linear layers of one shape class in ONE launch:  out_i[M, N_i] = LN?(x_i) . w_i^T + bias_i
 * (same M and K; N_i % 128 == 0; no residual operands; x, w, bias, out, N, ldx, ldo are HOST arrays of n entries; bias_i may be NULL).
 * Replaces the groups of projections that feed one attention call from different inputs: Wk . Key | Wv_scale . Scale | Wv_shift . Shift
 * of the sigma/mu attention (style_transformer.py:515-517) and Wq|Wk . Key + Wv . Scale of the StyleEncoder's Scale / Shift call
 * (:114-116 with q = k) - each a one-tile-per-CU launch on its own at bs = 32.  MASTER_BF16, K in {128, 256}; MASTER_ERR_UNSUPPORTED
 * otherwise (callers then use master_linear_fwd per problem). */
int master_linear_group_fwd(int n, const void* const* x, const void* const* w, const void* const* bias, void* const* out, const int* N,
                            const int* ldx, const int* ldo, const float* ln_gamma, const float* ln_beta, int M, int K, float eps,
                            int dtype, master_stream_t stream);

/* Whole attention half of a (Style)SwinTransformerBlock with q = k = v in ONE launch:
 *     out = x + proj( window_attention( qkv( LayerNorm?(x) ) ) )
 * x, out: token matrices [B*T, C]; wqkv [3C, C] = rows of Wq | Wk | Wv (torchvision's fused qkv.weight), bqkv [3C] fp32;
 * wproj [C, C], bproj fp32; bias64 / src / region as in master_window_attn_fwd; ln_gamma == NULL skips the LayerNorm.
 * Q, K, V and the probabilities never leave the register file.  Replaces style_transformer.py:390 (norm1), :83-168.
 * MASTER_BF16 with C == 128, heads == 4 (Swin-B stage 1) only; MASTER_ERR_UNSUPPORTED otherwise (callers then use
 * master_linear_ln_fwd + master_window_attn_fwd + master_linear_fwd). */
int master_attn_block_fwd(const void* x, const float* ln_gamma, const float* ln_beta, const void* wqkv, const float* bqkv,
                          const void* wproj, const float* bproj, const float* bias64, const int32_t* src,
                          const uint8_t* region, void* out, int B, int T, int C, int heads, int nW, int N, float eps,
                          float scale, int dtype, master_stream_t stream);

/* out[m,:] = LayerNorm(x[m,:]) * gamma + beta over C channels (nn.LayerNorm, eps inside the sqrt;
 * style_transformer.py:340-343,:390-392).  C % 4 == 0, C <= 1024. */
int master_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* out, int M, int C,
                         float eps, int dtype, master_stream_t stream);

/* InstanceNorm2d(affine=False) on a token-major map: per (b, c) over the T tokens of image b, biased
 * variance.  When n_pad > 0 the statistics also include n_pad copies of pad_row_in[c] (the bias-only
 * rows of the zero-padded map, style_transformer.py:520-530) and pad_row_out[b, c] receives the
 * normalised pad row.  repeat = 2 applies the norm twice in one launch (Query is instance-normalised at :1056 and
 * again at :468; the second pass of a mean-0 map only rescales by rsqrt(v/(v+eps) + eps)); repeat = 1 otherwise.
 * Replaces style_transformer.py:468, :1056-1057, :526. */
int master_instnorm_fwd(const void* x, void* out, const void* pad_row_in, void* pad_row_out, int B, int T,
                        int C, int n_pad, float eps, int repeat, int dtype, master_stream_t stream);

/* Swin patch embedding: Conv2d(3, E, 4, stride 4) on an fp32 NCHW image, NHWC output, LayerNorm(E)
 * fused.  w is the conv weight [E, 3, 4, 4] flattened to [E, 48].  (torchvision swin_b.features[0];
 * reference codes/utils.py:89-93.)  H % 4 == 0, W % 4 == 0; E == 128 (swin_B), or 96 (swin_T / swin_S: MASTER_F32 only). */
int master_patch_embed_fwd(const float* img, const void* w, const void* bias, const void* gamma,
                           const void* beta, void* out, int B, int H, int W, int E, float eps, int dtype,
                           master_stream_t stream);

/* Swin PatchMerging front half: gather x[0::2,0::2] | x[1::2,0::2] | x[0::2,1::2] | x[1::2,1::2]
 * -> LayerNorm(4C).  x NHWC [B,H,W,C] -> out [B,H/2,W/2,4C].  The Linear(4C,2C,no bias) that follows
 * is a master_linear_fwd.  H, W even. */
int master_patch_merge_ln_fwd(const void* x, const void* gamma, const void* beta, void* out, int B, int H,
                              int W, int C, float eps, int dtype, master_stream_t stream);

/* The whole Swin PatchMerging in one launch (bf16): the 2x2 gather and LayerNorm(4C) above happen in the prologue of
 * the token-stationary Linear(4C, N, no bias): out[B*H/2*W/2, N] = LN(gather(x)) . w^T, w = [N, 4C].  Replaces torchvision
 * swin_b.features[2] (built at reference codes/utils.py:89-93).  MASTER_ERR_UNSUPPORTED unless dtype = bf16, 4C in
 * {128,256,512}, N % 128 == 0. */
int master_patch_merge_linear_fwd(const void* x, const float* gamma, const float* beta, const void* w, void* out, int B,
                                  int H, int W, int C, int N, float eps, int dtype, master_stream_t stream);

/* ---- perceptual loss forward (reference codes/loss.py: VGG19_custom :15-37, custom_loss.get_overall_loss :201-336; evaluated under
 *      no_grad by test_model.py:190-203).  All maps NHWC.
 * master_image_to_nhwc_fwd: fp32 NCHW image [N,3,H,W] -> [N,H,W,Cpad], channels >= 3 zero (the first VGG conv runs with Cin = Cpad = 32
 *   and zero weights on the padding).
 * master_conv3x3_zero_fwd: nn.Conv2d(Cin, Cout, 3, padding=1) (+ReLU): the implicit GEMM of master_conv3x3_reflect_fwd with zero
 *   padding; w = [Cout][3][3][Cin], Cin % 32 == 0.  dtype = MASTER_F32X3: fp32 maps, w = master_split_weights of [Cout, 9*Cin].
 * master_maxpool2x2_fwd: nn.MaxPool2d(2, 2).
 * master_channel_stats_fwd: mean and BIASED variance over the HW positions of every (image, channel) (two passes, fp32):
 *   x.mean([2,3]); x.std([2,3]) = sqrt(var * HW / (HW-1)); the statistics of nn.InstanceNorm2d.  work = 2 * N * S * C floats of
 *   workspace, S = master_stats_slices(HW) row slices (per-slice partials are added in a fixed order: deterministic).
 * master_instnorm_distance_fwd: for pairs (image i, image i + pair_offset), i < pairs: per (row slice, pair) partial sums of
 *   |IN(a) - IN(b)| (squared = 0) or (IN(a) - IN(b))^2 into partial[pairs * S] - the caller's fixed-order sum / (pairs*C*HW)
 *   is content_loss_each_term (loss.py:112-118).  C a power-of-two multiple of 8 (bf16) / 4 (fp32) up to 2048 / 1024. */
int master_image_to_nhwc_fwd(const float* img, void* out, int N, int H, int W, int Cpad, int dtype, master_stream_t stream);
/* VGG-19 conv1_1 (+ ReLU) straight from the fp32 NCHW image: Conv2d(3, 64, 3, padding = 1) (loss.py:23-37) with w in nn.Conv2d's own
 * [64][3][3][3] layout -> NHWC [N,H,W,64] in `dtype`.  Replaces master_image_to_nhwc_fwd + master_conv3x3_zero_fwd for that layer (no
 * zero-padded 32-channel copy of the images, 27 real taps instead of 288). */
int master_vgg_conv1_fwd(const float* img, const float* w, const float* bias, void* out, int N, int H, int W, int relu, int dtype,
                         master_stream_t stream);
int master_conv3x3_zero_fwd(const void* x, const void* w, const void* bias, void* out, int B, int H, int W, int Cin, int Cout,
                            int relu, int dtype, master_stream_t stream);
int master_maxpool2x2_fwd(const void* x, void* out, int N, int H, int W, int C, int dtype, master_stream_t stream);
int master_stats_slices(int HW);
int master_channel_stats_fwd(const void* x, float* mean, float* var, float* work, int N, int HW, int C, int dtype,
                             master_stream_t stream);
int master_instnorm_distance_fwd(const void* x, const float* mean, const float* var, float* partial, int pairs, int pair_offset,
                                 int HW, int C, float eps, int squared, int dtype, master_stream_t stream);

/* ---- pre/post-processing around the forward (reference test_model.py:39-48,110-125 and :207) -------------------------
 * master_resize_plan: HOST helper, no GPU work.  Pillow's bilinear resample tables for one axis (what
 *   transforms.Resize does to a PIL image): bounds[out][2] = (first source index, tap count), coeffs[out][ksize] = 22-bit
 *   fixed-point weights.  Call with bounds = coeffs = NULL to get ksize, allocate, call again; upload once per geometry.
 * master_preprocess_fwd: img uint8 [B,H,W,3] RGB -> out fp32 [B,3,OH,OW]: horizontal pass (skipped when OW == W, as Pillow
 *   does) into tmp uint8 [B,H,OW,3], vertical pass, ToTensor (/255), and Normalize ((x-mean)/std, host pointers to 3 floats
 *   each, both NULL = off).  Integer arithmetic identical to Pillow: bit-exact.  Tables are DEVICE pointers.
 * master_postprocess_fwd: img fp32 [B,3,H,W] -> out uint8 [B,H,W,3] = clip(x*255, 0, 255) truncated (np.clip(...).astype(uint8)). */
int master_resize_plan(int in_size, int out_size, int32_t* bounds, int32_t* coeffs, int* ksize);
int master_preprocess_fwd(const uint8_t* img, int B, int H, int W, float* out, int OH, int OW, const int32_t* h_bounds,
                          const int32_t* h_coeffs, int h_ksize, const int32_t* v_bounds, const int32_t* v_coeffs, int v_ksize,
                          uint8_t* tmp, const float* mean, const float* stdv, master_stream_t stream);
int master_postprocess_fwd(const float* img, uint8_t* out, int B, int H, int W, master_stream_t stream);

/* Element type conversion fp32 <-> bf16 for n elements (weights at load, activations at the edges). */
int master_convert(const void* src, void* dst, size_t n, int src_dtype, int dst_dtype, master_stream_t stream);

/* ======================================================================================================================================
 * Train step (reference train_only_inner_loop.py:519-575: train-mode forward, custom_loss, total_loss.backward(), Adam.step()).
 * The Swin encoder and VGG-19 are frozen (:254-257, loss.py:173-175): gradients flow through the VGG convolutions (data gradient
 * only), the image decoder and the StyleTransformer.  All entry points below are MASTER_F32 (the train step runs in the fp32 mode:
 * exact-fp32 matrix cores); MASTER_ERR_UNSUPPORTED for bf16.  Workspaces are caller-allocated device memory; every reduction adds its
 * partial sums in a fixed order, so gradients are bitwise reproducible.
 * ====================================================================================================================================== */

/* out = act(a . w^T + bias) like master_linear_fwd, and aux (optional, [M, ldo]) receives the PRE-activation a . w^T + bias, which the
 * GELU backward needs (torchvision MLP fc1, style_transformer.py:366). */
int master_linear_act_fwd(const void* a, const void* w, const void* bias, void* out, void* aux, int M, int N, int K, int lda, int ldo,
                          int act, int dtype, master_stream_t stream);

/* master_linear_act_fwd on a residual branch under torchvision StochasticDepth(p, "row") (style_transformer.py:390-397), one launch:
 *     out[m,:] = res[m,:] + rowscale[m / rows_per_scale] * act(a[m,:] . w^T + bias)        (res, rowscale optional = NULL)
 * rowscale: one fp32 factor per image (0 or 1/(1-p)), rows_per_scale = tokens per image.  In the backward the same factor applies to the
 * upstream gradient of the branch: master_linear_dgrad / master_linear_wgrad take it as `rowscale` (dgrad scales its output rows, wgrad
 * the rows of g it stages, and with them the bias gradient), so no scaled copy of g is ever written. */
int master_linear_res_fwd(const void* a, const void* w, const void* bias, const void* res, const float* rowscale, int rows_per_scale, void* out,
                          void* aux, int M, int N, int K, int lda, int ldo, int ldr, int act, int dtype, master_stream_t stream);

/* Data gradient of out = a . w^T:  da[M,K] = g[M,N] . w[N,K]; with gelu_pre != NULL the result is multiplied by gelu'(gelu_pre[M,K])
 * (the backward of the GELU that produced `a`); with add != NULL, add[M,K] (row stride ld_add) is added last - the gradient reaching `a`
 * over a skip connection.  N % 32 == 0, K % 4 == 0. */
int master_linear_dgrad(const void* g, const void* w, void* da, const void* gelu_pre, const void* add, const float* rowscale, int rows_per_scale,
                        int M, int N, int K, int ldg, int ldda, int ld_pre, int ld_add, int dtype, master_stream_t stream);

/* Weight gradient of out = a . w^T + bias:  dw[N,K] = g[M,N]^T . a[M,K] (dense, row stride K) and, when dbias != NULL, the bias gradient
 * dbias[N] = sum_m g[m][n] from the same pass over g (what autograd's Linear backward returns for weight and bias; MASTER_ACCUMULATE adds to
 * both).  The contraction over the M tokens is split across workgroups; ws holds the partial tiles and row sums:
 * master_gemm_workspace_floats(N, K) floats are always enough. */
size_t master_gemm_workspace_floats(int rows, int cols);
int master_linear_wgrad(const void* g, const void* a, void* dw, void* dbias, const float* rowscale, int rows_per_scale, int M, int N, int K, int ldg,
                        int lda, void* ws, size_t ws_floats, int dtype, master_stream_t stream);

/* out[n] = sum_m x[m][n]  (bias gradients).  N % 4 == 0; ws: master_colsum_workspace_floats(M, N) floats. */
size_t master_colsum_workspace_floats(int M, int N);
int master_colsum(const void* x, void* out, void* ws, int M, int N, int ld, int dtype, master_stream_t stream);

/* out[n] = sum_m x[m][n] * y[m][n]: the weight gradient of a per-channel affine (affine InstanceNorm).  Same workspace as master_colsum. */
int master_colsum_prod(const void* x, const void* y, void* out, void* ws, int M, int N, int ld, int dtype, master_stream_t stream);

/* Non-default StyleDecoder branches of the reference (both dtypes in the forward):
 * master_channel_affine_fwd: out[m,c] = x[m,c] * gamma[c] + beta[c] (beta optional) - the affine half of
 *   nn.InstanceNorm2d(dim, affine=True) (style_transformer.py:983-984) after master_instnorm_fwd; its data gradient is the same call
 *   with beta = NULL on the upstream gradient.
 * master_softmax_rows_fwd / _bwd: out[m,:] = softmax(scale * x[m,:]) over the first N columns of rows of stride ld (columns [N, ld)
 *   are row padding and are written as zeros; N % 4 == 0, ld % 4 == 0) and its backward - the global single-head
 *   attention of decoder_use_regular_MHA_instead_of_Swin_at_the_end (style_transformer.py:1100-1106); Q.K^T and P.V run on
 *   master_linear_fwd. */
int master_channel_affine_fwd(const void* x, const float* gamma, const float* beta, void* out, int M, int C, int dtype,
                              master_stream_t stream);
int master_softmax_rows_fwd(const void* x, void* out, int M, int N, int ld, float scale, int dtype, master_stream_t stream);
int master_softmax_rows_bwd(const void* p, const void* g, void* dx, int M, int N, int ld, float scale, int dtype, master_stream_t stream);

/* out[i] = sum_{k < splits} part[k * slab + i], i < n  (fixed order).  Used to add up per-window / per-workgroup partial gradients. */
int master_reduce_slabs(const float* part, float* out, size_t n, size_t slab, int splits, master_stream_t stream);
/* Up to four such reductions in ONE launch, each with its own split count; accumulate != 0: out[i] += the sum (a gradient that several
 * uses of one tensor contribute to).  The backward of one attention call reduces its bias-tile slabs and its pad-row gradients this way. */
typedef struct master_reduce_seg { const float* part; float* out; size_t n; size_t slab; int splits; int accumulate; } master_reduce_seg;
int master_reduce_segments(const master_reduce_seg* segs, int count, master_stream_t stream);

/* nn.LayerNorm backward: dx[M,C] (+ add[M,C] when add != NULL: the gradient that reaches x over the skip connection around the norm,
 * style_transformer.py:390-397 - autograd would add it with a separate elementwise launch); dgb[2][C] = dgamma | dbeta.
 * ws: master_layernorm_bwd_workspace_floats(M, C) floats. */
size_t master_layernorm_bwd_workspace_floats(int M, int C);
int master_layernorm_bwd(const void* x, const float* gamma, const void* g, const void* add, void* dx, float* dgb, float* ws, int M, int C, float eps,
                         int dtype, master_stream_t stream);

/* nn.InstanceNorm2d(affine=False) backward on a token-major map (forward: master_instnorm_fwd with repeat = 1).  With n_pad > 0 the
 * normalised set also holds n_pad copies of pad_row_in[c]; g_pad[b,c] is the sum of the upstream gradients of those copies and
 * dpad[b,c] receives the gradient w.r.t. pad_row_in from image b (the caller adds the images: it is a bias gradient). */
int master_instnorm_bwd(const void* x, const void* g, const void* pad_row_in, const void* g_pad, void* dx, void* dpad, int B, int T,
                        int C, int n_pad, float eps, int dtype, master_stream_t stream);

/* Backward of master_window_attn_fwd.  Inputs as in the forward plus
 *   pad_ord   : int32 [nW, 64]; for a zero-padded slot its ordinal among the image's n_pad padded slots, -1 otherwise;
 *   do0, do1  : upstream gradients of out0 / out1 [B*T, *] (row stride ldo).
 * Outputs: dq, dk, dv0, dv1 token matrices (row strides lddq/lddk/lddv; every real token row is written exactly once);
 *   dk_pad, dv0_pad, dv1_pad : [B, n_pad, heads*32] rows of the padded slots (their K/V rows are projection biases);
 *   dbias_ws  : [B*nW, heads, 64, 64] the dS tile of every (image, window, head); master_reduce_slabs over the B*nW slabs gives the
 *               gradient of the [heads,64,64] bias tile (NULL: skipped). */
int master_window_attn_bwd(const void* q, const void* k, const void* v0, const void* v1, int ldq, int ldk, int ldv, const void* k_pad,
                           const void* v0_pad, const void* v1_pad, int k_pad_bstride, int v_pad_bstride, const float* bias,
                           const int32_t* src, const uint8_t* region, const int32_t* pad_ord, const void* do0, const void* do1, int ldo,
                           void* dq, void* dk, void* dv0, void* dv1, int lddq, int lddk, int lddv, void* dk_pad, void* dv0_pad,
                           void* dv1_pad, float* dbias_ws, int B, int T, int heads, int nW, int N, int n_pad, float scale, int dtype,
                           master_stream_t stream);

/* Data gradient of a 3x3 convolution: a zero-padded correlation of g [B,H,W,Cg] with the flipped, transposed weights
 * wf [Cx][3][3][Cg] (wf[ci][ty][tx][co] = weight[co][ci][2-ty][2-tx]).  full = 0: out [B,H,W,Cx] (zero-padded convolution: VGG-19,
 * loss.py:23-37).  full = 1: out [B,H+2,W+2,Cx] = the gradient on the padded domain (reflect-padded decoder convolutions,
 * decoder.py:24-54), folded back by master_fold_reflect.  Optional epilogue: out = relu_src > 0 ? acc + add : 0  (relu_src, add:
 * [.., Cx] like out; the ReLU backward of the layer below and a gradient joining there).  Cg % 32 == 0.
 * dtype = MASTER_F32X3 (full = 0 only): g, relu_src, add, out fp32, wf = master_split_weights of the [Cx, 9*Cg] matrix. */
int master_conv3x3_dgrad(const void* g, const void* wf, void* out, const void* relu_src, const void* add, int B, int H, int W, int Cg,
                         int Cx, int full, int dtype, master_stream_t stream);

/* Weight gradient of the 3x3 convolutions of master_conv3x3_reflect_fwd / master_conv3x3_zero_fwd:
 * dw[Cout][3][3][Cin] = sum over output pixels of g[pix][Cout] (row stride ldg) x the gathered input patch (x: NHWC [B,Hin,Win,Cin],
 * same reflect / zero padding and upsample gather as the forward); dbias[Cout] = sum over the pixels of g (the bias gradient) when
 * dbias != NULL.  ws: master_gemm_workspace_floats(Cout, 9*Cin). */
int master_conv3x3_wgrad(const void* g, const void* x, void* dw, void* dbias, int B, int Hin, int Win, int Cin, int Cout, int ldg, int upsample,
                         int zero_pad, void* ws, size_t ws_floats, int dtype, master_stream_t stream);

/* gm[b,p,c] = (c < C and (relu_out == NULL or relu_out[b,p,c] > 0)) ? g[b,p,c] : 0 for c < Cp: ReLU backward + channel padding.
 * g is NHWC [B,HW,C] or (g_nchw) NCHW [B,C,HW]; gm is NHWC [B,HW,Cp]. */
int master_relu_mask(const void* g, const void* relu_out, void* gm, int B, int HW, int C, int Cp, int g_nchw, int dtype,
                     master_stream_t stream);

/* Backward of the decoder's LAST convolution, Conv2d(Cin = 32, 3, 3, padding_mode='reflect') without ReLU (decoder.py:54), straight from the
 * image gradient g in NCHW [B,3,H,W] and nn.Conv2d's own weight layout w [3][Cin][3][3] (no 32-channel padded copies):
 * _dgrad: dxp [B,H+2,W+2,Cin], the data gradient on the reflect-padded domain (fold it with master_fold_reflect);
 * _wgrad: dw [3][Cin][3][3] and dbias [3] (NULL: skipped) from the NHWC input x [B,H,W,Cin]; ws: _workspace_floats(B, H, W) floats. */
int master_conv3x3_cout3_dgrad(const float* g, const float* w, float* dxp, int B, int H, int W, int Cin, master_stream_t stream);
size_t master_conv3x3_cout3_wgrad_workspace_floats(int B, int H, int W);
int master_conv3x3_cout3_wgrad(const float* g, const float* x, float* dw, float* dbias, float* ws, size_t ws_floats, int B, int H, int W, int Cin,
                               master_stream_t stream);

/* Weight layouts of a trainable 3x3 convolution (decoder.py:24-54) from nn.Conv2d's w [Cout][Cin][3][3], one launch:
 * wp [Cout][3][3][Cin] (master_conv3x3_reflect_fwd / master_conv3x3_wgrad) and, when wf != NULL, wf [Cin][3][3][Cp] = taps flipped, channels
 * swapped, Cout zero-padded to Cp (master_conv3x3_dgrad). */
int master_conv_weight_pack(const float* w, float* wp, float* wf, int Cout, int Cin, int Cp, master_stream_t stream);

/* Relative-position bias tile of one attention module (style_transformer.py:21-28: relative_position_bias_table [entries, heads] gathered by
 * relative_position_index [N*N] int64) in the [heads,64,64] form the attention kernels read: tile[h][i][j] = table[index[i*N+j]][h] for
 * i, j < N, 0 for rows i >= N, -1e30 for key slots j >= N.  _bwd: dtable[e][h] = sum of dtile[h][i][j] over the positions with index e. */
int master_bias_tile_fwd(const float* table, const long long* index, float* tile, int N, int heads, master_stream_t stream);
int master_bias_tile_bwd(const float* dtile, const long long* index, float* dtable, int N, int heads, int entries, master_stream_t stream);

/* Backward of ReflectionPad(1) (+ nearest x2 upsample): dxp [B,H+2,W+2,C] (gradient on the padded H x W domain) -> out
 * [B, H>>upsample, W>>upsample, C].  C % 4 == 0, H, W >= 2. */
int master_fold_reflect(const void* dxp, void* out, int B, int H, int W, int C, int upsample, int dtype, master_stream_t stream);

/* nn.MaxPool2d(2,2) backward fused with the ReLU backward of the layer below: y [N,H,W,C] is that layer's ReLU output (the pool's
 * input), g [N,H/2,W/2,C]; the gradient goes to the first maximum of each 2x2 block (scan order) where y > 0. */
int master_maxpool2x2_bwd(const void* g, const void* y, void* dx, int N, int H, int W, int C, int dtype, master_stream_t stream);

/* NHWC [N,HW,Cp] -> fp32 NCHW [N,3,HW]: the image gradient out of the first (channel-padded) VGG convolution. */
int master_nhwc_to_image(const void* g, float* out, int N, int HW, int Cp, int dtype, master_stream_t stream);

/* torchvision StochasticDepth(p, "row") on a residual branch (style_transformer.py:390-397, :865-882, :1125):
 * out[m,:] = res[m,:] + scale[m / T] * x[m,:], scale [B] = bernoulli(1-p)/(1-p); res optional.  The backward of the branch is the
 * same call with res = NULL.  C % 4 == 0. */
int master_rowscale_add(const void* x, const void* res, const float* scale, void* out, int B, int T, int C, int dtype,
                        master_stream_t stream);

/* Query * sigma + mu (style_transformer.py:1123) and its backward (dq = g*sigma, dsigma = g*q; dmu = g).  n % 4 == 0. */
int master_mul_add_fwd(const void* q, const void* sigma, const void* mu, void* out, size_t n, int dtype, master_stream_t stream);
int master_mul_add_bwd(const void* g, const void* q, const void* sigma, void* dq, void* dsigma, size_t n, int dtype,
                       master_stream_t stream);

/* The scalar arithmetic of one tap of the perceptual loss (codes/loss.py:112-130 content term, :272-322 style term), from the
 * [3B,C] statistics of master_channel_stats_fwd (rows: content | style | output images) and the per-workgroup partial sums of
 * master_instnorm_distance_fwd (n_partial of them):
 *     losses[0] (+)= sum(dist_partial) / (B*C*HW)
 *     losses[1] (+)= mean_{i,c} f(mean_style - mean_out) + mean_{i,c} f(std_style - std_out)     f = |.| or, squared_style, (.)^2; unbiased std
 * ("first": overwrite instead of add - no memset needed), and the per-(image, channel) coefficients of the style term's gradient for a
 * unit upstream gradient: coefA = d/d mean_out / HW, coefB = d/d std_out / ((HW-1) std_out) (0 where std_out = 0). */
int master_loss_terms_fwd(const float* mean, const float* var, const float* dist_partial, int n_partial, float* losses, float* coefA, float* coefB,
                          int B, int HW, int C, int squared_style, int first, master_stream_t stream);

/* Perceptual-loss tap backward (codes/loss.py:112-130, 272-322).  x: features [3B,HW,C] of [content | style | output]; mean, var
 * [3B,C] from master_channel_stats_fwd; coefA, coefB [B,C] from master_loss_terms_fwd.  g_content, g_style: DEVICE pointers to the upstream
 * gradients of the two loss scalars (NULL = 0).  dx [B,HW,C] = gradient w.r.t. the OUTPUT image's features of
 *     g_content / (B*C*HW) * sum |IN(x_i) - IN(x_2B+i)| (squared: (.)^2)  +  g_style * sum_p ( coefA[i,c] + coefB[i,c] * (x_2B+i - mean) );
 * relu_gate: zero the result where x_2B+i <= 0.  ws: master_loss_tap_workspace_floats(B, HW, C) floats. */
size_t master_loss_tap_workspace_floats(int B, int HW, int C);
int master_loss_tap_bwd(const void* x, const float* mean, const float* var, const float* coefA, const float* coefB, float* ws, void* dx,
                        int B, int HW, int C, float eps, const float* g_content, const float* g_style, int squared, int relu_gate, int dtype,
                        master_stream_t stream);

/* Multi-tensor optimizer steps.  table: DEVICE array of ntensors records {float* param; const float* grad; float* exp_avg;
 * float* exp_avg_sq; int64 numel; int64 first_block} (6 x 8 bytes), first_block = running sum of ceil(numel / master_adam_chunk()),
 * nblocks = its total.  master_adam_step: torch.optim.Adam (train_only_inner_loop.py:468-478, :575), step is 1-based.
 * master_lerp: param += eta * (grad_slot - param) - the Reptile outer update theta += eta (omega - theta) (train.py:524-534) with
 * omega in the grad slot. */
int master_adam_chunk(void);
int master_adam_step(const void* table, int ntensors, long long nblocks, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, master_stream_t stream);
/* master_adam_step with its scalars in DEVICE memory, for a hipGraph capture of the whole train step (kernel arguments are baked into a
 * graph; the step count of the bias corrections is not a constant): hyper = {lr, beta1, beta2, eps, weight_decay} (5 floats), *step_done =
 * updates applied so far; the call uses step = *step_done + 1 and then increments the counter (a second one-thread launch). */
int master_adam_step_dev(const void* table, int ntensors, long long nblocks, const float* hyper, int* step_done, master_stream_t stream);
int master_lerp(const void* table, int ntensors, long long nblocks, float eta, master_stream_t stream);

/* Lightweight per-kernel timing used by bench.py's roofline leg: when enabled, every launch is
 * bracketed by hipEvents on the launch stream.  master_prof_read() synchronises those events and
 * returns, per kernel family, the launch count and the summed milliseconds. */
enum master_kernel_family {
    MASTER_K_LINEAR = 0, MASTER_K_CONV = 1, MASTER_K_ATTN = 2, MASTER_K_LAYERNORM = 3,
    MASTER_K_INSTNORM = 4, MASTER_K_PATCH_EMBED = 5, MASTER_K_PATCH_MERGE = 6, MASTER_K_CONVERT = 7,
    MASTER_K_MLP = 8, MASTER_K_DECODER_TAIL = 9, MASTER_K_ATTN_BLOCK = 10, MASTER_K_CONV_DGRAD = 11, MASTER_K_CONV_WGRAD = 12,
    MASTER_K_LINEAR_BWD = 13, MASTER_K_ATTN_BWD = 14, MASTER_K_TRAIN_MISC = 15, MASTER_K_COUNT = 16
};
int master_prof_enable(int on);
int master_prof_reset(void);
int master_prof_read(int family, int64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* MASTER_HIP_H */
