// HBM-bound pieces of the train step (reference train_only_inner_loop.py:519-575): bias gradients (column sums), the ReLU / reflect-pad /
// upsample / max-pool backward maps of the convolutions, StochasticDepth row scaling, the sigma*Query+mu combine and its backward, the
// perceptual-loss tap gradient, multi-tensor Adam and the Reptile interpolation.  fp32 (the train step runs in the fp32 mode).
// Every reduction adds partial sums in a fixed order: results are bitwise reproducible.
#include "common.h"

int reduce_slabs_launch(const float* part, float* out, size_t n, size_t slab, int splits, hipStream_t s, int accumulate = 0);

namespace {

// element access for the kernels that also serve a bf16 loss network (fp32 math either way)
template <typename T> struct El;
template <> struct El<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }
    static __device__ __forceinline__ void st4(float* p, f32x4 v) { *(f32x4*)p = v; }
};
template <> struct El<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
    static __device__ __forceinline__ f32x4 ld4(const bf16_t* p) {
        const uint2 u = *(const uint2*)p;
        return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
    }
    static __device__ __forceinline__ void st4(bf16_t* p, f32x4 v) { uint2 u; u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]); *(uint2*)p = u; }
};

// ---- slab reduce: out[i] = sum_k part[k*slab + i].  A workgroup owns 64 outputs; its 4 waves take every 4th slab (loads of one
//      wave are 64 consecutive floats), 8 independent partial sums per thread keep the loads in flight; the four wave partials are
//      added in a fixed order through LDS.  (r02: one thread per output walking 512 slabs serially took 41 us per bias gradient.) -------
struct ReduceSeg { const float* part; float* out; size_t n, slab; int splits, accumulate; unsigned block0; };
struct ReduceSegs { ReduceSeg s[4]; int count; };
__global__ __launch_bounds__(256) void reduce_slabs_kernel(ReduceSegs segs) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // up to four independent reductions in one launch (r03: a weight gradient with its bias gradient; the bias-tile and pad-row gradients of
    // one attention backward): a workgroup belongs to the last segment whose first block it has reached
    int which = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q) if (q < segs.count && blockIdx.x >= segs.s[q].block0) which = q;
    const ReduceSeg sg = segs.s[which];
    const float* __restrict__ part = sg.part;
    const size_t n = sg.n, slab = sg.slab;
    const int splits = sg.splits;
    const size_t i = (size_t)(blockIdx.x - sg.block0) * 64 + lane;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < n) {
        int k = wv;
        for (; k + 28 < splits; k += 32)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += part[(size_t)(k + 4 * u) * slab + i];
        for (int u = 0; k < splits; k += 4, ++u) acc[u & 7] += part[(size_t)k * slab + i];
    }
    red[wv][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    // accumulate: out += (r03) - the gradient of a tensor used several times in one graph (the k StyleTransformer layers share their
    // weights) is summed here instead of by one autograd add launch per use
    if (wv == 0 && i < n) sg.out[i] = (sg.accumulate ? sg.out[i] : 0.0f) + ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
}

// ---- column sums: part[slab][n] = sum over the slab's rows of x[m][n]   (bias gradients; N % 4 == 0).  tpr = min(N/4, 64) threads
//      cover a row chunk of 16-byte vectors, the other 256/tpr thread rows walk the slab's rows. ---------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ part, int M, int N,
                                                     int ld, int rows_per_slab, int tpr) {
    __shared__ f32x4 red[256];
    const int lanes = 256 / tpr;
    const int tx = threadIdx.x % tpr, ty = threadIdx.x / tpr;
    const int n = (blockIdx.x * tpr + tx) * 4;
    const int r0 = blockIdx.y * rows_per_slab, r1 = min(r0 + rows_per_slab, M);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (n < N && ty < lanes)
        for (int r = r0 + ty; r < r1; r += lanes) {
            const f32x4 v = *(const f32x4*)(x + (size_t)r * ld + n);
            acc += y ? v * *(const f32x4*)(y + (size_t)r * ld + n) : v;          // y: column sums of the product x*y (affine-norm weight gradient)
        }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (ty == 0 && n < N) {
        f32x4 t = red[tx];
        for (int k = 1; k < lanes; ++k) t += red[k * tpr + tx];
        *(f32x4*)(part + (size_t)blockIdx.y * N + n) = t;
    }
}

// ---- ReLU backward + layout: gm[b,y,x,c] = (c < C && (relu_out == NULL || relu_out[..c] > 0)) ? g : 0, Cp >= C output channels.
//      g is NHWC [.., C] or, with g_nchw, NCHW [B,C,H,W] (the gradient of the output image). ---------------------------------------------
__global__ __launch_bounds__(256) void relu_mask_kernel(const float* __restrict__ g, const float* __restrict__ relu_out, float* __restrict__ gm,
                                                        int HW, int C, int Cp, int g_nchw, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;             // (pixel, padded channel)
    if (i >= total) return;
    const int c = (int)(i % Cp); const size_t px = i / Cp;
    float v = 0.f;
    if (c < C) {
        const size_t n = px / HW, q = px - n * HW;
        v = g_nchw ? g[(n * C + c) * (size_t)HW + q] : g[px * C + c];
        if (relu_out && !(relu_out[px * C + c] > 0.f)) v = 0.f;
    }
    gm[i] = v;
}

// ---- backward of ReflectionPad(1) (+ nearest x2 upsample): dxp is the gradient on the padded domain [B,H+2,W+2,C] of the H x W map the
//      convolution read; pad row -1 mirrors row 1 and pad row H mirrors row H-2 (same for columns), and with `up` each source pixel
//      collects its 2x2 block.  out [B, H>>up, W>>up, C]. ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fold_reflect_kernel(const float* __restrict__ dxp, float* __restrict__ out, int H, int W, int C, int up,
                                                           size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;             // (source pixel, channel group of 4)
    if (i >= total) return;
    const int C4 = C >> 2;
    const int c = (int)(i % C4) * 4; size_t t = i / C4;
    const int Ws = W >> up, Hs = H >> up;
    const int sx = (int)(t % Ws); t /= Ws; const int sy = (int)(t % Hs); const size_t b = t / Hs;
    const int Wp = W + 2;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int reps = up ? 2 : 1;
    for (int uy = 0; uy < reps; ++uy) {
        const int Y = (sy << up) + uy;
        int ry[3], ny = 0;
        ry[ny++] = Y + 1;
        if (Y == 1) ry[ny++] = 0;
        if (Y == H - 2) ry[ny++] = H + 1;
        for (int ux = 0; ux < reps; ++ux) {
            const int X = (sx << up) + ux;
            int rx[3], nx = 0;
            rx[nx++] = X + 1;
            if (X == 1) rx[nx++] = 0;
            if (X == W - 2) rx[nx++] = W + 1;
            for (int a = 0; a < ny; ++a)
                for (int e = 0; e < nx; ++e)
                    acc += *(const f32x4*)(dxp + ((b * (H + 2) + ry[a]) * (size_t)Wp + rx[e]) * C + c);
        }
    }
    *(f32x4*)(out + i * 4) = acc;
}

// ---- MaxPool2d(2,2) backward fused with the ReLU backward of the layer below: y is that layer's ReLU output [N,H,W,C]; the gradient of
//      a pooled value goes to the first maximum of its 2x2 block (scan order, like torch) and only where y > 0. ---------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* __restrict__ g, const T* __restrict__ y, T* __restrict__ dx, int H,
                                                           int W, int C, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;             // (pooled pixel, channel)
    if (i >= total) return;
    const int c = (int)(i % C); size_t t = i / C;
    const int W2 = W >> 1, H2 = H >> 1;
    const int ox = (int)(t % W2); t /= W2; const int oy = (int)(t % H2); const size_t n = t / H2;
    const size_t base = ((n * H + 2 * oy) * (size_t)W + 2 * ox) * C + c;
    const size_t o[4] = {base, base + C, base + (size_t)W * C, base + (size_t)W * C + C};
    float best = El<T>::ld(y + o[0]); int arg = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) { const float v = El<T>::ld(y + o[k]); if (v > best) { best = v; arg = k; } }
    const float gv = best > 0.f ? El<T>::ld(g + i) : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) El<T>::st(dx + o[k], k == arg ? gv : 0.f);
}

// NHWC [N,H,W,Cp] -> NCHW [N,3,H,W] (the gradient of the image that image_to_nhwc padded to Cp channels)
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_image_kernel(const T* __restrict__ g, float* __restrict__ out, int HW, int Cp, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;             // (n, c, pixel)
    if (i >= total) return;
    const size_t q = i % HW, nc = i / HW, n = nc / 3; const int c = (int)(nc - n * 3);
    out[i] = El<T>::ld(g + (n * HW + q) * Cp + c);
}

// ---- StochasticDepth row mode:  out[m,:] = res[m,:] + scale[m / T] * x[m,:]   (res optional) ------------------------------------------
__global__ __launch_bounds__(256) void rowscale_add_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ scale,
                                                           float* __restrict__ out, int T, int C4, size_t total4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const size_t m = i / C4;
        const float sc = scale[m / T];
        f32x4 v = *(const f32x4*)(x + 4 * i) * sc;
        if (res) v += *(const f32x4*)(res + 4 * i);
        *(f32x4*)(out + 4 * i) = v;
    }
}

// ---- Query * sigma + mu (style_transformer.py:1123) and its backward ------------------------------------------------------------------
__global__ __launch_bounds__(256) void mul_add_kernel(const float* __restrict__ q, const float* __restrict__ sg, const float* __restrict__ mu,
                                                      float* __restrict__ out, size_t total4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256)
        *(f32x4*)(out + 4 * i) = *(const f32x4*)(q + 4 * i) * *(const f32x4*)(sg + 4 * i) + *(const f32x4*)(mu + 4 * i);
}
__global__ __launch_bounds__(256) void mul_add_bwd_kernel(const float* __restrict__ g, const float* __restrict__ q, const float* __restrict__ sg,
                                                          float* __restrict__ dq, float* __restrict__ dsg, size_t total4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const f32x4 gv = *(const f32x4*)(g + 4 * i);
        *(f32x4*)(dq + 4 * i) = gv * *(const f32x4*)(sg + 4 * i);
        *(f32x4*)(dsg + 4 * i) = gv * *(const f32x4*)(q + 4 * i);
    }
}

// ---- multi-tensor Adam (torch.optim.Adam semantics, train_only_inner_loop.py:468-478, :575) and Reptile interpolation -----------------
// table[t] = {param, grad, exp_avg, exp_avg_sq, numel, first_block}; a block finds its tensor by binary search over first_block.
struct TensorRec { float* p; const float* g; float* m; float* v; long long numel; long long first_block; };
constexpr int ADAM_CHUNK = 256 * 4 * 4;                                   // elements per block

__device__ __forceinline__ int find_tensor(const TensorRec* __restrict__ table, int n, long long block) {
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (table[mid].first_block <= block) lo = mid; else hi = mid - 1; }
    return lo;
}

__global__ __launch_bounds__(256) void adam_kernel(const TensorRec* __restrict__ table, int ntensors, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt) {
    const int t = find_tensor(table, ntensors, blockIdx.x);
    const TensorRec rec = table[t];
    const long long e0 = ((long long)blockIdx.x - rec.first_block) * ADAM_CHUNK;
    const float step = lr / bc1;
#pragma unroll
    for (int k = 0; k < ADAM_CHUNK / 256; ++k) {
        const long long i = e0 + k * 256 + threadIdx.x;
        if (i >= rec.numel) break;
        float p = rec.p[i], g = rec.g[i];
        if (wd != 0.f) g = fmaf(wd, p, g);
        const float m = fmaf(b1, rec.m[i], (1.f - b1) * g);
        const float v = fmaf(b2, rec.v[i], (1.f - b2) * g * g);
        rec.m[i] = m; rec.v[i] = v;
        rec.p[i] = p - step * (m / (sqrtf(v) / bc2_sqrt + eps));
    }
}

// The same update with its scalars in DEVICE memory (r03): hyper = {lr, beta1, beta2, eps, weight_decay}, *step = number of updates done so
// far.  A hipGraph capture of the whole train step bakes kernel arguments in; the step count of the bias corrections (and a scheduled
// learning rate) must therefore be read at run time.  adam_step_inc_kernel, launched behind it, advances the counter.
__global__ __launch_bounds__(256) void adam_dev_kernel(const TensorRec* __restrict__ table, int ntensors, const float* __restrict__ hyper,
                                                       const int* __restrict__ step_done) {
    __shared__ float sc[2];
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
    if (threadIdx.x == 0) {                                       // in double, like adam_launch (torch's Python floats)
        const double st = (double)(*step_done + 1);
        sc[0] = (float)(1.0 - pow((double)b1, st));
        sc[1] = (float)sqrt(1.0 - pow((double)b2, st));
    }
    __syncthreads();
    const float bc1 = sc[0], bc2_sqrt = sc[1];
    const int t = find_tensor(table, ntensors, blockIdx.x);
    const TensorRec rec = table[t];
    const long long e0 = ((long long)blockIdx.x - rec.first_block) * ADAM_CHUNK;
    const float step = lr / bc1;
#pragma unroll
    for (int k = 0; k < ADAM_CHUNK / 256; ++k) {
        const long long i = e0 + k * 256 + threadIdx.x;
        if (i >= rec.numel) break;
        float p = rec.p[i], g = rec.g[i];
        if (wd != 0.f) g = fmaf(wd, p, g);
        const float m = fmaf(b1, rec.m[i], (1.f - b1) * g);
        const float v = fmaf(b2, rec.v[i], (1.f - b2) * g * g);
        rec.m[i] = m; rec.v[i] = v;
        rec.p[i] = p - step * (m / (sqrtf(v) / bc2_sqrt + eps));
    }
}
__global__ void adam_step_inc_kernel(int* step_done) { *step_done += 1; }

// Reptile outer update (reference train.py:524-534): theta += eta * (omega - theta);  table: p = theta, g = omega
__global__ __launch_bounds__(256) void lerp_kernel(const TensorRec* __restrict__ table, int ntensors, float eta) {
    const int t = find_tensor(table, ntensors, blockIdx.x);
    const TensorRec rec = table[t];
    const long long e0 = ((long long)blockIdx.x - rec.first_block) * ADAM_CHUNK;
#pragma unroll
    for (int k = 0; k < ADAM_CHUNK / 256; ++k) {
        const long long i = e0 + k * 256 + threadIdx.x;
        if (i >= rec.numel) break;
        const float th = rec.p[i];
        rec.p[i] = fmaf(eta, rec.g[i] - th, th);
    }
}

// ---- perceptual-loss tap backward (codes/loss.py:112-130, 272-322): gradient w.r.t. the OUTPUT image's feature map ---------------------
// x holds the features of [content | style | output] images (3B maps of HW x C).  For output image j = 2B + i:
//   content term  mean |IN(x_i) - IN(x_j)| (or squared)   ->  gy[p,c] = -wc * sgn(yi - yj)  (or -wc * 2 (yi - yj)),  wc = g_content / (B*C*HW)
//                 and through InstanceNorm:  rstd_j * (gy - mean_p gy - yj * mean_p(gy * yj))
//   style term    through mean and unbiased std: coefA[i,c] + coefB[i,c] * (x_j - mean_j)     (coefficients from the [B,C] statistics)
// Pass 1 (MODE 0) writes per-slice partial sums of gy and gy*yj; pass 2 (MODE 1) adds them up per channel and writes the gradient,
// gated by x_j > 0 when relu_gate (the tap is a ReLU output and nothing else consumes it: relu5_1).
constexpr int SLICE_ROWS = 256;
template <typename T, int MODE>
__global__ __launch_bounds__(256) void loss_tap_bwd_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ var,
                                                           const float* __restrict__ coefA, const float* __restrict__ coefB, float* __restrict__ partial,
                                                           const float* __restrict__ sums, T* __restrict__ dx, int B, int HW, int C, float eps, float wc_unit,
                                                           const float* __restrict__ g_content, const float* __restrict__ g_style, int squared,
                                                           int relu_gate, int nslice) {
    extern __shared__ float red[];                              // MODE 0: [2][RPB][C]
    // upstream gradients of the two loss scalars, read on the device (r03: float(g) on the host drained the whole queue at the start of
    // every backward pass); a NULL pointer = that loss does not reach the objective
    const float wc = wc_unit * (g_content ? *g_content : 0.f), gs = g_style ? *g_style : 0.f;
    const int TPR = C / 4, RPB = 256 / TPR;
    const int tc = threadIdx.x % TPR, tr = threadIdx.x / TPR, i = blockIdx.y, j = 2 * B + i, sl = blockIdx.x;
    const int r0 = sl * SLICE_ROWS, r1 = min(r0 + SLICE_ROWS, HW);
    float mi[4], ri[4], mj[4], rj[4], s1[4], s2[4], ca[4], cb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = tc * 4 + e;
        mi[e] = mean[(size_t)i * C + c]; ri[e] = rsqrtf(var[(size_t)i * C + c] + eps);
        mj[e] = mean[(size_t)j * C + c]; rj[e] = rsqrtf(var[(size_t)j * C + c] + eps);
        s1[e] = 0.f; s2[e] = 0.f;
        if (MODE == 1) {                                           // sums[i][2][C]: the slice partials added up by loss_tap_finalize_kernel
            s1[e] = sums[((size_t)i * 2 + 0) * C + c]; s2[e] = sums[((size_t)i * 2 + 1) * C + c];
            ca[e] = gs * coefA[(size_t)i * C + c]; cb[e] = gs * coefB[(size_t)i * C + c];
        }
    }
    if (tr < RPB)
        for (int r = r0 + tr; r < r1; r += RPB) {
            const f32x4 va = El<T>::ld4(x + ((size_t)i * HW + r) * C + tc * 4);
            const f32x4 vb = El<T>::ld4(x + ((size_t)j * HW + r) * C + tc * 4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float yi = (va[e] - mi[e]) * ri[e], yj = (vb[e] - mj[e]) * rj[e], d = yi - yj;
                const float gy = squared ? -2.f * wc * d : (d > 0.f ? -wc : (d < 0.f ? wc : 0.f));
                if (MODE == 0) { s1[e] += gy; s2[e] += gy * yj; }
                else {
                    float gx = ca[e] + cb[e] * (vb[e] - mj[e]) + rj[e] * (gy - s1[e] - yj * s2[e]);
                    if (relu_gate && !(vb[e] > 0.f)) gx = 0.f;
                    o[e] = gx;
                }
            }
            if (MODE == 1) El<T>::st4(dx + ((size_t)i * HW + r) * C + tc * 4, o);
        }
    if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[tr * C + tc * 4 + e] = s1[e]; red[(RPB + tr) * C + tc * 4 + e] = s2[e]; }
        __syncthreads();
        for (int c = threadIdx.x; c < 2 * C; c += 256) {
            const int which = c / C, cc = c - which * C;
            float t = 0.f;
            for (int k = 0; k < RPB; ++k) t += red[(which * RPB + k) * C + cc];
            partial[(((size_t)i * nslice + sl) * 2 + which) * C + cc] = t;
        }
    }
}

// ---- weight layouts of a trainable 3x3 convolution, both in one launch (r03: permute/flip/pad/contiguous were four torch launches per
//      convolution and step):  w [Cout][Cin][3][3] (nn.Conv2d) ->  wp [Cout][3][3][Cin] for the forward and the weight gradient,
//      wf [Cin][3][3][Cp] = taps flipped, channels swapped, Cout zero-padded to Cp, for the data gradient (a convolution of g with wf) ----
__global__ __launch_bounds__(256) void conv_weight_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, float* __restrict__ wf, int Cout,
                                                               int Cin, int Cp, size_t n_wp, size_t n_wf) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_wp) {
        const int ci = (int)(i % Cin), tap = (int)((i / Cin) % 9), co = (int)(i / ((size_t)9 * Cin));
        wp[i] = w[((size_t)co * Cin + ci) * 9 + tap];
    }
    if (wf && i < n_wf) {
        const int co = (int)(i % Cp), tap = (int)((i / Cp) % 9), ci = (int)(i / ((size_t)9 * Cp));
        wf[i] = co < Cout ? w[((size_t)co * Cin + ci) * 9 + (8 - tap)] : 0.f;
    }
}

// ---- relative-position bias tile (style_transformer.py:21-28 + the key-slot padding of the attention kernels):
//      tile[h][i][j] = table[index[i*N + j]][h] for i, j < N; 0 for i >= N; -1e30 for j >= N (padded key slots never win the softmax).
//      Backward: dtable[e][h] = sum over the (i, j) with index[i*N + j] == e of dtile[h][i][j] - a gather over the N*N positions per
//      table entry (deterministic; torch's index_put backward sorts the indices: ~10 launches). ----------------------------------------
__global__ __launch_bounds__(256) void bias_tile_fwd_kernel(const float* __restrict__ table, const long long* __restrict__ index, float* __restrict__ tile,
                                                            int N, int heads) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= heads * 4096) return;
    const int j = t & 63, i = (t >> 6) & 63, h = t >> 12;
    float v = 0.f;
    if (j >= N) v = -1e30f;
    else if (i < N) v = table[(size_t)index[i * N + j] * heads + h];
    tile[t] = v;
}
// one workgroup per table entry: its threads walk the N*N positions once (index compared in registers), then one block sum per head
__global__ __launch_bounds__(256) void bias_tile_bwd_kernel(const float* __restrict__ dtile, const long long* __restrict__ index, float* __restrict__ dtable,
                                                            int N, int heads, int entries) {
    __shared__ float red[4];
    const int e = blockIdx.x, tid = threadIdx.x, NN = N * N;
    int off[16];                                              // tile offsets of this thread's matching positions (N <= 64: at most 16 per thread), -1 = none
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int p = tid + 256 * q;
        off[q] = (p < NN && index[p] == e) ? (p / N) * 64 + (p % N) : -1;
    }
    for (int h = 0; h < heads; ++h) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) if (off[q] >= 0) s += dtile[(size_t)h * 4096 + off[q]];
        s = wave_sum(s);
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) dtable[(size_t)e * heads + h] = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
    }
}

// ---- the scalar arithmetic of one tap of the perceptual loss (codes/loss.py:112-130 content term, :272-322 style term), one workgroup:
//   losses[0] (+)= sum(dist_partial) / (B*C*HW)                                  (the |IN - IN| partials of master_instnorm_distance_fwd)
//   losses[1] (+)= mean_{i,c} f(mean_s - mean_o) + mean_{i,c} f(std_s - std_o)    f = |.| or (.)^2, std unbiased
//   coefA[i,c] = d style / d mean_o / HW,  coefB[i,c] = d style / d std_o / ((HW-1) std_o)  (0 where std_o = 0: torch's sqrt'(0) guard
//   of the reference's backward) - what master_loss_tap_bwd adds per position, for a unit upstream gradient.
// r03: this was ~30 torch elementwise launches per tap and pass on [B,C] tensors.  Sums in a fixed order (deterministic).
__global__ __launch_bounds__(256) void loss_terms_kernel(const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ dist_partial,
                                                         int n_partial, float* __restrict__ losses, float* __restrict__ coefA, float* __restrict__ coefB,
                                                         int B, int HW, int C, int squared_s, int first) {
    __shared__ float red[2][256];
    const int tid = threadIdx.x, n = B * C;
    const float n1 = (float)HW / (float)(HW - 1), cnt = (float)n;
    float st = 0.f, ct = 0.f;
    for (int i = tid; i < n; i += 256) {
        const float ms = mean[n + i], mo = mean[2 * n + i];
        const float ss = sqrtf(var[n + i] * n1), so = sqrtf(var[2 * n + i] * n1);
        const float dm = ms - mo, ds = ss - so;
        float dmu, dsg;
        if (squared_s) { st += dm * dm + ds * ds; dmu = -2.f * dm / cnt; dsg = -2.f * ds / cnt; }
        else {
            st += fabsf(dm) + fabsf(ds);
            dmu = (dm > 0.f ? -1.f : (dm < 0.f ? 1.f : 0.f)) / cnt; dsg = (ds > 0.f ? -1.f : (ds < 0.f ? 1.f : 0.f)) / cnt;
        }
        coefA[i] = dmu / (float)HW;
        coefB[i] = so > 0.f ? dsg / ((float)(HW - 1) * so) : 0.f;
    }
    for (int i = tid; i < n_partial; i += 256) ct += dist_partial[i];
    red[0][tid] = st; red[1][tid] = ct;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) { red[0][tid] += red[0][tid + w]; red[1][tid] += red[1][tid + w]; }
        __syncthreads();
    }
    if (tid == 0) {
        const float c_term = red[1][0] / ((float)n * (float)HW), s_term = red[0][0] / cnt;
        losses[0] = (first ? 0.f : losses[0]) + c_term;
        losses[1] = (first ? 0.f : losses[1]) + s_term;
    }
}

// sums[i][which][c] = (sum over slices of partial[i][slice][which][c]) / HW
__global__ __launch_bounds__(256) void loss_tap_finalize_kernel(const float* __restrict__ partial, float* __restrict__ sums, int HW, int C, int nslice,
                                                                int total) {
    const int t = blockIdx.x * 256 + threadIdx.x;               // (i, which, c)
    if (t >= total) return;
    const int c = t % C, iw = t / C, which = iw & 1, i = iw >> 1;
    float s = 0.f;
    for (int k = 0; k < nslice; ++k) s += partial[(((size_t)i * nslice + k) * 2 + which) * C + c];
    sums[t] = s / (float)HW;
}

// ---- per-channel affine of an affine InstanceNorm2d (style_transformer.py:983-984): out[m,c] = x[m,c] * gamma[c] + beta[c] (beta optional) -----
template <typename T>
__global__ __launch_bounds__(256) void channel_affine_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             T* __restrict__ out, int C4, size_t total4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        f32x4 v = El<T>::ld4(x + 4 * i) * *(const f32x4*)(gamma + c);
        if (beta) v += *(const f32x4*)(beta + c);
        El<T>::st4(out + 4 * i, v);
    }
}

// ---- row softmax of the regular-MHA decoder end (style_transformer.py:1104-1106): out[m,:] = softmax(scale * x[m,:]); one wave per row ----------
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const T* __restrict__ x, T* __restrict__ out, int M, int N, int ld, float scale_log2e) {
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
        const T* xr = x + (size_t)row * ld; T* orow = out + (size_t)row * ld;
        float mx = -3.0e38f;
        for (int c = lane * 4; c < N; c += 256) { const f32x4 v = El<T>::ld4(xr + c); mx = fmaxf(mx, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]))); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sum = 0.f;
        for (int c = lane * 4; c < N; c += 256) {
            const f32x4 v = El<T>::ld4(xr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += exp2f((v[e] - mx) * scale_log2e);
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        for (int c = lane * 4; c < ld; c += 256) {                 // columns [N, ld) are padding of the row: written as zeros
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if (c < N) {
                const f32x4 v = El<T>::ld4(xr + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = exp2f((v[e] - mx) * scale_log2e) * inv;
            }
            El<T>::st4(orow + c, o);
        }
    }
}
// dx[m,:] = scale * p[m,:] * (g[m,:] - sum_j g[m,j] p[m,j])      (p = the softmax output)
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ p, const float* __restrict__ g, float* __restrict__ dx, int M,
                                                               int N, int ld, float scale) {
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
        const float* pr = p + (size_t)row * ld; const float* gr = g + (size_t)row * ld;
        float dot = 0.f;
        for (int c = lane * 4; c < N; c += 256) { const f32x4 a = *(const f32x4*)(pr + c), b = *(const f32x4*)(gr + c); dot += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
        dot = wave_sum(dot);
        for (int c = lane * 4; c < ld; c += 256) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if (c < N) { const f32x4 a = *(const f32x4*)(pr + c), b = *(const f32x4*)(gr + c); o = a * (b - dot) * scale; }
            *(f32x4*)(dx + (size_t)row * ld + c) = o;
        }
    }
}

inline unsigned blocks_for(size_t n, unsigned cap = 0x7fffffffu) { size_t b = (n + 255) / 256; return (unsigned)(b > cap ? cap : (b ? b : 1)); }

}  // namespace

// segs: { part, out, n, slab, splits, accumulate } x count (<= 4); out[i] (+)= sum_k part[k*slab + i] for each
int reduce_segments_launch(const float* const* part, float* const* out, const size_t* n, const size_t* slab, const int* splits, const int* accumulate,
                           int count, hipStream_t s) {
    ReduceSegs segs{};
    unsigned blocks = 0;
    for (int q = 0; q < count; ++q) {
        segs.s[q] = ReduceSeg{part[q], out[q], n[q], slab[q], splits[q], accumulate[q], blocks};
        blocks += (unsigned)((n[q] + 63) / 64);
    }
    segs.count = count;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, s, segs);
    return master_check_launch("reduce_slabs");
}
int reduce_slabs_launch(const float* part, float* out, size_t n, size_t slab, int splits, hipStream_t s, int accumulate) {
    return reduce_segments_launch(&part, &out, &n, &slab, &splits, &accumulate, 1, s);
}
// two reductions with the same split count in one launch: out[i] (+)= sum_k part[k*slab + i], out2[j] (+)= sum_k part2[k*slab2 + j]
int reduce_slabs2_launch(const float* part, float* out, size_t n, size_t slab, const float* part2, float* out2, size_t n2, size_t slab2, int splits,
                         hipStream_t s, int accumulate) {
    const float* parts[2] = {part, part2}; float* outs[2] = {out, out2};
    const size_t ns[2] = {n, n2}, slabs[2] = {slab, slab2};
    const int sp[2] = {splits, splits}, ac[2] = {accumulate, accumulate};
    return reduce_segments_launch(parts, outs, ns, slabs, sp, ac, 2, s);
}

static int colsum_tpr(int N) { return N / 4 >= 64 ? 64 : N / 4; }
int colsum_slabs(int M, int N) {                      // row slabs such that the grid has ~512 workgroups, >= 64 rows each
    const int tpr = colsum_tpr(N), gx = (N / 4 + tpr - 1) / tpr;
    int slabs = (512 + gx - 1) / gx;
    if (slabs > (M + 63) / 64) slabs = (M + 63) / 64;
    return slabs < 1 ? 1 : slabs;
}
// ws: colsum_slabs(M, N) * N floats
int colsum_launch(const float* x, const float* y, float* out, float* ws, int M, int N, int ld, hipStream_t s, int accumulate) {
    const int slabs = colsum_slabs(M, N), rps = (M + slabs - 1) / slabs, tpr = colsum_tpr(N);
    const bool direct = slabs == 1 && !accumulate;            // accumulate: always through the slab reduce, which adds to out
    hipLaunchKernelGGL(colsum_kernel, dim3((N / 4 + tpr - 1) / tpr, slabs), dim3(256), 0, s, x, y, direct ? out : ws, M, N, ld, rps, tpr);
    if (!direct) return reduce_slabs_launch(ws, out, (size_t)N, (size_t)N, slabs, s, accumulate);
    return master_check_launch("colsum");
}
int relu_mask_launch(const float* g, const float* relu_out, float* gm, int B, int HW, int C, int Cp, int g_nchw, hipStream_t s) {
    const size_t total = (size_t)B * HW * Cp;
    hipLaunchKernelGGL(relu_mask_kernel, dim3(blocks_for(total)), dim3(256), 0, s, g, relu_out, gm, HW, C, Cp, g_nchw, total);
    return master_check_launch("relu_mask");
}
int conv_weight_pack_launch(const float* w, float* wp, float* wf, int Cout, int Cin, int Cp, hipStream_t s) {
    const size_t n_wp = (size_t)Cout * 9 * Cin, n_wf = wf ? (size_t)Cin * 9 * Cp : 0, total = n_wp > n_wf ? n_wp : n_wf;
    hipLaunchKernelGGL(conv_weight_pack_kernel, dim3(blocks_for(total)), dim3(256), 0, s, w, wp, wf, Cout, Cin, Cp, n_wp, n_wf);
    return master_check_launch("conv_weight_pack");
}
int bias_tile_fwd_launch(const float* table, const long long* index, float* tile, int N, int heads, hipStream_t s) {
    hipLaunchKernelGGL(bias_tile_fwd_kernel, dim3((heads * 4096 + 255) / 256), dim3(256), 0, s, table, index, tile, N, heads);
    return master_check_launch("bias_tile_fwd");
}
int bias_tile_bwd_launch(const float* dtile, const long long* index, float* dtable, int N, int heads, int entries, hipStream_t s) {
    hipLaunchKernelGGL(bias_tile_bwd_kernel, dim3(entries), dim3(256), 0, s, dtile, index, dtable, N, heads, entries);
    return master_check_launch("bias_tile_bwd");
}
int fold_reflect_launch(const float* dxp, float* out, int B, int H, int W, int C, int up, hipStream_t s) {
    const size_t total = (size_t)B * (H >> up) * (W >> up) * (C / 4);
    hipLaunchKernelGGL(fold_reflect_kernel, dim3(blocks_for(total)), dim3(256), 0, s, dxp, out, H, W, C, up, total);
    return master_check_launch("fold_reflect");
}
template <typename T> int maxpool2_bwd_launch(const void* g, const void* y, void* dx, int N, int H, int W, int C, hipStream_t s) {
    const size_t total = (size_t)N * (H / 2) * (W / 2) * C;
    // odd H or W (floor mode, like torch's MaxPool2d): the last row / column belongs to no 2x2 block and takes a zero gradient; the kernel
    // writes only the blocks, so dx is cleared first on this rare path (r05: an 88 x 136 image reaches VGG-19's fourth pool as 11 x 17)
    if ((H | W) & 1) (void)hipMemsetAsync(dx, 0, (size_t)N * H * W * C * sizeof(T), s);
    hipLaunchKernelGGL((maxpool2_bwd_kernel<T>), dim3(blocks_for(total)), dim3(256), 0, s, (const T*)g, (const T*)y, (T*)dx, H, W, C, total);
    return master_check_launch("maxpool2_bwd");
}
template int maxpool2_bwd_launch<float>(const void*, const void*, void*, int, int, int, int, hipStream_t);
template int maxpool2_bwd_launch<bf16_t>(const void*, const void*, void*, int, int, int, int, hipStream_t);
template <typename T> int nhwc_to_image_launch(const void* g, float* out, int N, int HW, int Cp, hipStream_t s) {
    const size_t total = (size_t)N * 3 * HW;
    hipLaunchKernelGGL((nhwc_to_image_kernel<T>), dim3(blocks_for(total)), dim3(256), 0, s, (const T*)g, out, HW, Cp, total);
    return master_check_launch("nhwc_to_image");
}
template int nhwc_to_image_launch<float>(const void*, float*, int, int, int, hipStream_t);
template int nhwc_to_image_launch<bf16_t>(const void*, float*, int, int, int, hipStream_t);
int rowscale_add_launch(const float* x, const float* res, const float* scale, float* out, int B, int T, int C, hipStream_t s) {
    const size_t total4 = (size_t)B * T * (C / 4);
    hipLaunchKernelGGL(rowscale_add_kernel, dim3(blocks_for(total4, 2048)), dim3(256), 0, s, x, res, scale, out, T, C / 4, total4);
    return master_check_launch("rowscale_add");
}
int mul_add_launch(const float* q, const float* sg, const float* mu, float* out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(mul_add_kernel, dim3(blocks_for(n / 4, 2048)), dim3(256), 0, s, q, sg, mu, out, n / 4);
    return master_check_launch("mul_add");
}
int mul_add_bwd_launch(const float* g, const float* q, const float* sg, float* dq, float* dsg, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(mul_add_bwd_kernel, dim3(blocks_for(n / 4, 2048)), dim3(256), 0, s, g, q, sg, dq, dsg, n / 4);
    return master_check_launch("mul_add_bwd");
}
int adam_chunk() { return ADAM_CHUNK; }
int adam_launch(const void* table, int ntensors, long long nblocks, float lr, float b1, float b2, float eps, float wd, int step, hipStream_t s) {
    const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);     // in double, like torch's Python floats
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, (const TensorRec*)table, ntensors, lr, b1, b2, eps, wd, (float)bc1,
                       (float)sqrt(bc2));
    return master_check_launch("adam");
}
int adam_dev_launch(const void* table, int ntensors, long long nblocks, const float* hyper, int* step_done, hipStream_t s) {
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, (const TensorRec*)table, ntensors, hyper, (const int*)step_done);
    hipLaunchKernelGGL(adam_step_inc_kernel, dim3(1), dim3(1), 0, s, step_done);
    return master_check_launch("adam_dev");
}
int lerp_launch(const void* table, int ntensors, long long nblocks, float eta, hipStream_t s) {
    hipLaunchKernelGGL(lerp_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, (const TensorRec*)table, ntensors, eta);
    return master_check_launch("lerp");
}
int loss_tap_slices(int HW) { return (HW + SLICE_ROWS - 1) / SLICE_ROWS; }
// partial: B * (slices + 1) * 2 * C floats of workspace (slice partials, then their sums); dx: [B, HW, C]
template <typename T>
int loss_tap_bwd_launch(const void* xv, const float* mean, const float* var, const float* coefA, const float* coefB, float* partial, void* dxv,
                        int B, int HW, int C, float eps, float wc_unit, const float* g_content, const float* g_style, int squared, int relu_gate,
                        hipStream_t s) {
    const T* x = (const T*)xv; T* dx = (T*)dxv;
    const int ns = loss_tap_slices(HW);
    float* sums = partial + (size_t)B * ns * 2 * C;
    const size_t lds = (size_t)2 * (256 / (C / 4)) * C * sizeof(float);
    hipLaunchKernelGGL((loss_tap_bwd_kernel<T, 0>), dim3(ns, B), dim3(256), lds, s, x, mean, var, coefA, coefB, partial, (const float*)nullptr, dx, B, HW,
                       C, eps, wc_unit, g_content, g_style, squared, relu_gate, ns);
    hipLaunchKernelGGL(loss_tap_finalize_kernel, dim3((B * 2 * C + 255) / 256), dim3(256), 0, s, (const float*)partial, sums, HW, C, ns, B * 2 * C);
    hipLaunchKernelGGL((loss_tap_bwd_kernel<T, 1>), dim3(ns, B), dim3(256), 0, s, x, mean, var, coefA, coefB, partial, (const float*)sums, dx, B, HW, C,
                       eps, wc_unit, g_content, g_style, squared, relu_gate, ns);
    return master_check_launch("loss_tap_bwd");
}
template int loss_tap_bwd_launch<float>(const void*, const float*, const float*, const float*, const float*, float*, void*, int, int, int, float, float, const float*, const float*, int, int, hipStream_t);
template int loss_tap_bwd_launch<bf16_t>(const void*, const float*, const float*, const float*, const float*, float*, void*, int, int, int, float, float, const float*, const float*, int, int, hipStream_t);
int loss_terms_launch(const float* mean, const float* var, const float* dist_partial, int n_partial, float* losses, float* coefA, float* coefB, int B,
                      int HW, int C, int squared_s, int first, hipStream_t s) {
    hipLaunchKernelGGL(loss_terms_kernel, dim3(1), dim3(256), 0, s, mean, var, dist_partial, n_partial, losses, coefA, coefB, B, HW, C, squared_s, first);
    return master_check_launch("loss_terms");
}
template <typename T> int channel_affine_launch(const void* x, const float* gamma, const float* beta, void* out, int M, int C, hipStream_t s) {
    const size_t total4 = (size_t)M * (C / 4);
    hipLaunchKernelGGL((channel_affine_kernel<T>), dim3(blocks_for(total4, 2048)), dim3(256), 0, s, (const T*)x, gamma, beta, (T*)out, C / 4, total4);
    return master_check_launch("channel_affine");
}
template int channel_affine_launch<float>(const void*, const float*, const float*, void*, int, int, hipStream_t);
template int channel_affine_launch<bf16_t>(const void*, const float*, const float*, void*, int, int, hipStream_t);
template <typename T> int softmax_rows_launch(const void* x, void* out, int M, int N, int ld, float scale, hipStream_t s) {
    int blocks = (M + 3) / 4; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((softmax_rows_kernel<T>), dim3(blocks), dim3(256), 0, s, (const T*)x, (T*)out, M, N, ld, scale * 1.4426950408889634f);
    return master_check_launch("softmax_rows");
}
template int softmax_rows_launch<float>(const void*, void*, int, int, int, float, hipStream_t);
template int softmax_rows_launch<bf16_t>(const void*, void*, int, int, int, float, hipStream_t);
int softmax_rows_bwd_launch(const float* p, const float* g, float* dx, int M, int N, int ld, float scale, hipStream_t s) {
    int blocks = (M + 3) / 4; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3(blocks), dim3(256), 0, s, p, g, dx, M, N, ld, scale);
    return master_check_launch("softmax_rows_bwd");
}
