// Backward of the two normalisations on the train step (fp32): nn.LayerNorm (style_transformer.py:340-343, :390-392) and
// nn.InstanceNorm2d(affine=False) over the tokens of an image (:468, :1056-1057, and :520-530 where the statistics also run over the
// bias-only rows of the zero-padded map).  Statistics are recomputed from x (two-pass, like the forward); HBM-bound.
#include "common.h"

int reduce_slabs_launch(const float* part, float* out, size_t n, size_t slab, int splits, hipStream_t s, int accumulate = 0);

namespace {

// ---- LayerNorm backward.  A wave owns a row at a time; lane l holds the 16-byte chunks l, l+64, ... of the row.
//      dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat));  dgamma = sum_rows g*xhat;  dbeta = sum_rows g.
//      part[block][2][C]: per-workgroup partial dgamma | dbeta (added by reduce_slabs in block order). -------------------------------------
template <int CH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ g,
                                                            float* __restrict__ dx, float* __restrict__ part, int M, int C, float eps,
                                                            const float* __restrict__ add) {
    extern __shared__ float red[];                                // [4][2][C]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nchunk = C >> 2;
    f32x4 gam[CH], adg[CH], adb[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c4 = lane + 64 * i;
        gam[i] = c4 < nchunk ? *(const f32x4*)(gamma + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
        adg[i] = f32x4{0.f, 0.f, 0.f, 0.f}; adb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float invC = 1.0f / (float)C;
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        f32x4 xv[CH], gv[CH];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c4 = lane + 64 * i;
            const bool on = c4 < nchunk;
            xv[i] = on ? *(const f32x4*)(x + (size_t)row * C + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
            gv[i] = on ? *(const f32x4*)(g + (size_t)row * C + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
            sum += xv[i][0] + xv[i][1] + xv[i][2] + xv[i][3];
        }
        const float mean = wave_sum(sum) * invC;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (lane + 64 * i < nchunk)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = xv[i][e] - mean; sq += d * d; }
        const float rstd = rsqrtf(wave_sum(sq) * invC + eps);
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xh = (xv[i][e] - mean) * rstd, gg = gv[i][e] * gam[i][e];
                xv[i][e] = xh;                                    // keep xhat
                a += gg; b += gg * xh;
            }
        a = wave_sum(a) * invC; b = wave_sum(b) * invC;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < nchunk) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = rstd * (gv[i][e] * gam[i][e] - a - xv[i][e] * b);
                    adg[i][e] += gv[i][e] * xv[i][e];
                    adb[i][e] += gv[i][e];
                }
                if (add) o += *(const f32x4*)(add + (size_t)row * C + 4 * c4);       // + the gradient arriving over the skip connection around this norm
                *(f32x4*)(dx + (size_t)row * C + 4 * c4) = o;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c4 = lane + 64 * i;
        if (c4 < nchunk) {
            *(f32x4*)(red + (wave * 2 + 0) * C + 4 * c4) = adg[i];
            *(f32x4*)(red + (wave * 2 + 1) * C + 4 * c4) = adb[i];
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * C; t += 256)
        part[(size_t)blockIdx.x * 2 * C + t] = red[t] + red[2 * C + t] + red[4 * C + t] + red[6 * C + t];
}

// ---- InstanceNorm backward.  Workgroup = (image b, 32-channel strip); thread (token group tg of 32, 4-channel vector cq of 8).
//      The normalised set is the T real rows plus n_pad copies of pad_in (one value per channel); g_pad[b,c] is the SUM of the upstream
//      gradients of those copies.  With N = T + n_pad:
//        dx_t  = rstd * (g_t - S1/N - y_t * S2/N),          S1 = sum g (all N),  S2 = sum g*y (all N)
//        dpad  = rstd * (g_pad - n_pad * S1/N - n_pad * y_pad * S2/N)         (per image; the caller adds the images up) ----------------------
template <int REG>
__global__ __launch_bounds__(256) void instnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ pad_in,
                                                           const float* __restrict__ g_pad, float* __restrict__ dx, float* __restrict__ dpad, int Tn,
                                                           int C, int n_pad, float eps) {
    __shared__ f32x4 red[32][8];
    const int strips = C >> 5;
    const int b = blockIdx.x / strips, c0 = (blockIdx.x % strips) * 32;
    const int cq = threadIdx.x & 7, tg = threadIdx.x >> 3;
    const size_t base = (size_t)b * Tn * C + c0 + cq * 4;
    const float cnt = (float)(Tn + n_pad), np = (float)n_pad;
    f32x4 pr = {0.f, 0.f, 0.f, 0.f}, gp = {0.f, 0.f, 0.f, 0.f};
    if (n_pad > 0) {
        pr = *(const f32x4*)(pad_in + c0 + cq * 4);
        if (g_pad) gp = *(const f32x4*)(g_pad + (size_t)b * C + c0 + cq * 4);
    }
    auto block_sum = [&](f32x4 v) -> f32x4 {
        red[tg][cq] = v;
        __syncthreads();
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < 32; ++i) s += red[i][cq];
        __syncthreads();
        return s;
    };
    // REG > 0 (maps of up to 32 REG tokens: the 32x32 maps of the train step): x and g are read ONCE into registers (2 REG float4
    // per thread) instead of four and two times from memory - with 64 workgroups on 256 CUs the kernel is a chain of latency-bound
    // passes, so the passes that disappear are its time (r03: 40.6 -> see profiles/r03_train_trace_by_shape.txt).
    f32x4 xs[REG > 0 ? REG : 1], gs[REG > 0 ? REG : 1];
    if (REG > 0) {
#pragma unroll
        for (int i = 0; i < REG; ++i) {
            const int t = tg + 32 * i;
            xs[i] = gs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < Tn) { xs[i] = *(const f32x4*)(x + base + (size_t)t * C); gs[i] = *(const f32x4*)(g + base + (size_t)t * C); }
        }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (REG > 0) {
#pragma unroll
        for (int i = 0; i < REG; ++i) acc += xs[i];                       // rows past Tn hold zeros
    } else {
        for (int t = tg; t < Tn; t += 32) acc += *(const f32x4*)(x + base + (size_t)t * C);
    }
    f32x4 mean = (block_sum(acc) + pr * np) / cnt;
    acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (REG > 0) {
#pragma unroll
        for (int i = 0; i < REG; ++i) if (tg + 32 * i < Tn) { const f32x4 d = xs[i] - mean; acc += d * d; }
    } else {
        for (int t = tg; t < Tn; t += 32) { const f32x4 d = *(const f32x4*)(x + base + (size_t)t * C) - mean; acc += d * d; }
    }
    const f32x4 dp = pr - mean;
    const f32x4 var = (block_sum(acc) + dp * dp * np) / cnt;
    f32x4 rstd;
#pragma unroll
    for (int e = 0; e < 4; ++e) rstd[e] = rsqrtf(var[e] + eps);
    const f32x4 ypad = dp * rstd;
    f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
    if (REG > 0) {
#pragma unroll
        for (int i = 0; i < REG; ++i) if (tg + 32 * i < Tn) { const f32x4 y = (xs[i] - mean) * rstd; a1 += gs[i]; a2 += gs[i] * y; }
    } else {
        for (int t = tg; t < Tn; t += 32) {
            const f32x4 gv = *(const f32x4*)(g + base + (size_t)t * C);
            const f32x4 y = (*(const f32x4*)(x + base + (size_t)t * C) - mean) * rstd;
            a1 += gv; a2 += gv * y;
        }
    }
    const f32x4 s1 = (block_sum(a1) + gp) / cnt, s2 = (block_sum(a2) + gp * ypad) / cnt;
    if (REG > 0) {
#pragma unroll
        for (int i = 0; i < REG; ++i) {
            const int t = tg + 32 * i;
            if (t < Tn) { const f32x4 y = (xs[i] - mean) * rstd; *(f32x4*)(dx + base + (size_t)t * C) = rstd * (gs[i] - s1 - y * s2); }
        }
    } else {
        for (int t = tg; t < Tn; t += 32) {
            const f32x4 gv = *(const f32x4*)(g + base + (size_t)t * C);
            const f32x4 y = (*(const f32x4*)(x + base + (size_t)t * C) - mean) * rstd;
            *(f32x4*)(dx + base + (size_t)t * C) = rstd * (gv - s1 - y * s2);
        }
    }
    if (n_pad > 0 && dpad && tg == 0)
        *(f32x4*)(dpad + (size_t)b * C + c0 + cq * 4) = rstd * (gp - s1 * np - ypad * s2 * np);
}

}  // namespace

int layernorm_bwd_blocks(int M) { int b = (M + 15) / 16; return b > 512 ? 512 : (b < 1 ? 1 : b); }
// ws: layernorm_bwd_blocks(M) * 2 * C floats; dgb: [2][C] = dgamma | dbeta
int layernorm_bwd_launch(const float* x, const float* gamma, const float* g, float* dx, float* dgb, float* ws, int M, int C, float eps, hipStream_t s, int accumulate,
                         const float* add) {
    if (C % 4 != 0 || C > 1024) { master_set_error("layernorm_bwd: C %% 4 == 0 and C <= 1024 (C=%d)", C); return MASTER_ERR_INVALID; }
    const int blocks = layernorm_bwd_blocks(M);
    const size_t lds = (size_t)8 * C * sizeof(float);
    const int ch = (C / 4 + 63) / 64;
    if (ch == 1) hipLaunchKernelGGL((layernorm_bwd_kernel<1>), dim3(blocks), dim3(256), lds, s, x, gamma, g, dx, ws, M, C, eps, add);
    else if (ch == 2) hipLaunchKernelGGL((layernorm_bwd_kernel<2>), dim3(blocks), dim3(256), lds, s, x, gamma, g, dx, ws, M, C, eps, add);
    else hipLaunchKernelGGL((layernorm_bwd_kernel<4>), dim3(blocks), dim3(256), lds, s, x, gamma, g, dx, ws, M, C, eps, add);
    return reduce_slabs_launch(ws, dgb, (size_t)2 * C, (size_t)2 * C, blocks, s, accumulate);
}
int instnorm_bwd_launch(const float* x, const float* g, const float* pad_in, const float* g_pad, float* dx, float* dpad, int B, int Tn, int C,
                        int n_pad, float eps, hipStream_t s) {
    if (C % 32 != 0) { master_set_error("instnorm_bwd: C %% 32 == 0 (C=%d)", C); return MASTER_ERR_INVALID; }
    if (Tn <= 32 * 32) hipLaunchKernelGGL(instnorm_bwd_kernel<32>, dim3(B * (C / 32)), dim3(256), 0, s, x, g, pad_in, g_pad, dx, dpad, Tn, C, n_pad, eps);
    else hipLaunchKernelGGL(instnorm_bwd_kernel<0>, dim3(B * (C / 32)), dim3(256), 0, s, x, g, pad_in, g_pad, dx, dpad, Tn, C, n_pad, eps);
    return master_check_launch("instnorm_bwd");
}
