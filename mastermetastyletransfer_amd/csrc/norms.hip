// HBM-bound row kernels: LayerNorm (optionally on a 2x2 patch-merge gather), InstanceNorm over the
// tokens of an image, Swin patch embedding (+LayerNorm), dtype conversion.  All statistics in fp32,
// two-pass (mean, then centred sum of squares) like the CPU path.
#include "common.h"

namespace {

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ f32x4 load(const float* p) { return *(const f32x4*)p; }
    static __device__ __forceinline__ void store(float* p, f32x4 v) { *(f32x4*)p = v; }
};
template <> struct Vec4<bf16_t> {
    static __device__ __forceinline__ f32x4 load(const bf16_t* p) {
        const uint2 u = *(const uint2*)p;
        f32x4 v; v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        return v;
    }
    static __device__ __forceinline__ void store(bf16_t* p, f32x4 v) {
        uint2 u;
        u.x = pack_bf16x2(v[0], v[1]);
        u.y = pack_bf16x2(v[2], v[3]);
        *(uint2*)p = u;
    }
};

// ---- LayerNorm: a row of C channels is C/VEC 16-byte chunks (VEC = 4 fp32 or 8 bf16); LPR = min(64, chunks)
//      lanes share a row, so one wave instruction covers 64/LPR rows, and each wave walks RG row groups
//      with all their loads issued before the first reduction.  MERGE gathers the row from the four 2x2
//      neighbours of an NHWC map (Swin PatchMerging order (0,0) (1,0) (0,1) (1,1)). -------------------
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float* v) { const f32x4 t = *(const f32x4*)p; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
    static __device__ __forceinline__ void store(float* p, const float* v) { f32x4 t; t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3]; *(f32x4*)p = t; }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const bf16_t* p, float* v) {
        const uint4 u = *(const uint4*)p;
        const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float* v) {
        uint4 u;
        u.x = pack_bf16x2(v[0], v[1]); u.y = pack_bf16x2(v[2], v[3]);
        u.z = pack_bf16x2(v[4], v[5]); u.w = pack_bf16x2(v[6], v[7]);
        *(uint4*)p = u;
    }
};

template <typename T, bool MERGE, int CH>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ out, int M, int C,
                                                        float eps, int H, int W, int Cin, int lpr) {
    constexpr int VEC = Vec16<T>::N;
    constexpr int RG = CH == 1 ? 4 : 1;               // row groups per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane & (lpr - 1), rsel = lane / lpr, rpw = 64 / lpr;
    const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
    float v[RG][CH][VEC];
    int rows[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        const int row = (wave_id * RG + g) * rpw + rsel;
        rows[g] = row;
        const int rr = row < M ? row : M - 1;
        size_t base[4];
        if (MERGE) {
            const int W2 = W >> 1, H2 = H >> 1;
            const int b = rr / (H2 * W2), rem = rr - b * (H2 * W2), y2 = rem / W2, x2 = rem - y2 * W2;
#pragma unroll
            for (int s = 0; s < 4; ++s) base[s] = (((size_t)b * H + 2 * y2 + (s & 1)) * W + 2 * x2 + (s >> 1)) * Cin;
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = (sub + lpr * i) * VEC;
            const T* src = MERGE ? x + base[c / Cin] + (c % Cin) : x + (size_t)rr * C + c;
            Vec16<T>::load(src, v[g][i]);
        }
    }
#pragma unroll
    for (int g = 0; g < RG; ++g) {
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int e = 0; e < VEC; ++e) sum += v[g][i][e];
        for (int o = lpr >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float mean = sum / (float)C;
        float sq = 0.0f;
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = v[g][i][e] - mean; sq += d * d; }
        for (int o = lpr >> 1; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
        const float rstd = rsqrtf(sq / (float)C + eps);
        if (rows[g] < M) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int c = (sub + lpr * i) * VEC;
                float o_[VEC];
#pragma unroll
                for (int e4 = 0; e4 < VEC / 4; ++e4) {
                    const f32x4 gm = *(const f32x4*)(gamma + c + 4 * e4), bt = *(const f32x4*)(beta + c + 4 * e4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o_[4 * e4 + e] = (v[g][i][4 * e4 + e] - mean) * rstd * gm[e] + bt[e];
                }
                Vec16<T>::store(out + (size_t)rows[g] * C + c, o_);
            }
        }
    }
}

// ---- InstanceNorm over tokens: block = (image b, 32-channel strip); a wave reads 8 (fp32) or 16 (bf16) rows x 128/64 B.
//      REG variant (T <= 256*ROWS/chunks-per-row): the whole strip lives in registers - one HBM read, one write, statistics
//      two-pass on the registers.  Generic variant: three sweeps of the strip (the 2nd and 3rd from L2). -----------------
template <typename T, int ROWS>
__global__ __launch_bounds__(256) void instnorm_reg_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                           const T* __restrict__ pad_in, T* __restrict__ pad_out, int Tn,
                                                           int C, int n_pad, float eps, int repeat) {
    constexpr int VEC = Vec16<T>::N;                 // elements per 16-B chunk
    constexpr int CPR = 32 / VEC;                    // chunks per 32-channel row segment (8 fp32, 4 bf16)
    constexpr int TG = 256 / CPR;                    // token groups
    __shared__ float red[4][33];
    const int strips = C >> 5;
    const int b = blockIdx.x / strips, c0 = (blockIdx.x % strips) * 32;
    const int cq = threadIdx.x % CPR, tg = threadIdx.x / CPR;
    const T* xb = x + (size_t)b * Tn * C + c0 + cq * VEC;
    T* ob = out + (size_t)b * Tn * C + c0 + cq * VEC;
    float v[ROWS][VEC];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const int t = tg + i * TG;
        if (t < Tn) Vec16<T>::load(xb + (size_t)t * C, v[i]);
        else
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[i][e] = 0.f;
    }
    float pr[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) pr[e] = 0.f;
    if (n_pad > 0) Vec16<T>::load(pad_in + c0 + cq * VEC, pr);
    const float cnt = (float)(Tn + n_pad);

    auto block_sum = [&](float* a) {                 // a[VEC] summed over the TG token groups, result broadcast
        // lanes of a wave that share cq differ by multiples of CPR: butterfly over those, then 4 partials through LDS
#pragma unroll
        for (int e = 0; e < VEC; ++e)
#pragma unroll
            for (int o = CPR; o < 64; o <<= 1) a[e] += __shfl_xor(a[e], o, 64);
        const int wv = threadIdx.x >> 6;
        if ((threadIdx.x & 63) < CPR)
#pragma unroll
            for (int e = 0; e < VEC; ++e) red[wv][cq * VEC + e] = a[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] = red[0][cq * VEC + e] + red[1][cq * VEC + e] + red[2][cq * VEC + e] + red[3][cq * VEC + e];
        __syncthreads();
    };
    float mean[VEC], var[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { mean[e] = 0.f; for (int i = 0; i < ROWS; ++i) mean[e] += v[i][e]; }
    block_sum(mean);
#pragma unroll
    for (int e = 0; e < VEC; ++e) mean[e] = (mean[e] + (float)n_pad * pr[e]) / cnt;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        var[e] = 0.f;
        for (int i = 0; i < ROWS; ++i) { const float d = (tg + i * TG < Tn) ? v[i][e] - mean[e] : 0.f; var[e] += d * d; }
    }
    block_sum(var);
    float rstd[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const float dp = pr[e] - mean[e];
        const float vr = (var[e] + (float)n_pad * dp * dp) / cnt;
        rstd[e] = rsqrtf(vr + eps);
        // InstanceNorm applied twice (reference style_transformer.py:1056 then :468): the first output has mean 0 and
        // variance vr/(vr+eps), so the second pass only rescales by rsqrt(vr/(vr+eps) + eps)
        if (repeat == 2) rstd[e] *= rsqrtf(vr * rstd[e] * rstd[e] + eps);
    }
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const int t = tg + i * TG;
        if (t < Tn) {
            float o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = (v[i][e] - mean[e]) * rstd[e];
            Vec16<T>::store(ob + (size_t)t * C, o);
        }
    }
    if (n_pad > 0 && pad_out && tg == 0) {
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = (pr[e] - mean[e]) * rstd[e];
        Vec16<T>::store(pad_out + (size_t)b * C + c0 + cq * VEC, o);
    }
}

// STRIP = channels per workgroup (32, or 16 when 32-channel strips would leave CUs idle: the 64x64 maps of 512x512 images at batch 16
// make 128 strips of 32 on 256 CUs - r03: 151 us per launch of the fp32 mode's BASELINE config 5)
template <typename T, int STRIP>
__global__ __launch_bounds__(256) void instnorm_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                       const T* __restrict__ pad_in, T* __restrict__ pad_out, int Tn,
                                                       int C, int n_pad, float eps, int repeat) {
    constexpr int CQ = STRIP / 4, TG = 256 / CQ;                         // 4-channel vectors per strip, token groups
    __shared__ f32x4 red[TG][CQ];
    const int strips = C / STRIP;
    const int b = blockIdx.x / strips, c0 = (blockIdx.x % strips) * STRIP;
    const int cq = threadIdx.x % CQ, tg = threadIdx.x / CQ;
    const T* xb = x + (size_t)b * Tn * C + c0 + cq * 4;
    T* ob = out + (size_t)b * Tn * C + c0 + cq * 4;
    f32x4 pr = {0.f, 0.f, 0.f, 0.f};
    if (n_pad > 0) pr = Vec4<T>::load(pad_in + c0 + cq * 4);
    const float cnt = (float)(Tn + n_pad);

    auto block_sum = [&](f32x4 v) -> f32x4 {
        red[tg][cq] = v;
        __syncthreads();
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < TG; ++i) s += red[i][cq];
        __syncthreads();
        return s;
    };

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int t = tg; t < Tn; t += TG) acc += Vec4<T>::load(xb + (size_t)t * C);
    f32x4 mean = block_sum(acc);
#pragma unroll
    for (int e = 0; e < 4; ++e) mean[e] = (mean[e] + (float)n_pad * pr[e]) / cnt;

    acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = tg; t < Tn; t += TG) {
        const f32x4 d = Vec4<T>::load(xb + (size_t)t * C) - mean;
        acc += d * d;
    }
    f32x4 var = block_sum(acc), rstd;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float dp = pr[e] - mean[e];
        const float vr = (var[e] + (float)n_pad * dp * dp) / cnt;
        rstd[e] = rsqrtf(vr + eps);
        if (repeat == 2) rstd[e] *= rsqrtf(vr * rstd[e] * rstd[e] + eps);
    }
    for (int t = tg; t < Tn; t += TG)
        Vec4<T>::store(ob + (size_t)t * C, (Vec4<T>::load(xb + (size_t)t * C) - mean) * rstd);
    if (n_pad > 0 && pad_out && tg == 0)
        Vec4<T>::store(pad_out + (size_t)b * C + c0 + cq * 4, (pr - mean) * rstd);
}

__device__ __forceinline__ void store2(float* p, float a, float b) { *(f32x2*)p = f32x2{a, b}; }
__device__ __forceinline__ void store2(bf16_t* p, float a, float b) { *(unsigned*)p = pack_bf16x2(a, b); }

// ---- Swin patch embedding: Conv2d(3,E,4,4) + LayerNorm(E), E = 128 (swin_B) or 96 (swin_T / swin_S).  A unit of work = 64 patches
//      of one patch row; a workgroup stages the weights once and loops over units.  Thread (patch group pg, channel group cq) produces
//      channels [cq*E/16, (cq+1)*E/16) of patches 4pg .. 4pg+3: per k it reads E/16 weights and 4 pixels from LDS for 4*E/16 FMAs.
//      (r04: the first form gave a thread 1 patch x 32 channels - 132 B of LDS reads per k for 32 FMAs - and re-staged the 24 KB of weights
//      for every 64 patches through a 48-way bank-conflicted transpose: 74 us per launch at 32 images, against 15 us of HBM time.) -------
template <typename T, int E>
__global__ __launch_bounds__(256) void patch_embed_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                          const float* __restrict__ bias, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ out, int B, int H, int W,
                                                          float eps) {
    constexpr int KP = 48, PSTR = 49, CPT = E / 16;
    static_assert(E % 32 == 0 && CPT % 2 == 0, "channel groups of an even number of channels");
    __shared__ __attribute__((aligned(16))) float pix[64 * PSTR];
    __shared__ __attribute__((aligned(16))) float wt[KP * E];   // [k][e]
    const int PW = W >> 2, PH = H >> 2;
    const int chunks = (PW + 63) / 64;
    const int units = B * PH * chunks;
    const int tid = threadIdx.x;

    // w[e][k] -> wt[k][e]: lane <-> e, so the transposed stores walk the banks (the loads are 16 B at a 192-B stride, from L2)
    for (int i = tid; i < 12 * E; i += 256) {
        const int e = i % E, k4 = i / E;
        const f32x4 v = *(const f32x4*)(w + e * KP + k4 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) wt[(k4 * 4 + j) * E + e] = v[j];
    }
    const int pg = tid >> 4, cq = tid & 15;
    float bv[CPT], gv[CPT], btv[CPT];
#pragma unroll
    for (int e = 0; e < CPT; ++e) { bv[e] = bias[cq * CPT + e]; gv[e] = gamma[cq * CPT + e]; btv[e] = beta[cq * CPT + e]; }

    for (int unit = blockIdx.x; unit < units; unit += gridDim.x) {
        int u = unit;
        const int pc = u % chunks; u /= chunks;
        const int py = u % PH; const int b = u / PH;
        const int px0 = pc * 64;
        __syncthreads();                                 // the previous unit's pixels are consumed (first pass: the weights are staged)
        for (int seg = 0; seg < 12; ++seg) {            // (c, dy) image row segments, 256 contiguous pixels
            const int c = seg >> 2, dy = seg & 3;
            const int xx = px0 * 4 + tid;
            pix[(tid >> 2) * PSTR + c * 16 + dy * 4 + (tid & 3)] = xx < W ? img[(((size_t)b * 3 + c) * H + py * 4 + dy) * W + xx] : 0.f;
        }
        __syncthreads();

        float acc[4][CPT];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < CPT; ++e) acc[j][e] = bv[e];
#pragma unroll 4
        for (int k = 0; k < KP; ++k) {
            float wv[CPT];
            const f32x2* wr = (const f32x2*)(wt + k * E + cq * CPT);
#pragma unroll
            for (int e2 = 0; e2 < CPT / 2; ++e2) { const f32x2 t = wr[e2]; wv[2 * e2] = t[0]; wv[2 * e2 + 1] = t[1]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = pix[(pg * 4 + j) * PSTR + k];
#pragma unroll
                for (int e = 0; e < CPT; ++e) acc[j][e] = fmaf(a, wv[e], acc[j][e]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < CPT; ++e) s += acc[j][e];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
            const float mean = s / (float)E;
            float sq = 0.f;
#pragma unroll
            for (int e = 0; e < CPT; ++e) { const float d = acc[j][e] - mean; sq += d * d; }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sq += __shfl_xor(sq, o, 64);
            const float rstd = rsqrtf(sq / (float)E + eps);
            const int p = pg * 4 + j;
            if (px0 + p < PW) {
                T* op = out + ((((size_t)b * PH + py) * PW + px0 + p) * E) + cq * CPT;
                float o_[CPT];
#pragma unroll
                for (int e = 0; e < CPT; ++e) o_[e] = (acc[j][e] - mean) * rstd * gv[e] + btv[e];
                if constexpr (CPT % 4 == 0) {
#pragma unroll
                    for (int e4 = 0; e4 < CPT / 4; ++e4) Vec4<T>::store(op + e4 * 4, f32x4{o_[4 * e4], o_[4 * e4 + 1], o_[4 * e4 + 2], o_[4 * e4 + 3]});
                } else {
#pragma unroll
                    for (int e2 = 0; e2 < CPT / 2; ++e2) store2(op + 2 * e2, o_[2 * e2], o_[2 * e2 + 1]);
                }
            }
        }
    }
}

template <typename S, typename D>
__global__ __launch_bounds__(256) void convert_kernel(const S* __restrict__ src, D* __restrict__ dst, size_t n) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (; i + 3 < n; i += stride) Vec4<D>::store(dst + i, Vec4<S>::load(src + i));
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t t = n & ~(size_t)3; t < n; ++t) {
            float v = sizeof(S) == 4 ? ((const float*)src)[t] : bf16_to_f32(((const bf16_t*)src)[t]);
            if (sizeof(D) == 4) ((float*)dst)[t] = v; else ((bf16_t*)dst)[t] = f32_to_bf16(v);
        }
}

// ---- LayerNorm for widths whose vector count is not a power of two (swin_T / swin_S: 96, 192, 384 channels): one wave per row,
//      lane l holds the 16-byte vectors l, l+64, l+128, l+192 of the row (C <= 256 vectors), sums over the whole wave. ----------------
template <typename T, bool MERGE>
__global__ __launch_bounds__(256) void layernorm_any_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ out, int M, int C,
                                                            float eps, int H, int W, int Cin) {
    constexpr int VEC = Vec16<T>::N;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;                                        // whole waves leave together: no barrier in this kernel
    const int chunks = C / VEC;
    size_t base[4] = {0, 0, 0, 0};
    if (MERGE) {
        const int W2 = W >> 1, H2 = H >> 1;
        const int b = row / (H2 * W2), rem = row - b * (H2 * W2), y2 = rem / W2, x2 = rem - y2 * W2;
#pragma unroll
        for (int s = 0; s < 4; ++s) base[s] = (((size_t)b * H + 2 * y2 + (s & 1)) * W + 2 * x2 + (s >> 1)) * Cin;
    }
    float v[4][VEC];
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (lane + 64 * i) * VEC;
        if (lane + 64 * i < chunks) {
            const int seg = MERGE ? c / Cin : 0;
            const T* src = MERGE ? x + (seg == 0 ? base[0] : seg == 1 ? base[1] : seg == 2 ? base[2] : base[3]) + (c - seg * Cin) : x + (size_t)row * C + c;
            Vec16<T>::load(src, v[i]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) sum += v[i][e];
        }
    }
    sum = wave_sum(sum);
    const float mean = sum / (float)C;
    float sq = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (lane + 64 * i < chunks)
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    sq = wave_sum(sq);
    const float rstd = rsqrtf(sq / (float)C + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = (lane + 64 * i) * VEC;
        if (lane + 64 * i < chunks) {
            float o_[VEC];
#pragma unroll
            for (int e4 = 0; e4 < VEC / 4; ++e4) {
                const f32x4 gm = *(const f32x4*)(gamma + c + 4 * e4), bt = *(const f32x4*)(beta + c + 4 * e4);
#pragma unroll
                for (int e = 0; e < 4; ++e) o_[4 * e4 + e] = (v[i][4 * e4 + e] - mean) * rstd * gm[e] + bt[e];
            }
            Vec16<T>::store(out + (size_t)row * C + c, o_);
        }
    }
}

}  // namespace

template <typename T, bool MERGE>
static void layernorm_dispatch(const void* x, const void* g, const void* b, void* out, int M, int C, float eps, int H,
                               int W, int Cin, hipStream_t s) {
    constexpr int VEC = Vec16<T>::N;
    const int chunks = C / VEC;
    const int lpr = chunks >= 64 ? 64 : chunks;            // power of two (checked by the caller)
    const int ch = chunks / lpr;
    const int rpw = 64 / lpr;
#define LN_LAUNCH(CHV)                                                                                               \
    {                                                                                                                \
        const int rg = CHV == 1 ? 4 : 1;                                                                             \
        const int waves = (M + rpw * rg - 1) / (rpw * rg);                                                           \
        hipLaunchKernelGGL((layernorm_kernel<T, MERGE, CHV>), dim3((waves + 3) / 4), dim3(256), 0, s, (const T*)x,   \
                           (const float*)g, (const float*)b, (T*)out, M, C, eps, H, W, Cin, lpr);                    \
    }
    if (ch == 1) LN_LAUNCH(1) else if (ch == 2) LN_LAUNCH(2) else LN_LAUNCH(4)
#undef LN_LAUNCH
}

template <typename T>
int layernorm_launch(const void* x, const void* g, const void* b, void* out, int M, int C, float eps, int merge, int H,
                     int W, int Cin, hipStream_t s) {
    constexpr int VEC = Vec16<T>::N;
    const int chunks = C / VEC;
    if (C % VEC != 0 || chunks > 256 || (merge && Cin % VEC != 0)) {
        master_set_error("layernorm: C must be a multiple of %d and at most %d (C=%d)", VEC, 256 * VEC, C);
        return MASTER_ERR_INVALID;
    }
    if ((chunks & (chunks - 1)) != 0) {          // not a power of two: the one-wave-per-row kernel
        if (merge) hipLaunchKernelGGL((layernorm_any_kernel<T, true>), dim3((M + 3) / 4), dim3(256), 0, s, (const T*)x, (const float*)g,
                                      (const float*)b, (T*)out, M, C, eps, H, W, Cin);
        else hipLaunchKernelGGL((layernorm_any_kernel<T, false>), dim3((M + 3) / 4), dim3(256), 0, s, (const T*)x, (const float*)g,
                                (const float*)b, (T*)out, M, C, eps, 0, 0, C);
        return master_check_launch("layernorm");
    }
    if (merge) layernorm_dispatch<T, true>(x, g, b, out, M, C, eps, H, W, Cin, s);
    else layernorm_dispatch<T, false>(x, g, b, out, M, C, eps, 0, 0, C, s);
    return master_check_launch("layernorm");
}
template int layernorm_launch<float>(const void*, const void*, const void*, void*, int, int, float, int, int, int, int, hipStream_t);
template int layernorm_launch<bf16_t>(const void*, const void*, const void*, void*, int, int, float, int, int, int, int, hipStream_t);

template <typename T>
int instnorm_launch(const void* x, void* out, const void* pad_in, void* pad_out, int B, int Tn, int C, int n_pad,
                    float eps, int repeat, hipStream_t s) {
    constexpr int TG = 256 / (32 / Vec16<T>::N);
    if (Tn <= TG * 16)         // whole strip in registers: one read, one write
        hipLaunchKernelGGL((instnorm_reg_kernel<T, 16>), dim3(B * (C / 32)), dim3(256), 0, s, (const T*)x, (T*)out,
                           (const T*)pad_in, (T*)pad_out, Tn, C, n_pad, eps, repeat);
    else if (Tn <= TG * 32 && sizeof(T) == 4)      // fp32: 32 rows x 4 values per thread
        hipLaunchKernelGGL((instnorm_reg_kernel<T, 32>), dim3(B * (C / 32)), dim3(256), 0, s, (const T*)x, (T*)out,
                           (const T*)pad_in, (T*)pad_out, Tn, C, n_pad, eps, repeat);
    else if (B * (C / 32) < 256 && C % 16 == 0)          // fewer 32-channel strips than CUs: 16-channel strips double the workgroups
        hipLaunchKernelGGL((instnorm_kernel<T, 16>), dim3(B * (C / 16)), dim3(256), 0, s, (const T*)x, (T*)out,
                           (const T*)pad_in, (T*)pad_out, Tn, C, n_pad, eps, repeat);
    else
        hipLaunchKernelGGL((instnorm_kernel<T, 32>), dim3(B * (C / 32)), dim3(256), 0, s, (const T*)x, (T*)out,
                           (const T*)pad_in, (T*)pad_out, Tn, C, n_pad, eps, repeat);
    return master_check_launch("instnorm");
}
template int instnorm_launch<float>(const void*, void*, const void*, void*, int, int, int, int, float, int, hipStream_t);
template int instnorm_launch<bf16_t>(const void*, void*, const void*, void*, int, int, int, int, float, int, hipStream_t);

template <typename T>
int patch_embed_launch(const float* img, const void* w, const void* bias, const void* g, const void* b, void* out,
                       int B, int H, int W, int E, float eps, hipStream_t s) {
    const int PW = W / 4, PH = H / 4, chunks = (PW + 63) / 64;
    const int units = B * PH * chunks, grid = units < 1024 ? units : 1024;       // four resident workgroups per CU (37 KB of LDS each), looping
    if (E == 128)
        hipLaunchKernelGGL((patch_embed_kernel<T, 128>), dim3(grid), dim3(256), 0, s, img, (const float*)w,
                           (const float*)bias, (const float*)g, (const float*)b, (T*)out, B, H, W, eps);
    else if (E == 96)
        hipLaunchKernelGGL((patch_embed_kernel<T, 96>), dim3(grid), dim3(256), 0, s, img, (const float*)w,
                           (const float*)bias, (const float*)g, (const float*)b, (T*)out, B, H, W, eps);
    else { master_set_error("patch_embed: embed dim %d not built (128 = swin_B, 96 = swin_T / swin_S)", E); return MASTER_ERR_UNSUPPORTED; }
    return master_check_launch("patch_embed");
}
template int patch_embed_launch<float>(const float*, const void*, const void*, const void*, const void*, void*, int, int, int, int, float, hipStream_t);
template int patch_embed_launch<bf16_t>(const float*, const void*, const void*, const void*, const void*, void*, int, int, int, int, float, hipStream_t);

int convert_launch(const void* src, void* dst, size_t n, int sdt, int ddt, hipStream_t s) {
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    if (sdt == MASTER_F32 && ddt == MASTER_BF16)
        hipLaunchKernelGGL((convert_kernel<float, bf16_t>), dim3(blocks), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, n);
    else if (sdt == MASTER_BF16 && ddt == MASTER_F32)
        hipLaunchKernelGGL((convert_kernel<bf16_t, float>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, n);
    else if (sdt == MASTER_F32 && ddt == MASTER_F32)
        hipLaunchKernelGGL((convert_kernel<float, float>), dim3(blocks), dim3(256), 0, s, (const float*)src, (float*)dst, n);
    else
        hipLaunchKernelGGL((convert_kernel<bf16_t, bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, n);
    return master_check_launch("convert");
}
