// Pre/post-processing around the forward on the device (reference test_model.py:39-48,110-125 and :207).
//
// Pre:  uint8 HWC RGB -> Pillow-compatible antialiased bilinear resize -> ToTensor (/255) -> optional Normalize -> fp32 NCHW.
//       transforms.Resize on a PIL image is Pillow's two-pass resample in 8-bit fixed point (src/libImaging/Resample.c):
//       per output index a window of source taps (bounds = first index, count) with triangle-filter weights normalised in
//       double and rounded to 22-bit integers; each pass is  clip8((2^21 + sum tap * k) >> 22), horizontal first, and the
//       intermediate image is uint8.  The tables are computed on the HOST in double exactly as Pillow does
//       (master_resize_plan) so the device only does integer arithmetic: results are bit-identical to Pillow.
// Post: clip(x * 255, 0, 255) truncated to uint8, NCHW -> HWC.
#include "common.h"
#include <cmath>

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ unsigned clip8(int v) { v >>= PRECISION_BITS; return (unsigned)(v < 0 ? 0 : v > 255 ? 255 : v); }

// horizontal pass: in [B][H][W][3] u8 -> tmp [B][H][OW][3] u8; thread = (ox, y, b)
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ tmp, int H, int W, int OW,
                                                       const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ks) {
    const int ox = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (ox >= OW) return;
    const int x0 = bounds[2 * ox], n = bounds[2 * ox + 1];
    const uint8_t* row = in + ((size_t)b * H + y) * W * 3 + (size_t)x0 * 3;
    const int32_t* k = kk + (size_t)ox * ks;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) { const int c = k[x]; s0 += row[3 * x] * c; s1 += row[3 * x + 1] * c; s2 += row[3 * x + 2] * c; }
    uint8_t* o = tmp + (((size_t)b * H + y) * OW + ox) * 3;
    o[0] = (uint8_t)clip8(s0); o[1] = (uint8_t)clip8(s1); o[2] = (uint8_t)clip8(s2);
}

// vertical pass + ToTensor + Normalize: src [B][H][OW][3] u8 -> out [B][3][OH][OW] fp32; thread = (ox, oy, b)
__global__ __launch_bounds__(256) void resize_v_tensor_kernel(const uint8_t* __restrict__ src, float* __restrict__ out, int H, int OH, int OW,
                                                              const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ks,
                                                              int normalize, float m0, float m1, float m2, float d0, float d1, float d2) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y, b = blockIdx.z;
    if (ox >= OW) return;
    const int y0 = bounds[2 * oy], n = bounds[2 * oy + 1];
    const int32_t* k = kk + (size_t)oy * ks;
    const uint8_t* p = src + (((size_t)b * H + y0) * OW + ox) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < n; ++y, p += (size_t)OW * 3) { const int c = k[y]; s0 += p[0] * c; s1 += p[1] * c; s2 += p[2] * c; }
    float v0 = (float)clip8(s0) / 255.0f, v1 = (float)clip8(s1) / 255.0f, v2 = (float)clip8(s2) / 255.0f;     // ToTensor
    if (normalize) { v0 = (v0 - m0) / d0; v1 = (v1 - m1) / d1; v2 = (v2 - m2) / d2; }                             // Normalize: sub, then div
    const size_t plane = (size_t)OH * OW, o = ((size_t)b * 3) * plane + (size_t)oy * OW + ox;
    out[o] = v0; out[o + plane] = v1; out[o + 2 * plane] = v2;
}

// np.clip(x * 255, 0, 255).astype(uint8): [B][3][H][W] fp32 -> [B][H][W][3] u8; thread = pixel
__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ img, uint8_t* __restrict__ out, int HW, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t b = i / HW, px = i - b * HW;
    const float* p = img + b * 3 * (size_t)HW + px;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = p[(size_t)c * HW] * 255.0f;
        v = v < 0.0f ? 0.0f : v > 255.0f ? 255.0f : v;           // NaN falls through both comparisons like np.clip; the cast is then undefined there too
        out[i * 3 + c] = (uint8_t)(int)v;                         // truncation toward zero
    }
}

}  // namespace

// Pillow precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter over the whole axis, on the host, in double.
int resize_plan_host(int in_size, int out_size, int32_t* bounds, int32_t* coeffs, int* ksize_out) {
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    if (ksize_out) *ksize_out = ksize;
    if (!bounds || !coeffs) return 0;
    const double ss = 1.0 / filterscale;
    double* w = new double[ksize];
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5); if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5); if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < ksize; ++x) w[x] = 0.0;
        for (int x = 0; x < xmax; ++x) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            w[x] = a < 1.0 ? 1.0 - a : 0.0;
            ww += w[x];
        }
        for (int x = 0; x < xmax; ++x) if (ww != 0.0) w[x] /= ww;
        for (int x = 0; x < ksize; ++x)
            coeffs[(size_t)xx * ksize + x] = w[x] < 0 ? (int)(-0.5 + w[x] * (1 << PRECISION_BITS)) : (int)(0.5 + w[x] * (1 << PRECISION_BITS));
        bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
    }
    delete[] w;
    return 0;
}

int preprocess_launch(const uint8_t* img, int B, int H, int W, float* out, int OH, int OW, const int32_t* hb, const int32_t* hk, int hks,
                      const int32_t* vb, const int32_t* vk, int vks, uint8_t* tmp, const float* mean, const float* stdv, hipStream_t s) {
    const uint8_t* src = img;
    if (OW != W) {                                               // Pillow skips a pass whose size does not change
        hipLaunchKernelGGL(resize_h_kernel, dim3((OW + 255) / 256, H, B), dim3(256), 0, s, img, tmp, H, W, OW, hb, hk, hks);
        src = tmp;
    }
    const int norm = mean && stdv;
    hipLaunchKernelGGL(resize_v_tensor_kernel, dim3((OW + 255) / 256, OH, B), dim3(256), 0, s, src, out, H, OH, OW, vb, vk, vks, norm,
                       norm ? mean[0] : 0.f, norm ? mean[1] : 0.f, norm ? mean[2] : 0.f, norm ? stdv[0] : 1.f, norm ? stdv[1] : 1.f, norm ? stdv[2] : 1.f);
    return master_check_launch("preprocess");
}

int postprocess_launch(const float* img, uint8_t* out, int B, int H, int W, hipStream_t s) {
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(postprocess_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, img, out, H * W, total);
    return master_check_launch("postprocess");
}
