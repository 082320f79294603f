// Pieces of the perceptual-loss forward (reference codes/loss.py, evaluated under no_grad by test_model.py:190-203) that are not
// convolutions: image -> padded NHWC, MaxPool 2x2/2, per-(image, channel) statistics, and the InstanceNorm'd L1/L2 distance.
// The 13 VGG-19 convolutions (zero padding) run on the implicit-GEMM family (gemm_{f32,bf16}.hip, zero_pad = 1).
#include "common.h"

namespace {

template <typename T> struct Elem;
template <> struct Elem<float> { static __device__ float ld(const float* p) { return *p; } static __device__ void st(float* p, float v) { *p = v; } };
template <> struct Elem<bf16_t> {
    static __device__ float ld(const bf16_t* p) { return __uint_as_float((unsigned)*p << 16); }
    static __device__ void st(bf16_t* p, float v) { *p = (bf16_t)(pack_bf16x2(v, 0.f) & 0xffffu); }
};

// fp32 NCHW [N,3,H,W] -> T NHWC [N,H,W,CP], channels 3..CP-1 zero (the first VGG conv then sees Cin = CP with zero weights there)
template <typename T>
__global__ __launch_bounds__(256) void image_to_nhwc_kernel(const float* __restrict__ img, T* __restrict__ out, int HW, int CP, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;            // one thread per (pixel, channel)
    if (i >= total) return;
    const int c = (int)(i % CP); const size_t px = i / CP, n = px / HW, q = px - n * HW;
    Elem<T>::st(out + i, c < 3 ? img[(n * 3 + c) * (size_t)HW + q] : 0.f);
}

// MaxPool2d(2, 2) on NHWC: thread = (output pixel, channel)
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_kernel(const T* __restrict__ x, T* __restrict__ out, int H, int W, int C, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C); size_t t = i / C;
    const int W2 = W >> 1, H2 = H >> 1;
    const int ox = (int)(t % W2); t /= W2; const int oy = (int)(t % H2); const size_t n = t / H2;
    const T* p = x + ((n * H + 2 * oy) * (size_t)W + 2 * ox) * C + c;
    const float a = Elem<T>::ld(p), b = Elem<T>::ld(p + C), d = Elem<T>::ld(p + (size_t)W * C), e = Elem<T>::ld(p + (size_t)W * C + C);
    Elem<T>::st(out + i, fmaxf(fmaxf(a, b), fmaxf(d, e)));
}

// ---- per-(image, channel) statistics and the InstanceNorm'd distance --------------------------------------------------------------
// Layout of the work: a thread owns V consecutive channels (16 bytes), TPR = C/V threads cover a row, a 256-thread workgroup walks
// 256/TPR rows at a time over its SLICE of the image's rows; grid = (row slices, images).  Thousands of workgroups stream whole
// contiguous rows.  Every kernel writes per-slice partial sums [image][slice][C]; the next kernel (or stats_finalize) adds the
// slices in a fixed order, so results are deterministic.  Variance is two-pass (mean first, then squared deviations) like torch.
template <typename T> struct VecLd;
template <> struct VecLd<float> { static constexpr int V = 4; static __device__ void ld(const float* p, float (&v)[4]) { const f32x4 u = *(const f32x4*)p; v[0] = u[0]; v[1] = u[1]; v[2] = u[2]; v[3] = u[3]; } };
template <> struct VecLd<bf16_t> {
    static constexpr int V = 8;
    static __device__ void ld(const bf16_t* p, float (&v)[8]) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 u = *(const u32x4*)p;
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(u[e] << 16); v[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
    }
};
constexpr int SLICE_ROWS = 256;            // rows per workgroup

// MODE 0: partial[n][slice][c] = sum x.   MODE 1: partial = sum (x - mean)^2 with mean = (sum over slices of psum) / HW.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void chan_partial_kernel(const T* __restrict__ x, const float* __restrict__ psum, float* __restrict__ partial,
                                                           int HW, int C, int nslice) {
    constexpr int V = VecLd<T>::V;
    extern __shared__ float red[];                              // [256 / TPR][C]
    const int TPR = C / V, RPB = 256 / TPR;                     // threads per row, rows per pass
    const int tc = threadIdx.x % TPR, tr = threadIdx.x / TPR, n = blockIdx.y, sl = blockIdx.x;
    const int r0 = sl * SLICE_ROWS, r1 = min(r0 + SLICE_ROWS, HW);
    float m[V], acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { m[e] = 0.f; acc[e] = 0.f; }
    if (MODE == 1) {
        for (int s2 = 0; s2 < nslice; ++s2)
#pragma unroll
            for (int e = 0; e < V; ++e) m[e] += psum[((size_t)n * nslice + s2) * C + tc * V + e];
#pragma unroll
        for (int e = 0; e < V; ++e) m[e] /= (float)HW;
    }
    if (tr < RPB)
        for (int r = r0 + tr; r < r1; r += RPB) {
            float v[V];
            VecLd<T>::ld(x + ((size_t)n * HW + r) * C + tc * V, v);
#pragma unroll
            for (int e = 0; e < V; ++e) { const float d = v[e] - m[e]; acc[e] += MODE == 1 ? d * d : d; }
        }
#pragma unroll
    for (int e = 0; e < V; ++e) red[tr * C + tc * V + e] = acc[e];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float t = 0.f;
        for (int j = 0; j < RPB; ++j) t += red[j * C + c];
        partial[((size_t)n * nslice + sl) * C + c] = t;
    }
}

__global__ __launch_bounds__(256) void stats_finalize_kernel(const float* __restrict__ psum, const float* __restrict__ psq, float* __restrict__ mean,
                                                             float* __restrict__ var, int HW, int C, int nslice, int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;               // (n, c)
    if (i >= total) return;
    const int n = i / C, c = i - n * C;
    float s = 0.f, q = 0.f;
    for (int s2 = 0; s2 < nslice; ++s2) { s += psum[((size_t)n * nslice + s2) * C + c]; q += psq[((size_t)n * nslice + s2) * C + c]; }
    mean[i] = s / (float)HW; var[i] = q / (float)HW;
}

// sum over a row slice and all channels of f(IN(a) - IN(b)) for the pair (image i, image i + pair_off); one partial per workgroup
template <typename T>
__global__ __launch_bounds__(256) void in_dist_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ var,
                                                      float* __restrict__ partial, int pair_off, int HW, int C, float eps, int squared) {
    constexpr int V = VecLd<T>::V;
    __shared__ float red[4];
    const int TPR = C / V, RPB = 256 / TPR;
    const int tc = threadIdx.x % TPR, tr = threadIdx.x / TPR, i = blockIdx.y, j = i + pair_off, sl = blockIdx.x;
    const int r0 = sl * SLICE_ROWS, r1 = min(r0 + SLICE_ROWS, HW);
    float ma[V], ra[V], mb[V], rb[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const int c = tc * V + e;
        ma[e] = mean[(size_t)i * C + c]; ra[e] = rsqrtf(var[(size_t)i * C + c] + eps);
        mb[e] = mean[(size_t)j * C + c]; rb[e] = rsqrtf(var[(size_t)j * C + c] + eps);
    }
    float s = 0.f;
    if (tr < RPB)
        for (int r = r0 + tr; r < r1; r += RPB) {
            float va[V], vb[V];
            VecLd<T>::ld(x + ((size_t)i * HW + r) * C + tc * V, va);
            VecLd<T>::ld(x + ((size_t)j * HW + r) * C + tc * V, vb);
#pragma unroll
            for (int e = 0; e < V; ++e) { const float d = (va[e] - ma[e]) * ra[e] - (vb[e] - mb[e]) * rb[e]; s += squared ? d * d : fabsf(d); }
        }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ---- VGG-19 conv1_1 + ReLU (loss.py:23-37: Conv2d(3, 64, 3, padding=1) on the fp32 NCHW image), direct (r03).  As an implicit GEMM the
//      layer ran with its 3 input channels zero-padded to 32 (K = 288 for 27 real taps) behind a pass that wrote that padded NHWC copy of
//      the images (201 MB at 24 x 256^2): ~0.7 ms of the train step for 5 GFLOP.  Here: workgroup = 4 output rows x 64 pixels, wave = one
//      row, LANE = OUTPUT CHANNEL (Cout = 64 = the wave): the lane's 27 weights live in registers, the 3 x 6 x 66 input halo tile (zero
//      padding resolved at load) in LDS; every input value is a wave-uniform LDS read (a 3-wide sliding window per (channel, ky): 9
//      reads per pixel), 27 FMAs per pixel and lane, and the store of a pixel is ONE contiguous 256-byte (fp32) / 128-byte (bf16) row.
template <typename T>
__global__ __launch_bounds__(256) void vgg_conv1_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
                                                        T* __restrict__ out, int H, int W, int relu) {
    __shared__ float halo[3][6][68];                             // rows y0-1 .. y0+4, columns x0-1 .. x0+64
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.z, y0 = blockIdx.y * 4, x0 = blockIdx.x * 64;
    for (int i = tid; i < 3 * 6 * 66; i += 256) {
        const int ci = i / (6 * 66), rem = i - ci * (6 * 66), ry = rem / 66, rx = rem - ry * 66;
        const int gy = y0 - 1 + ry, gx = x0 - 1 + rx;
        halo[ci][ry][rx] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? img[(((size_t)n * 3 + ci) * H + gy) * W + gx] : 0.f;
    }
    // fp64 accumulation (vector fp64 FMAs issue at the rate of un-packed fp32 ones on this chip): the layer's output is the correctly
    // rounded fp32 value of the 27-term sum.  The loss behind it normalises nearly constant channels (InstanceNorm, eps 1e-5) and
    // amplifies first-layer rounding by orders of magnitude on small maps (profiles/r02_grad_conditioning.txt).
    double wg[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) wg[k] = (double)w[lane * 27 + k]; // nn.Conv2d layout [co][ci][ky][kx]
    const double b = (double)bias[lane];
    __syncthreads();
    const int y = y0 + wave;
    if (y >= H) return;
    int npx = W - x0; npx = npx > 64 ? 64 : npx;
    T* orow = out + (((size_t)n * H + y) * W + x0) * 64 + lane;
    double win[9][3];                                            // [(ci, ky)][kx]: the three columns under the current pixel
#pragma unroll
    for (int r = 0; r < 9; ++r) { win[r][0] = (double)halo[r / 3][wave + r % 3][0]; win[r][1] = (double)halo[r / 3][wave + r % 3][1]; }
    auto pixel = [&](int px, int i0, int i1, int i2) __attribute__((always_inline)) {      // window columns (i0, i1, i2) = (px-1, px, px+1)
#pragma unroll
        for (int r = 0; r < 9; ++r) win[r][i2] = (double)halo[r / 3][wave + r % 3][px + 2];
        double acc = b;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            acc = fma(wg[3 * r + 0], win[r][i0], acc);
            acc = fma(wg[3 * r + 1], win[r][i1], acc);
            acc = fma(wg[3 * r + 2], win[r][i2], acc);
        }
        float o = (float)acc;
        if (relu) o = fmaxf(o, 0.f);
        Elem<T>::st(orow + (size_t)px * 64, o);
    };
    int px = 0;
    for (; px + 3 <= npx; px += 3) { pixel(px, 0, 1, 2); pixel(px + 1, 1, 2, 0); pixel(px + 2, 2, 0, 1); }   // the window rotates: no register moves
    if (px < npx) { pixel(px, 0, 1, 2); ++px; }
    if (px < npx) pixel(px, 1, 2, 0);
}

}  // namespace

template <typename T> int image_to_nhwc_launch(const float* img, void* out, int N, int H, int W, int CP, hipStream_t s) {
    const size_t total = (size_t)N * H * W * CP;
    hipLaunchKernelGGL((image_to_nhwc_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, img, (T*)out, H * W, CP, total);
    return master_check_launch("image_to_nhwc");
}
template <typename T> int maxpool2_launch(const void* x, void* out, int N, int H, int W, int C, hipStream_t s) {
    const size_t total = (size_t)N * (H / 2) * (W / 2) * C;
    hipLaunchKernelGGL((maxpool2_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const T*)x, (T*)out, H, W, C, total);
    return master_check_launch("maxpool2");
}
// C must be a multiple of the 16-byte channel vector and C / V <= 256 (VGG: 128, 256, 512); workspace = 2 * N * nslice * C floats
int stats_slices(int HW) { return (HW + SLICE_ROWS - 1) / SLICE_ROWS; }
template <typename T> int chan_stats_launch(const void* x, float* mean, float* var, float* work, int N, int HW, int C, hipStream_t s) {
    const int ns = stats_slices(HW);
    float* psum = work; float* psq = work + (size_t)N * ns * C;
    const size_t lds = (size_t)(256 / (C / VecLd<T>::V)) * C * sizeof(float);
    hipLaunchKernelGGL((chan_partial_kernel<T, 0>), dim3(ns, N), dim3(256), lds, s, (const T*)x, (const float*)nullptr, psum, HW, C, ns);
    hipLaunchKernelGGL((chan_partial_kernel<T, 1>), dim3(ns, N), dim3(256), lds, s, (const T*)x, (const float*)psum, psq, HW, C, ns);
    hipLaunchKernelGGL(stats_finalize_kernel, dim3((N * C + 255) / 256), dim3(256), 0, s, (const float*)psum, (const float*)psq, mean, var, HW, C, ns, N * C);
    return master_check_launch("chan_stats");
}
template <typename T> int in_dist_launch(const void* x, const float* mean, const float* var, float* partial, int pairs, int pair_off, int HW,
                                         int C, float eps, int squared, hipStream_t s) {
    hipLaunchKernelGGL((in_dist_kernel<T>), dim3(stats_slices(HW), pairs), dim3(256), 0, s, (const T*)x, mean, var, partial, pair_off, HW, C, eps, squared);
    return master_check_launch("in_dist");
}
template <typename T> int vgg_conv1_launch(const float* img, const float* w, const float* bias, void* out, int N, int H, int W, int relu, hipStream_t s) {
    hipLaunchKernelGGL((vgg_conv1_kernel<T>), dim3((W + 63) / 64, (H + 3) / 4, N), dim3(256), 0, s, img, w, bias, (T*)out, H, W, relu);
    return master_check_launch("vgg_conv1");
}
template int vgg_conv1_launch<float>(const float*, const float*, const float*, void*, int, int, int, int, hipStream_t);
template int vgg_conv1_launch<bf16_t>(const float*, const float*, const float*, void*, int, int, int, int, hipStream_t);
template int image_to_nhwc_launch<float>(const float*, void*, int, int, int, int, hipStream_t);
template int image_to_nhwc_launch<bf16_t>(const float*, void*, int, int, int, int, hipStream_t);
template int maxpool2_launch<float>(const void*, void*, int, int, int, int, hipStream_t);
template int maxpool2_launch<bf16_t>(const void*, void*, int, int, int, int, hipStream_t);
template int chan_stats_launch<float>(const void*, float*, float*, float*, int, int, int, hipStream_t);
template int chan_stats_launch<bf16_t>(const void*, float*, float*, float*, int, int, int, hipStream_t);
template int in_dist_launch<float>(const void*, const float*, const float*, float*, int, int, int, int, float, int, hipStream_t);
template int in_dist_launch<bf16_t>(const void*, const float*, const float*, float*, int, int, int, int, float, int, hipStream_t);
