// Backward of the decoder's last convolution, Conv2d(Cin, 3, 3, padding_mode='reflect') without ReLU (reference codes/decoder.py:54),
// fp32, direct (r03).  As implicit GEMMs the layer's 3 output channels were zero-padded to 32: a 67-MB padded NHWC copy of the image
// gradient (master_relu_mask), a data-gradient GEMM and a weight-gradient GEMM that each did 32/3 of the useful work (~0.5 ms of the
// train step for 2 GFLOP).  Both kernels read the image gradient as it arrives from the loss: NCHW [B,3,H,W].
//
//   data gradient    dxp[n][py][px][ci] = sum_{co,ky,kx} gz[n][co][py-ky][px-kx] w[co][ci][ky][kx]   on the PADDED domain (H+2) x (W+2)
//                    (gz = g zero-extended; master_fold_reflect then folds the mirror rows / columns back, as for every decoder layer):
//                    workgroup = 8 rows x 64 columns, half-wave = one row, lane & 31 = input channel ci; the lane's 27 (flipped) weights
//                    live in registers, the 3 x 10 x 66 tile of g in LDS, every g value is a half-wave-uniform LDS read
//                    (3-wide sliding windows: 9 reads per pixel), a pixel's 32 channels leave as one 128-byte row.
//   weight gradient  dw[co][ci][ky][kx] = sum_{n,y,x} g[n][co][y][x] xr[n][y+ky-1][x+kx-1][ci]   (xr = reflect-padded input), and
//                    dbias[co] = sum g: workgroup = 8 rows x 64 columns of one image, half-wave = one row, lane & 31 = ci; the x
//                    values slide through registers (three 128-byte row loads per pixel), the g tile sits in LDS; 27 accumulators per
//                    lane, folded over the workgroup's 8 rows through LDS into one slab of 3 x Cin x 9 (+ 3) partial sums per
//                    workgroup; the caller's slab reduce adds the workgroups in a fixed order.
#include "common.h"

namespace {

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

constexpr int CIN = 32, TROWS = 8, TCOLS = 64;

__global__ __launch_bounds__(256) void conv_cout3_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ dxp,
                                                               int H, int W) {
    __shared__ float tile[3][TROWS + 2][TCOLS + 4];              // g rows y0-2 .. y0+7, columns x0-2 .. x0+63 (zero outside the image)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ho = H + 2, Wo = W + 2;
    const int n = blockIdx.z, y0 = blockIdx.y * TROWS, x0 = blockIdx.x * TCOLS;
    for (int i = tid; i < 3 * (TROWS + 2) * (TCOLS + 2); i += 256) {
        const int co = i / ((TROWS + 2) * (TCOLS + 2)), rem = i - co * ((TROWS + 2) * (TCOLS + 2)), ry = rem / (TCOLS + 2), rx = rem - ry * (TCOLS + 2);
        const int gy = y0 - 2 + ry, gx = x0 - 2 + rx;
        tile[co][ry][rx] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? g[(((size_t)n * 3 + co) * H + gy) * W + gx] : 0.f;
    }
    const int ci = lane & 31, row = 2 * wave + (lane >> 5);
    float wg[27];                                                // [co][ky'][kx'] with the taps flipped: g row py - ky = (py - 2) + ky', ky' = 2 - ky
#pragma unroll
    for (int co = 0; co < 3; ++co)
#pragma unroll
        for (int k = 0; k < 9; ++k) wg[co * 9 + k] = w[((size_t)co * CIN + ci) * 9 + (8 - k)];
    __syncthreads();
    const int y = y0 + row;
    if (y >= Ho) return;
    int npx = Wo - x0; npx = npx > TCOLS ? TCOLS : npx;
    float* orow = dxp + (((size_t)n * Ho + y) * Wo + x0) * CIN + ci;
    float win[9][3];                                             // [(co, ky')][kx']
#pragma unroll
    for (int r = 0; r < 9; ++r) { win[r][0] = tile[r / 3][row + r % 3][0]; win[r][1] = tile[r / 3][row + r % 3][1]; }
    auto pixel = [&](int px, int i0, int i1, int i2) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 9; ++r) win[r][i2] = tile[r / 3][row + r % 3][px + 2];
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            acc = fmaf(wg[3 * r + 0], win[r][i0], acc);
            acc = fmaf(wg[3 * r + 1], win[r][i1], acc);
            acc = fmaf(wg[3 * r + 2], win[r][i2], acc);
        }
        orow[(size_t)px * CIN] = acc;
    };
    int px = 0;
    for (; px + 3 <= npx; px += 3) { pixel(px, 0, 1, 2); pixel(px + 1, 1, 2, 0); pixel(px + 2, 2, 0, 1); }
    if (px < npx) { pixel(px, 0, 1, 2); ++px; }
    if (px < npx) pixel(px, 1, 2, 0);
}

// part[blk][3*CIN*9 + 4]: the workgroup's partial dw in nn.Conv2d's [co][ci][ky][kx] order, then its three partial bias sums (+ 1 pad)
__global__ __launch_bounds__(256) void conv_cout3_wgrad_kernel(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ part,
                                                               int H, int W) {
    __shared__ float gt[3][TROWS][TCOLS];
    __shared__ float red[8][27][CIN + 1];                        // [half-wave][(co, tap)][ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.z, y0 = blockIdx.y * TROWS, x0 = blockIdx.x * TCOLS;
    float bsum = 0.f;                                            // thread t < 3*64: partial bias sums while loading
    for (int i = tid; i < 3 * TROWS * TCOLS; i += 256) {
        const int co = i / (TROWS * TCOLS), rem = i - co * (TROWS * TCOLS), ry = rem / TCOLS, rx = rem - ry * TCOLS;
        const int gy = y0 + ry, gx = x0 + rx;
        gt[co][ry][rx] = (gy < H && gx < W) ? g[(((size_t)n * 3 + co) * H + gy) * W + gx] : 0.f;
    }
    __syncthreads();
    const int ci = lane & 31, hw = 2 * wave + (lane >> 5), row = hw;
    const int y = y0 + row;
    float acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.f;
    if (y < H) {
        int npx = W - x0; npx = npx > TCOLS ? TCOLS : npx;
        const float* xr[3];                                      // the three reflect-padded input rows under this output row
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) xr[ky] = x + (((size_t)n * H + reflect1(y + ky - 1, H)) * W) * CIN + ci;
        float win[3][3];                                         // [ky][kx]
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            win[ky][0] = xr[ky][(size_t)reflect1(x0 - 1, W) * CIN];
            win[ky][1] = xr[ky][(size_t)x0 * CIN];
        }
        auto pixel = [&](int px, int i0, int i1, int i2) __attribute__((always_inline)) {
            const int xn = reflect1(x0 + px + 1, W);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) win[ky][i2] = xr[ky][(size_t)xn * CIN];
#pragma unroll
            for (int co = 0; co < 3; ++co) {
                const float gv = gt[co][row][px];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    acc[co * 9 + ky * 3 + 0] = fmaf(gv, win[ky][i0], acc[co * 9 + ky * 3 + 0]);
                    acc[co * 9 + ky * 3 + 1] = fmaf(gv, win[ky][i1], acc[co * 9 + ky * 3 + 1]);
                    acc[co * 9 + ky * 3 + 2] = fmaf(gv, win[ky][i2], acc[co * 9 + ky * 3 + 2]);
                }
            }
        };
        int px = 0;
        for (; px + 3 <= npx; px += 3) { pixel(px, 0, 1, 2); pixel(px + 1, 1, 2, 0); pixel(px + 2, 2, 0, 1); }
        if (px < npx) { pixel(px, 0, 1, 2); ++px; }
        if (px < npx) pixel(px, 1, 2, 0);
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) red[hw][k][ci] = acc[k];
    __syncthreads();
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float* out = part + blk * (size_t)(3 * CIN * 9 + 4);
    for (int i = tid; i < 27 * CIN; i += 256) {                  // i = (co, tap) * CIN + ci -> dw index co*CIN*9 + ci*9 + tap
        const int k = i / CIN, c = i - k * CIN;
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q][k][c];
        out[(k / 9) * (CIN * 9) + c * 9 + (k % 9)] = t;
    }
    if (tid < 4) {                                               // bias: sum of the workgroup's g tile, one thread per channel (fixed order)
        if (tid < 3) {
            for (int ry = 0; ry < TROWS; ++ry)
                for (int rx = 0; rx < TCOLS; ++rx) bsum += gt[tid][ry][rx];
        }
        out[3 * CIN * 9 + tid] = tid < 3 ? bsum : 0.f;
    }
}

}  // namespace

int reduce_segments_launch(const float* const* part, float* const* out, const size_t* n, const size_t* slab, const int* splits, const int* accumulate,
                           int count, hipStream_t s);                                                                       // train_misc.hip

// g: NCHW [B,3,H,W]; w: [3][32][3][3]; dxp: [B,H+2,W+2,32]
int conv_cout3_dgrad_launch(const float* g, const float* w, float* dxp, int B, int H, int W, hipStream_t s) {
    hipLaunchKernelGGL(conv_cout3_dgrad_kernel, dim3((W + 2 + TCOLS - 1) / TCOLS, (H + 2 + TROWS - 1) / TROWS, B), dim3(256), 0, s, g, w, dxp, H, W);
    return master_check_launch("conv_cout3_dgrad");
}
size_t conv_cout3_wgrad_workspace(int B, int H, int W) {
    return (size_t)B * ((H + TROWS - 1) / TROWS) * ((W + TCOLS - 1) / TCOLS) * (size_t)(3 * CIN * 9 + 4);
}
// x: NHWC [B,H,W,32]; dw: [3][32][3][3]; dbias: [3] (may be NULL); ws: conv_cout3_wgrad_workspace(B, H, W) floats
int conv_cout3_wgrad_launch(const float* g, const float* x, float* dw, float* dbias, float* ws, int B, int H, int W, hipStream_t s) {
    const dim3 grid((W + TCOLS - 1) / TCOLS, (H + TROWS - 1) / TROWS, B);
    hipLaunchKernelGGL(conv_cout3_wgrad_kernel, grid, dim3(256), 0, s, g, x, ws, H, W);
    const int blocks = (int)(grid.x * grid.y * grid.z);
    const size_t slab = 3 * CIN * 9 + 4;
    const float* parts[2] = {ws, ws + 3 * CIN * 9}; float* outs[2] = {dw, dbias};
    const size_t ns[2] = {(size_t)3 * CIN * 9, 3}, slabs[2] = {slab, slab};
    const int sp[2] = {blocks, blocks}, ac[2] = {0, 0};
    return reduce_segments_launch(parts, outs, ns, slabs, sp, ac, dbias ? 2 : 1, s);
}
