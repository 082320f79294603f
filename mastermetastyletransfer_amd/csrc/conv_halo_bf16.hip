// 3x3 reflect-pad convolution (+ optional nearest x2 upsample of the input, + ReLU) for Cin = 64, bf16 throughput mode
// (decoder.py:41-45: Upsample, Conv2d(64,64), ReLU, Conv2d(64,32), ReLU).
//
// The implicit-GEMM kernel re-gathers every input pixel 9 times from L2 and pays a load->LDS->barrier round trip per
// 64-deep K chunk; with K = 576 and 32-64 output channels it is latency-bound (166-370 TFLOP/s).  Here a workgroup owns a
// 16 x 32 (8 x 32 without upsample) tile of output pixels:
//   * the source halo tile (18x34 pixels, or 10x18 of the half-resolution map when the upsample is fused, clamped at the
//     image edge) is gathered into LDS ONCE, 144-B pixel records (16 consecutive pixels -> 16 distinct 16-B slots);
//   * a wave's 32 output channels x 576 K live in registers as 36 MFMA A fragments, loaded once;
//   * Y^T[n][pix] = sum_k W[n][k] X[pix][k]: the pixel sits on the MFMA lane, X fragments are 16-B LDS reads whose address
//     folds tap + reflect padding + (>>1 for the upsample);  36 MFMAs per 32-pixel tile, no barrier inside the tile loop;
//   * epilogue: bias + ReLU, NHWC bf16 stores (4 channels = 8 B per lane, the pixel's row of Cout channels is contiguous).
// Cout = 64: wave pairs take one 32-channel half each.
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int CIN = 64, TW = 32, PIX = CIN * 2 + 16;               // 144-B pixel record; tile = TH x 32 pixels, TH = 16 (upsampled) or 8
constexpr int KSTEPS = 9 * CIN / 16;                                // 36

struct HaloArgs { const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* out; int B, Hin, Win, relu; };

union Frag { u32x4 u; bf16x8_t b; };

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

template <int COUT, bool UP>
__global__ __launch_bounds__(256, 2) void conv_halo_bf16_kernel(HaloArgs p) {
    constexpr int TH = UP ? 16 : 8;             // 26-KB (upsampled source) / 49-KB halo tile: two workgroups per CU either way
    constexpr int SH = UP ? TH / 2 + 2 : TH + 2, SW = UP ? TW / 2 + 2 : TW + 2;   // source tile
    constexpr int NHALF = COUT / 32;                                              // 32-channel halves (1 or 2)
    __shared__ __attribute__((aligned(16))) unsigned char S[SH * SW * PIX];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int H = UP ? 2 * p.Hin : p.Hin, W = UP ? 2 * p.Win : p.Win;
    const int tiles_x = W / TW, tiles_y = H / TH;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; const int b = bid / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int sy0 = UP ? (y0 - 1) >> 1 : y0 - 1, sx0 = UP ? (x0 - 1) >> 1 : x0 - 1;   // window origin (clamped rows at the edge)

    // ---- this wave's output-channel half: 36 A fragments + bias ------------------------------------------------------
    const int nh = NHALF == 1 ? 0 : (wave >> 1);
    Frag wf[KSTEPS];
    {
        const bf16_t* wp = p.w + (size_t)(nh * 32 + r) * (9 * CIN) + 8 * h;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) wf[s].u = *(const u32x4*)(wp + 16 * s);
    }
    float bv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) bv[e] = p.bias[nh * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];

    // ---- source halo tile -> LDS -------------------------------------------------------------------------------------------
    for (int c = tid; c < SH * SW * 8; c += 256) {
        const int pix = c >> 3, part = c & 7;
        const int sy = clampi(sy0 + pix / SW, 0, p.Hin - 1), sx = clampi(sx0 + pix % SW, 0, p.Win - 1);
        *(u32x4*)(S + pix * PIX + part * 16) = *(const u32x4*)(p.x + (((size_t)b * p.Hin + sy) * p.Win + sx) * CIN + part * 8);
    }
    __syncthreads();

    // ---- 32-pixel tiles = the 16 rows of the output tile; wave (pair) w takes rows w, w+4, ... (w&1, +2, ... for Cout = 64) --
    const int first = NHALF == 1 ? wave : (wave & 1), stride = NHALF == 1 ? 4 : 2;
    for (int row = first; row < TH; row += stride) {
        const int ay = y0 + row, ax = x0 + r;                          // this lane's output pixel
        int ro[3], co[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int uy = reflect1(ay + d - 1, H), ux = reflect1(ax + d - 1, W);
            ro[d] = ((UP ? uy >> 1 : uy) - sy0) * SW;
            co[d] = (UP ? ux >> 1 : ux) - sx0;
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = bv[e];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const unsigned char* px = S + (ro[t / 3] + co[t % 3]) * PIX + 16 * h;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                Frag xf; xf.u = *(const u32x4*)(px + 32 * s4);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[4 * t + s4].b, xf.b, acc, 0, 0, 0);
            }
        }
        bf16_t* op = p.out + (((size_t)b * H + ay) * W + ax) * COUT + nh * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v0 = acc[4 * g], v1 = acc[4 * g + 1], v2 = acc[4 * g + 2], v3 = acc[4 * g + 3];
            if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            u32x2 o; o[0] = pack_bf16x2(v0, v1); o[1] = pack_bf16x2(v2, v3);
            *(u32x2*)(op + 8 * g) = o;
        }
    }
}

template <int COUT, bool UP>
int launch(const HaloArgs& p, hipStream_t s) {
    const int H = UP ? 2 * p.Hin : p.Hin, W = UP ? 2 * p.Win : p.Win;
    constexpr int TH = UP ? 16 : 8;
    hipLaunchKernelGGL((conv_halo_bf16_kernel<COUT, UP>), dim3(p.B * (H / TH) * (W / TW)), dim3(256), 0, s, p);
    return master_check_launch("conv_halo_bf16");
}

}  // namespace

// returns MASTER_ERR_UNSUPPORTED when the shape is not covered (caller falls back to the implicit-GEMM kernel)
int conv_halo_bf16(const bf16_t* x, const bf16_t* w, const float* bias, bf16_t* out, int B, int Hin, int Win, int Cin, int Cout,
                   int up, int relu, hipStream_t s) {
    const int H = Hin << up, W = Win << up;
    if (Cin != CIN || (Cout != 32 && Cout != 64) || H % (up ? 16 : 8) != 0 || W % TW != 0 || bias == nullptr) return MASTER_ERR_UNSUPPORTED;
    HaloArgs p{x, w, bias, out, B, Hin, Win, relu};
    if (Cout == 32) return up ? launch<32, true>(p, s) : launch<32, false>(p, s);
    return up ? launch<64, true>(p, s) : launch<64, false>(p, s);
}
