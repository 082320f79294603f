// Decoder tail, bf16 throughput mode: nearest x2 upsample -> Conv3x3(32->32, reflect) + ReLU -> Conv3x3(32->3,
// reflect) in ONE launch (reference codes/decoder.py:47-54: Upsample, Conv2d(c/8,c/8), ReLU, Conv2d(c/8,3)).
// The 32-channel full-resolution intermediate (134 MB at bs=32, the largest tensor of the network) stays in LDS.
//
// Workgroup = one 16 x 32 tile of output pixels of one image, 4 waves:
//   A. the source halo tile ((16/2+2) x (32/2+2) pixels x 32 ch of the half-resolution input, clamped at the
//      image edge) is gathered into LDS once: every input pixel is read from HBM/L2 once instead of 9 x 4 times;
//   B. conv 32->32 on the (16+2) x (32+2) region the second conv needs (its own reflect padding is resolved to
//      real coordinates, so border tiles just recompute a few pixels): implicit GEMM, transposed so the pixel
//      sits on the MFMA lane:  Y^T[n][pix] = sum_k W8[n][k] X[pix][k],  K = 9 taps x 32 ch = 18 k-steps of
//      v_mfma_f32_32x32x16_bf16; W8 fragments live in registers (72 VGPRs), X fragments are 16-B LDS reads of the
//      halo tile with the tap / reflect / >>1 (upsample) folded into the address;  bias + ReLU, bf16, to LDS;
//   C. conv 32->3 with v_mfma_f32_16x16x32_bf16 (one tap = one MFMA: K = 32 channels; 3 of 16 output rows real),
//      W9 fragments in registers, fp32 NCHW stores of 16 consecutive pixels (64 B) per channel.
// LDS pixel records are padded to 80 B so that 16 consecutive pixels hit 16 distinct 16-B slots.
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct TailArgs {
    const bf16_t* x; const bf16_t* w8; const float* b8; const bf16_t* w9; const float* b9; float* out;
    int B, Hs, Ws;
};

constexpr int TH = 16, TW = 32;                 // output tile
constexpr int SH = TH / 2 + 2, SW = TW / 2 + 2; // source (half-resolution) halo tile
constexpr int RH = TH + 2, RW = TW + 2;         // region of the first conv's outputs the second conv reads
constexpr int RP = RH * RW;
constexpr int PIX = 80;                         // bytes per pixel record in LDS (64 B of channels + 16 B pad)
constexpr int S_BYTES = SH * SW * PIX, Y_BYTES = RP * PIX;

union Frag { u32x4 u; bf16x8_t b; };

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

__global__ __launch_bounds__(256) void conv_tail_bf16_kernel(TailArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[S_BYTES + Y_BYTES];
    unsigned char* S = smem;
    unsigned char* Y = smem + S_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = 2 * p.Hs, W = 2 * p.Ws;
    const int tiles_x = W / TW, tiles_y = H / TH;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; const int b = bid / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int sy0 = (y0 - 2) >> 1, sx0 = (x0 - 2) >> 1;          // window origin (may be -1 at the edge: rows are clamped)

    // ---- weights to registers ---------------------------------------------------------------------------
    Frag w8f[18];                                                 // A operand 32x32x16: lane (n = l&31, h): W8[n][16s + 8h ..]
    {
        const bf16_t* wp = p.w8 + (lane & 31) * 288 + 8 * (lane >> 5);
#pragma unroll
        for (int s = 0; s < 18; ++s) w8f[s].u = *(const u32x4*)(wp + 16 * s);
    }
    Frag w9f[9];                                                  // A operand 16x16x32: lane (o = l&15, kg = l>>4): W9[o][32 tap + 8 kg ..]
    {
        const int o = lane & 15, kg = lane >> 4;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            w9f[t].u = u32x4{0u, 0u, 0u, 0u};
            if (o < 3) w9f[t].u = *(const u32x4*)(p.w9 + o * 288 + t * 32 + 8 * kg);
        }
    }

    // ---- A. source halo tile -> LDS -------------------------------------------------------------------------
    for (int c = tid; c < SH * SW * 4; c += 256) {
        const int pix = c >> 2, part = c & 3;
        const int sy = clampi(sy0 + pix / SW, 0, p.Hs - 1), sx = clampi(sx0 + pix % SW, 0, p.Ws - 1);
        *(u32x4*)(S + pix * PIX + part * 16) = *(const u32x4*)(p.x + (((size_t)b * p.Hs + sy) * p.Ws + sx) * 32 + part * 8);
    }
    __syncthreads();

    // ---- B. first conv on the RH x RW region, 32 region pixels per MFMA tile ------------------------------------
    {
        const int r = lane & 31, h = lane >> 5;
        float b8v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) b8v[e] = p.b8[(e & 3) + 8 * (e >> 2) + 4 * h];
        for (int tile = wave; tile < (RP + 31) / 32; tile += 4) {
            const int pr = tile * 32 + r;
            const int pc = pr < RP ? pr : RP - 1;
            const int ay = reflect1(y0 + pc / RW - 1, H), ax = reflect1(x0 + pc % RW - 1, W);   // the second conv's reflect padding
            int ro[3], co[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                ro[d] = ((reflect1(ay + d - 1, H) >> 1) - sy0) * SW;       // first conv's reflect padding, then nearest upsample
                co[d] = (reflect1(ax + d - 1, W) >> 1) - sx0;
            }
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const unsigned char* px = S + (ro[t / 3] + co[t % 3]) * PIX + 16 * h;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    Frag xf; xf.u = *(const u32x4*)(px + 32 * s2);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w8f[2 * t + s2].b, xf.b, acc, 0, 0, 0);
                }
            }
            if (pr < RP) {
                // acc[e]: channel n = (e&3) + 8(e>>2) + 4h of region pixel pr -> bias, ReLU, 4 channels per 8-B write
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 o;
                    o[0] = pack_bf16x2(fmaxf(acc[4 * g] + b8v[4 * g], 0.f), fmaxf(acc[4 * g + 1] + b8v[4 * g + 1], 0.f));
                    o[1] = pack_bf16x2(fmaxf(acc[4 * g + 2] + b8v[4 * g + 2], 0.f), fmaxf(acc[4 * g + 3] + b8v[4 * g + 3], 0.f));
                    *(u32x2*)(Y + pr * PIX + (8 * g + 4 * h) * 2) = o;
                }
            }
        }
    }
    __syncthreads();

    // ---- C. second conv (3 output channels), 16 output pixels per MFMA tile ---------------------------------------
    {
        const int q16 = lane & 15, kg = lane >> 4;
        const float bo0 = p.b9[0], bo1 = p.b9[1], bo2 = p.b9[2];
        const size_t plane = (size_t)H * W;
        for (int tile = wave; tile < TH * TW / 16; tile += 4) {
            const int q = tile * 16 + q16, oy = q / TW, ox = q % TW;
            const unsigned char* py = Y + (oy * RW + ox) * PIX + 16 * kg;     // region index of tap (0,0) = virtual (oy-1, ox-1)
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                Frag yf; yf.u = *(const u32x4*)(py + ((t / 3) * RW + (t % 3)) * PIX);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w9f[t].b, yf.b, acc, 0, 0, 0);
            }
            if (kg == 0) {                                   // rows 0..3 of the 16x16 result live in lanes 0..15
                float* op = p.out + (size_t)b * 3 * plane + (size_t)(y0 + oy) * W + x0 + ox;
                op[0] = acc[0] + bo0; op[plane] = acc[1] + bo1; op[2 * plane] = acc[2] + bo2;
            }
        }
    }
}

}  // namespace

int conv_tail_bf16(const bf16_t* x, const bf16_t* w8, const float* b8, const bf16_t* w9, const float* b9, float* out,
                   int B, int Hs, int Ws, hipStream_t s) {
    if ((2 * Hs) % TH != 0 || (2 * Ws) % TW != 0) {
        master_set_error("decoder_tail: output %dx%d is not a multiple of the %dx%d tile", 2 * Hs, 2 * Ws, TH, TW);
        return MASTER_ERR_UNSUPPORTED;
    }
    TailArgs p{x, w8, b8, w9, b9, out, B, Hs, Ws};
    const int grid = B * (2 * Hs / TH) * (2 * Ws / TW);
    hipLaunchKernelGGL(conv_tail_bf16_kernel, dim3(grid), dim3(256), 0, s, p);
    return master_check_launch("conv_tail_bf16");
}
