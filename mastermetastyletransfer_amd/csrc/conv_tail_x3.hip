// Decoder tail of the split-bf16 fast parity mode (MASTER_F32X3): nearest x2 upsample -> Conv3x3(32->32, reflect) + ReLU ->
// Conv3x3(32->3, reflect) in ONE launch (reference codes/decoder.py:47-54), every product as three bf16 MFMAs on split operands
// (x = hi + lo, a*w = a_lo*w_hi + a_hi*w_lo + a_hi*w_hi, fp32 accumulation; gemm_x3.hip).
//
// As two launches this is the most expensive piece of the mode (r04 trace, bs = 32: 355 us for the 32 -> 32 layer as an implicit GEMM
// whose nine taps re-read the map from L2 - 2.4 GB of L2 -> LDS traffic - plus 161 us for the direct 32 -> 3 kernel, with the 268-MB
// full-resolution 32-channel intermediate written and read in between).  The shape of conv_tail_bf16.hip:
//   A. the source halo tile ((8/2+2) x (32/2+2) pixels x 32 ch of the half-resolution input, clamped at the image edge) goes to LDS
//      once, as split pixel records [32 bf16 hi | 32 bf16 lo | 16 B pad] (144 B: 16 neighbouring pixels hit 16 distinct 16-byte
//      slots) - copied as they lie when the input map is already in the split layout (MASTER_A_SPLIT), split on the way otherwise;
//   B. conv 32->32 on the (8+2) x (32+2) region the second conv needs (its reflect padding resolved to real coordinates): implicit
//      GEMM, transposed so that the pixel sits on the MFMA lane: Y^T[n][pix] = sum_k W8[n][k] X[pix][k], 18 k-steps x 3 MFMAs; the
//      W8 fragments (hi and lo: 144 VGPRs) live in registers, X fragments are 16-B LDS reads with tap / reflect / >>1 folded into
//      the address; bias + ReLU in fp32, split, to LDS as the same pixel records;
//   C. conv 32->3 with v_mfma_f32_16x16x32_bf16 (one tap = K = 32 channels = 3 MFMAs; 3 of 16 output rows real), W9 fragments in
//      registers, fp32 NCHW stores of 16 consecutive pixels per channel.
// An 8 x 32 output tile keeps a workgroup at 64 KB of LDS and under 256 registers: two workgroups per CU.
#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct TailX3Args {
    const unsigned char* x; const unsigned char* w8; const float* b8; const unsigned char* w9; const float* b9; float* out;
    int B, Hs, Ws, x_split;
};

constexpr int TH = 8, TW = 32;                  // output tile
constexpr int SH = TH / 2 + 2, SW = TW / 2 + 2; // source (half-resolution) halo tile
constexpr int RH = TH + 2, RW = TW + 2;         // region of the first conv's outputs the second conv reads
constexpr int RP = RH * RW;
constexpr int PIX = 144;                        // bytes per pixel record in LDS: 64 B hi | 64 B lo | 16 B pad
constexpr int S_BYTES = SH * SW * PIX, Y_BYTES = RP * PIX;

struct Frag { u32x4 u; };        // 8 sixteen-bit halves: bf16 or IEEE half, by FMT

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
__device__ __forceinline__ int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

template <int FMT>
__global__ __launch_bounds__(256, 2) void conv_tail_x3_kernel(TailX3Args p) {
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

    // ---- A. source halo tile -> LDS (split records) ---------------------------------------------------------------------
    if (p.x_split) {
        for (int c = tid; c < SH * SW * 8; c += 256) {
            const int pix = c >> 3, part = c & 7;
            const int sy = clampi(sy0 + pix / SW, 0, p.Hs - 1), sx = clampi(sx0 + pix % SW, 0, p.Ws - 1);
            *(u32x4*)(S + pix * PIX + part * 16) = *(const u32x4*)(p.x + (((size_t)b * p.Hs + sy) * p.Ws + sx) * 128 + part * 16);
        }
    } else {
        for (int c = tid; c < SH * SW * 4; c += 256) {
            const int pix = c >> 2, part = c & 3;                   // 8 channels per part
            const int sy = clampi(sy0 + pix / SW, 0, p.Hs - 1), sx = clampi(sx0 + pix % SW, 0, p.Ws - 1);
            const float* src = (const float*)p.x + (((size_t)b * p.Hs + sy) * p.Ws + sx) * 32 + part * 8;
            const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
            u32x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                unsigned ah, al, ch, cl;
                split_x2<FMT>(v0[2 * j], v0[2 * j + 1], ah, al);
                split_x2<FMT>(v1[2 * j], v1[2 * j + 1], ch, cl);
                hi[j] = ah; lo[j] = al; hi[2 + j] = ch; lo[2 + j] = cl;
            }
            *(u32x4*)(S + pix * PIX + part * 16) = hi;
            *(u32x4*)(S + pix * PIX + 64 + part * 16) = lo;
        }
    }

    // ---- B. first conv on the RH x RW region, 32 region pixels per MFMA tile ------------------------------------
    {
        const int r = lane & 31, h = lane >> 5;
        // W8 fragments (A operand 32x32x16): lane (n = r, h): W8[n][16 s + 8 h .. +8], k = 32 tap + channel; split layout: chunk = tap
        Frag w8h[18], w8l[18];
        {
            const unsigned char* wp = p.w8 + (size_t)r * (288 * 4) + 16 * h;
#pragma unroll
            for (int s = 0; s < 18; ++s) {
                w8h[s].u = *(const u32x4*)(wp + (s >> 1) * 128 + (s & 1) * 32);
                w8l[s].u = *(const u32x4*)(wp + (s >> 1) * 128 + 64 + (s & 1) * 32);
            }
        }
        float b8v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) b8v[e] = p.b8[(e & 3) + 8 * (e >> 2) + 4 * h];
        __syncthreads();                                              // the halo tile is complete
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
                    Frag xh, xl; xh.u = *(const u32x4*)(px + 32 * s2); xl.u = *(const u32x4*)(px + 64 + 32 * s2);
                    acc = mfma_x3_32<FMT>(w8h[2 * t + s2].u, xl.u, acc);
                    acc = mfma_x3_32<FMT>(w8l[2 * t + s2].u, xh.u, acc);
                    acc = mfma_x3_32<FMT>(w8h[2 * t + s2].u, xh.u, acc);
                }
            }
            if (pr < RP) {
                // acc[e]: channel n = (e&3) + 8(e>>2) + 4h of region pixel pr -> bias, ReLU, split; 4 channels per 8-B write
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = relu_np(acc[4 * g + e] + b8v[4 * g + e]);
                    unsigned h0, h1, l0, l1;
                    split_x2<FMT>(v[0], v[1], h0, l0);
                    split_x2<FMT>(v[2], v[3], h1, l1);
                    *(u32x2*)(Y + pr * PIX + (8 * g + 4 * h) * 2) = u32x2{h0, h1};
                    *(u32x2*)(Y + pr * PIX + 64 + (8 * g + 4 * h) * 2) = u32x2{l0, l1};
                }
            }
        }
    }
    __syncthreads();

    // ---- C. second conv (3 output channels), 16 output pixels per MFMA tile ---------------------------------------
    {
        const int q16 = lane & 15, kg = lane >> 4;
        Frag w9h[9], w9l[9];                                          // A operand 16x16x32: lane (o = l&15, kg): W9[o][32 tap + 8 kg .. +8]
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            w9h[t].u = u32x4{0u, 0u, 0u, 0u}; w9l[t].u = u32x4{0u, 0u, 0u, 0u};
            if (q16 < 3) {
                const unsigned char* wp = p.w9 + (size_t)q16 * (288 * 4) + t * 128 + 16 * kg;
                w9h[t].u = *(const u32x4*)wp; w9l[t].u = *(const u32x4*)(wp + 64);
            }
        }
        const float bo0 = p.b9[0], bo1 = p.b9[1], bo2 = p.b9[2];
        const size_t plane = (size_t)H * W;
        for (int tile = wave; tile < TH * TW / 16; tile += 4) {
            const int q = tile * 16 + q16, oy = q / TW, ox = q % TW;
            const unsigned char* py = Y + (oy * RW + ox) * PIX + 16 * kg;     // region index of tap (0,0) = virtual (oy-1, ox-1)
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const unsigned char* pt = py + ((t / 3) * RW + (t % 3)) * PIX;
                Frag yh, yl; yh.u = *(const u32x4*)pt; yl.u = *(const u32x4*)(pt + 64);
                acc = mfma_x3_16<FMT>(w9h[t].u, yl.u, acc);
                acc = mfma_x3_16<FMT>(w9l[t].u, yh.u, acc);
                acc = mfma_x3_16<FMT>(w9h[t].u, yh.u, acc);
            }
            if (kg == 0) {                                   // rows 0..3 of the 16x16 result live in lanes 0..15
                float* op = p.out + (size_t)b * 3 * plane + (size_t)(y0 + oy) * W + x0 + ox;
                op[0] = acc[0] + bo0; op[plane] = acc[1] + bo1; op[2 * plane] = acc[2] + bo2;
            }
        }
    }
}

}  // namespace

int conv_tail_x3(const void* x, int x_split, const void* w8, const float* b8, const void* w9, const float* b9, float* out, int B, int Hs, int Ws,
                 hipStream_t s, int fmt) {
    if ((2 * Hs) % TH != 0 || (2 * Ws) % TW != 0) {
        master_set_error("decoder_tail(x3): output %dx%d is not a multiple of the %dx%d tile", 2 * Hs, 2 * Ws, TH, TW);
        return MASTER_ERR_UNSUPPORTED;
    }
    TailX3Args p{(const unsigned char*)x, (const unsigned char*)w8, b8, (const unsigned char*)w9, b9, out, B, Hs, Ws, x_split};
    const int grid = B * (2 * Hs / TH) * (2 * Ws / TW);
    if (fmt == X3_F16) hipLaunchKernelGGL(conv_tail_x3_kernel<X3_F16>, dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(conv_tail_x3_kernel<X3_BF16>, dim3(grid), dim3(256), 0, s, p);
    return master_check_launch("conv_tail_x3");
}
