// Swin patch embedding on the matrix cores, bf16 output: Conv2d(3,128,4,stride 4) + permute + LayerNorm(128)
// (torchvision swin_b.features[0]; reference codes/utils.py:89-93, called at codes/full_model.py:219-220).
//
// A patch is a 48-vector (3 x 4 x 4 pixels), the conv a [128 x 48] GEMM.  Workgroup = 128 consecutive patches:
//   * image rows are read as float4 (the 4 dx of a patch; consecutive patches = consecutive 16 B), converted to bf16
//     and laid out as X[patch][k = c*16 + dy*4 + dx] in LDS (K padded to 64 with zeros, 144-B records);
//   * the weight matrix lives in registers as MFMA A fragments (4 channel tiles x 3 k-steps, converted from fp32 once);
//   * Y^T[e][patch] = sum_k W[e][k] X[patch][k]: patch on the lane, 12 MFMAs per wave;
//   * LayerNorm over the 128 channels = this lane's 64 accumulator values + its half-wave partner's;
//   * the normalised tile is transposed through LDS and stored as whole 256-B token rows.
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
union Frag { u32x4 u; bf16x8_t b; };

constexpr int E = 128, KP = 48, XREC = 144;            // 64 bf16 of K (48 used) + 16 B pad per patch record

struct PeArgs { const float* img; const float* w; const float* bias; const float* gamma; const float* beta; bf16_t* out;
                int B, H, W; float eps; };

__global__ __launch_bounds__(256) void patch_embed_bf16_kernel(PeArgs p) {
    __shared__ __attribute__((aligned(16))) unsigned char X[4 * 8192];     // patch vectors, then reused as the output staging tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int PW = p.W >> 2, PH = p.H >> 2;
    const long npatch = (long)p.B * PH * PW;
    const long p0 = (long)blockIdx.x * 128;

    // ---- pixels: 128 patches x 12 (c, dy) float4 loads, 6 per thread, requested first (they land while the weights are arranged) ------
    f32x4 pv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int item = tid + 256 * i;
        const int pl = item & 127, cd = item >> 7;                  // consecutive threads -> consecutive patches (16 B apart in the row)
        long pg = p0 + pl; pg = pg < npatch ? pg : npatch - 1;
        const int px = (int)(pg % PW); const long t2 = pg / PW; const int py = (int)(t2 % PH); const int b = (int)(t2 / PH);
        const int c = cd >> 2, dy = cd & 3;
        pv[i] = *(const f32x4*)(p.img + (((size_t)b * 3 + c) * p.H + 4 * py + dy) * p.W + 4 * px);
    }

    // ---- weights -> bf16 A fragments: lane (channel tile*32 + r, half h) holds W[e][16s + 8h .. +8], s = 0..2.  r03: the [128][48] fp32
    //      matrix goes through LDS - six coalesced float4 loads per thread, rows padded to 52 floats (16 consecutive rows = 16 distinct
    //      16-B slots) - instead of 24 fragment loads per lane that each touched 32 rows x 32 B (the texture-address path, see
    //      profiles/r03_linear_probe_mi355x.txt) --------------------------------------------------------------------------------------
    constexpr int WLD = 52;
    {
        float* Wl = (float*)X;                                      // 128 x 52 x 4 B = 26.6 KB of the 32-KB region
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int item = tid + 256 * i;                         // float4 index into the dense [128][48] matrix
            const int row = item / 12, c4 = item - row * 12;
            *(f32x4*)(Wl + row * WLD + 4 * c4) = *(const f32x4*)(p.w + (size_t)item * 4);
        }
    }
    __syncthreads();
    Frag wf[4][3];
    {
        const float* Wl = (const float*)X;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const float* wp = Wl + (t * 32 + r) * WLD + 16 * s + 8 * h;
                const f32x4 a = *(const f32x4*)wp, c = *(const f32x4*)(wp + 4);
                wf[t][s].u[0] = pack_bf16x2(a[0], a[1]); wf[t][s].u[1] = pack_bf16x2(a[2], a[3]);
                wf[t][s].u[2] = pack_bf16x2(c[0], c[1]); wf[t][s].u[3] = pack_bf16x2(c[2], c[3]);
            }
    }
    __syncthreads();                                                // every wave holds its fragments: the region becomes the patch records

    // ---- X[patch][c*16 + dy*4 + dx] (bf16) ---------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int item = tid + 256 * i;
        const int pl = item & 127, cd = item >> 7;
        const int c = cd >> 2, dy = cd & 3;
        u32x2 o; o[0] = pack_bf16x2(pv[i][0], pv[i][1]); o[1] = pack_bf16x2(pv[i][2], pv[i][3]);
        *(u32x2*)(X + pl * XREC + (c * 16 + dy * 4) * 2) = o;
    }
    __syncthreads();

    // ---- Y^T[e][patch] for this wave's 32 patches ---------------------------------------------------------------------------------------
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = p.bias[t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
    {
        const unsigned char* xb = X + (wave * 32 + r) * XREC + 16 * h;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            Frag xf; xf.u = *(const u32x4*)(xb + 32 * s);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][s].b, xf.b, acc[t], 0, 0, 0);
        }
    }
    // ---- LayerNorm over the 128 channels of this lane's patch (64 values here, 64 in lane^32) ---------------------------------------
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[t][e];
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / E);
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float d = acc[t][e] - mean; sq += d * d; }
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = rsqrtf(sq * (1.0f / E) + p.eps);

    __syncthreads();                                                // all waves are done reading X: reuse it as the staging tile
    // staging: wave-private [32 patches][128 ch] bf16 = 8 KB, 16-B chunks swizzled by patch&15
    unsigned char* stage = X + wave * 8192;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = t * 32 + 8 * g + 4 * h;
            const f32x4 gm = *(const f32x4*)(p.gamma + ch), bt = *(const f32x4*)(p.beta + ch);
            u32x2 o;
            o[0] = pack_bf16x2((acc[t][4 * g] - mean) * rstd * gm[0] + bt[0], (acc[t][4 * g + 1] - mean) * rstd * gm[1] + bt[1]);
            o[1] = pack_bf16x2((acc[t][4 * g + 2] - mean) * rstd * gm[2] + bt[2], (acc[t][4 * g + 3] - mean) * rstd * gm[3] + bt[3]);
            *(u32x2*)(stage + r * 256 + (((t * 4 + g) ^ (r & 15)) << 4) + 8 * h) = o;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // wave-private tile
#pragma unroll
    for (int it = 0; it < 8; ++it) {                                // 4 token rows (256 B each) per wave instruction
        const int row = it * 4 + (lane >> 4), q = lane & 15;
        const long pg = p0 + wave * 32 + row;
        const u32x4 v = *(const u32x4*)(stage + row * 256 + ((q ^ (row & 15)) << 4));
        if (pg < npatch) *(u32x4*)(p.out + (size_t)pg * E + q * 8) = v;
    }
}

}  // namespace

int patch_embed_bf16(const float* img, const float* w, const float* bias, const float* gamma, const float* beta, bf16_t* out,
                     int B, int H, int W, float eps, hipStream_t s) {
    PeArgs p{img, w, bias, gamma, beta, out, B, H, W, eps};
    const long npatch = (long)B * (H / 4) * (W / 4);
    hipLaunchKernelGGL(patch_embed_bf16_kernel, dim3((unsigned)((npatch + 127) / 128)), dim3(256), 0, s, p);
    return master_check_launch("patch_embed_bf16");
}
