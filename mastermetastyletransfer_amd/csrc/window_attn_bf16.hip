// Windowed attention core, bf16 throughput mode: one wave per (image, window, head).
//
// S^T = K Q^T on v_mfma_f32_32x32x16_bf16 (query on the lane, keys in the accumulator registers, as
// in the fp32 kernel), softmax in fp32 in registers, and for O^T = V^T P^T the accumulator tile is the
// B operand again: registers 8s..8s+7 of a 32x32 tile, converted pairwise to bf16, are the fragment of
// k-step s, whose k order is  key = 16s + 8(j>>2) + 4*half + (j&3).  The matching A fragment
// V^T[d][those keys] is produced by the hardware transpose read ds_read_b64_tr_b16 from a row-major
// [64 keys][32 ch] LDS image of V (filled with coalesced 16-B gathers): two reads of 4 keys x 16
// channels per k-step.  64-B rows make every transposed read conflict-free (a half-wave covers
// 4 rows x 64 B = one contiguous 256-B bank row).  Each wave owns a private LDS slice; no block barrier.
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct AttnArgsB {
    const bf16_t* q; const bf16_t* k; const bf16_t* v0; const bf16_t* v1;
    int ldq, ldk, ldv;
    const bf16_t* k_pad; const bf16_t* v0_pad; const bf16_t* v1_pad;
    int k_pad_bs, v_pad_bs;
    const float* bias; const int32_t* src; const uint8_t* region;
    bf16_t* out0; bf16_t* out1; int ldo;
    int B, T, heads, nW, N;
    float scale_log2e;
};

constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ unsigned pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }

union Frag { u32x4 u; bf16x8_t b; };

template <bool DUAL>
__global__ __launch_bounds__(256) void window_attn_bf16_kernel(AttnArgsB p) {
    __shared__ __attribute__((aligned(16))) bf16_t vlds[4][2][64 * 32];     // [wave][stream][key][ch]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int pid = blockIdx.x * 4 + wave;
    const int total = p.B * p.nW * p.heads;
    if (pid >= total) return;                           // wave-uniform
    const int head = pid % p.heads;
    const int w = (pid / p.heads) % p.nW;
    const int b = pid / (p.heads * p.nW);
    const int32_t* src = p.src + w * 64;
    const size_t tok0 = (size_t)b * p.T;
    const int coff = head * 32;

    // ---- stage V (and V1) row-major into this wave's LDS slice: 256 chunks of 16 B, 4 per lane ------
    auto stage_v = [&](const bf16_t* v, const bf16_t* v_pad, bf16_t* dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + 64 * i, key = c >> 2, part = c & 3;
            u32x4 val = {0u, 0u, 0u, 0u};
            if (key < p.N) {
                const int idx = src[key];
                const bf16_t* vp = idx >= 0 ? v + (tok0 + idx) * p.ldv + coff + part * 8
                                            : v_pad + (size_t)b * p.v_pad_bs + coff + part * 8;
                val = *(const u32x4*)vp;
            }
            *(u32x4*)(dst + key * 32 + part * 8) = val;
        }
    };
    stage_v(p.v0, p.v0_pad, vlds[wave][0]);
    if (DUAL) stage_v(p.v1, p.v1_pad, vlds[wave][1]);

    // ---- Q and K fragments: lane (row r, half h) holds channels [16h, 16h+16); k-step s = [16h+8s, +8)
    Frag qf[2][2], kf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int slot = t * 32 + r;
        const int idx = slot < p.N ? src[slot] : -1;
        const bf16_t* qp = p.q + (tok0 + (idx >= 0 ? idx : 0)) * p.ldq + coff + 16 * h;
        const bf16_t* kp = idx >= 0 ? p.k + (tok0 + idx) * p.ldk + coff + 16 * h
                                    : p.k_pad + (size_t)b * p.k_pad_bs + coff + 16 * h;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            qf[t][s].u = *(const u32x4*)(qp + 8 * s);
            kf[t][s].u = *(const u32x4*)(kp + 8 * s);
        }
    }

    f32x16 st[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) st[kt][qt][e] = 0.0f;
#pragma unroll
            for (int s = 0; s < 2; ++s)
                st[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][s].b, qf[qt][s].b, st[kt][qt], 0, 0, 0);
        }

    // ---- S' = log2e * (scale * qk + bias + mask) ------------------------------------------------------
    const float* bias = p.bias + (size_t)head * 64 * 64;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float* brow = bias + (qt * 32 + r) * 64;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *(const f32x4*)(brow + kt * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    st[kt][qt][4 * g + e] = st[kt][qt][4 * g + e] * p.scale_log2e + bv[e] * LOG2E;
            }
    }
    if (p.region) {
        const uint8_t* reg = p.region + w * 64;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const unsigned rq = reg[qt * 32 + r];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned rk4 = *(const unsigned*)(reg + kt * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (((rk4 >> (8 * e)) & 0xffu) != rq) st[kt][qt][4 * g + e] += -100.0f * LOG2E;
                }
        }
    }

    // ---- softmax over keys (32 in-lane + partner half), probabilities packed to bf16 fragments -------
    float inv[2];
    Frag pf[2][2][2];                                  // [qt][kt][s]
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float m = st[0][qt][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) m = fmaxf(m, st[kt][qt][e]);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.0f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            // round to bf16 first, then sum the rounded values: the normaliser matches what the MFMA multiplies
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const unsigned pk = pack_bf16x2(__builtin_amdgcn_exp2f(st[kt][qt][8 * s + 2 * jj] - m), __builtin_amdgcn_exp2f(st[kt][qt][8 * s + 2 * jj + 1] - m));
                    pf[qt][kt][s].u[jj] = pk;
                    sum += __uint_as_float(pk << 16) + __uint_as_float(pk & 0xffff0000u);
                }
        }
        sum += __shfl_xor(sum, 32, 64);
        inv[qt] = 1.0f / sum;
    }

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's V image is in LDS (DS ops are in order per wave)

    // ---- O^T[d][q] = sum_key V^T[d][key] P^T[key][q] ---------------------------------------------------
    const int g16 = (lane >> 4) & 1, i16 = lane & 15;
    auto pv = [&](const bf16_t* vimg, bf16_t* out) {
        f32x16 o[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[qt][e] = 0.0f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                // element j of the A fragment = V[32kt + 16s + 8(j>>2) + 4h + (j&3)][d]; one transposed read
                // per j>>2: lane 4q+p of each 16-lane group addresses row q, channels 4p..4p+3 of the block
                union { s16x4 v[2]; bf16x8_t b; } vf;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int key0 = 32 * kt + 16 * s + 8 * jj + 4 * h;
                    const bf16_t* ap = vimg + (key0 + (i16 >> 2)) * 32 + 16 * g16 + 4 * (i16 & 3);
                    vf.v[jj] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)ap);
                }
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    o[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.b, pf[qt][kt][s].b, o[qt], 0, 0, 0);
            }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int slot = qt * 32 + r;
            const int idx = slot < p.N ? src[slot] : -1;
            if (idx < 0) continue;
            bf16_t* op = out + (tok0 + idx) * p.ldo + coff + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 ov;
                ov[0] = pack2(o[qt][4 * g] * inv[qt], o[qt][4 * g + 1] * inv[qt]);
                ov[1] = pack2(o[qt][4 * g + 2] * inv[qt], o[qt][4 * g + 3] * inv[qt]);
                *(u32x2*)(op + 8 * g) = ov;
            }
        }
    };
    pv(vlds[wave][0], p.out0);
    if (DUAL) pv(vlds[wave][1], p.out1);
}

}  // namespace

int window_attn_bf16(const bf16_t* q, const bf16_t* k, const bf16_t* v0, const bf16_t* v1, int ldq, int ldk, int ldv,
                     const bf16_t* k_pad, const bf16_t* v0_pad, const bf16_t* v1_pad, int k_pad_bs, int v_pad_bs,
                     const float* bias, const int32_t* src, const uint8_t* region, bf16_t* out0, bf16_t* out1, int ldo,
                     int B, int T, int heads, int nW, int N, float scale, hipStream_t s) {
    AttnArgsB p{q, k, v0, v1, ldq, ldk, ldv, k_pad, v0_pad, v1_pad, k_pad_bs, v_pad_bs, bias, src, region,
                out0, out1, ldo, B, T, heads, nW, N, scale * LOG2E};
    const int total = B * nW * heads;
    const int grid = (total + 3) / 4;
    if (v1) hipLaunchKernelGGL(window_attn_bf16_kernel<true>, dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(window_attn_bf16_kernel<false>, dim3(grid), dim3(256), 0, s, p);
    return master_check_launch("window_attn_bf16");
}
