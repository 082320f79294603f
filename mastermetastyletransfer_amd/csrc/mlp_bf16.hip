// Fused MLP block, bf16 throughput mode:   out = res + W2 . gelu(W1 . LN?(x) + b1) + b2
// (torchvision MLP = Linear, GELU, Linear and the residual add around it; optional LayerNorm on x:
// reference style_transformer.py:366,:392,:839-841,:865-882,:991,:1125 and the Swin blocks).
//
// One workgroup = 128 tokens, 4 waves = 4 SIMDs, one wave per SIMD with the whole register file;
// wave w owns tokens [32w, 32w+32) for the entire kernel.  Everything is computed TRANSPOSED so that the
// token sits on the MFMA lane (C/D column) and channels sit in accumulator registers:
//   phase 1   H^T[j][m] = sum_k W1[j][k] X[m][k]      A = W1 rows (LDS), B = X fragments (registers, loaded
//                                                     once; LayerNorm applied in registers)
//   phase 2   O^T[n][m] += sum_j W2[n][j] H^T[j][m]   A = W2 rows (LDS), B = the phase-1 accumulator tile
//                                                     itself (bias+GELU, packed to bf16 in registers)
// so the hidden activations never touch LDS or HBM: a 32x32 accumulator tile is a valid B operand whose
// k order is j = 16s + 8(e>>2) + 4*half + (e&3); the W2 fragment is read in that order (two 8-byte LDS
// reads).  Weights stream through a 2-slot LDS ring in 32-KB sub-chunks (global -> registers -> LDS, next
// sub-chunk's loads in flight under 32 MFMAs per wave, one barrier per sub-chunk):
//   W1 sub-chunk [128 j][128 k], rows padded to 272 B          -> conflict-free ds_read_b128
//   W2 sub-chunk [C n][64 j], 128-B rows, 8-B units XOR-swizzled by (n>>1)&15 -> conflict-free ds_read_b64
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct MlpArgs {
    const bf16_t* x; const float* ln_g; const float* ln_b;
    const bf16_t* w1; const float* b1; const bf16_t* w2; const float* b2;
    const bf16_t* res; bf16_t* out;
    int M, HID, ldx, ldr, ldo; float eps;
};

union Frag { u32x4 u; bf16x8_t b; u32x2 h2[2]; };

constexpr int SLOT_BYTES = 128 * 272;     // >= C*128 for C <= 256
constexpr int W1_STRIDE = 272;            // bytes per padded W1 row (128 k)

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) { return pack_bf16x2(lo, hi); }

template <int C>
__global__ __launch_bounds__(256, 1) void mlp_bf16_kernel(MlpArgs p) {
    constexpr int KS1 = C / 16;            // k-steps of phase 1 (all of K1 = C)
    constexpr int KB = C / 128;            // W1 sub-chunks per hidden chunk
    constexpr int NT2 = C / 32;            // output-channel tiles of phase 2
    constexpr int W1_CH = 8;               // 16-B chunks per thread for a W1 sub-chunk (128 rows x 256 B)
    constexpr int W2_CH = C * 128 / 16 / 256;   // 16-B chunks per thread for a W2 sub-chunk (C rows x 128 B)
    constexpr int ST_CH = W1_CH > W2_CH ? W1_CH : W2_CH;

    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];   // [2][SLOT_BYTES]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m = blockIdx.x * 128 + wave * 32 + r;
    const int mc = m < p.M ? m : p.M - 1;
    const int nhc = p.HID / 128;

    // ---- X fragments (B operand of phase 1): lane (token m, half h) holds x[m][16s + 8h .. +8] ---------
    Frag xf[KS1];
    {
        const bf16_t* xp = p.x + (size_t)mc * p.ldx + 8 * h;
#pragma unroll
        for (int s = 0; s < KS1; ++s) xf[s].u = *(const u32x4*)(xp + 16 * s);
        if (p.ln_g) {
            // LayerNorm over the C channels of token m: this lane holds half of them, lane^32 the rest
            float v[KS1][8];
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < KS1; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[s][2 * e] = __uint_as_float(xf[s].u[e] << 16);
                    v[s][2 * e + 1] = __uint_as_float(xf[s].u[e] & 0xffff0000u);
                    sum += v[s][2 * e] + v[s][2 * e + 1];
                }
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum / (float)C;
            float sq = 0.f;
#pragma unroll
            for (int s = 0; s < KS1; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[s][e] - mean; sq += d * d; }
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = rsqrtf(sq / (float)C + p.eps);
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                const float* gp = p.ln_g + 16 * s + 8 * h; const float* bp = p.ln_b + 16 * s + 8 * h;
                const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 4);
                const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
                float o[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (v[s][e] - mean) * rstd * g0[e] + b0[e];
                    o[4 + e] = (v[s][4 + e] - mean) * rstd * g1[e] + b1[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) xf[s].u[e] = pack_bf16(o[2 * e], o[2 * e + 1]);
            }
        }
    }

    // ---- weight streaming ------------------------------------------------------------------------------
    u32x4 st[ST_CH];
    auto load_w1 = [&](int hc, int kb) {          // rows j = hc*128 + c>>4, 16-B column c&15 of the 256-B k slice
#pragma unroll
        for (int i = 0; i < W1_CH; ++i) {
            const int c = tid + 256 * i;
            st[i] = *(const u32x4*)(p.w1 + (size_t)(hc * 128 + (c >> 4)) * C + kb * 128 + (c & 15) * 8);
        }
    };
    auto store_w1 = [&](int slot) {
#pragma unroll
        for (int i = 0; i < W1_CH; ++i) {
            const int c = tid + 256 * i;
            *(u32x4*)(ring + slot * SLOT_BYTES + (c >> 4) * W1_STRIDE + (c & 15) * 16) = st[i];
        }
    };
    auto load_w2 = [&](int hc, int sb) {          // rows n = c>>3, 16-B column c&7 of the 128-B j slice
#pragma unroll
        for (int i = 0; i < W2_CH; ++i) {
            const int c = tid + 256 * i;
            st[i] = *(const u32x4*)(p.w2 + (size_t)(c >> 3) * p.HID + hc * 128 + sb * 64 + (c & 7) * 8);
        }
    };
    auto store_w2 = [&](int slot) {               // 8-B units swizzled: unit' = unit ^ ((n>>1)&15)
#pragma unroll
        for (int i = 0; i < W2_CH; ++i) {
            const int c = tid + 256 * i;
            const int n = c >> 3, u = (c & 7) * 2, f = (n >> 1) & 15;
            unsigned char* rowp = ring + slot * SLOT_BYTES + n * 128;
            u32x2 lo, hi; lo[0] = st[i][0]; lo[1] = st[i][1]; hi[0] = st[i][2]; hi[1] = st[i][3];
            *(u32x2*)(rowp + ((u ^ f) * 8)) = lo;
            *(u32x2*)(rowp + (((u + 1) ^ f) * 8)) = hi;
        }
    };

    f32x16 oacc[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[t][e] = 0.f;

    // phase-2 LDS byte offsets of this lane's two 8-B pieces per (jl, s): the swizzle term f = (n>>1)&15
    // depends only on the row within a 32-row tile, so tile nt just adds nt*32*128 (an immediate)
    int off0[2][2], off1[2][2];
#pragma unroll
    for (int jl = 0; jl < 2; ++jl)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int u0 = jl * 8 + 4 * s + h, f = (r >> 1) & 15;
            off0[jl][s] = r * 128 + ((u0 ^ f) * 8);
            off1[jl][s] = r * 128 + (((u0 + 2) ^ f) * 8);
        }

    int slot = 0;
    load_w1(0, 0);
    store_w1(0);
    __syncthreads();

    for (int hc = 0; hc < nhc; ++hc) {
        // ================= phase 1: H^T chunk [128 j][32 m] =================
        f32x16 hacc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) hacc[t][e] = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const bool next_is_w1 = kb + 1 < KB;
            if (next_is_w1) load_w1(hc, kb + 1); else load_w2(hc, 0);
            const unsigned char* base = ring + slot * SLOT_BYTES + r * W1_STRIDE + 16 * h;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                Frag a[4];
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) a[jt].u = *(const u32x4*)(base + jt * 32 * W1_STRIDE + 32 * s);
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
                    hacc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[jt].b, xf[kb * 8 + s].b, hacc[jt], 0, 0, 0);
            }
            if (next_is_w1) store_w1(slot ^ 1); else store_w2(slot ^ 1);
            __syncthreads();
            slot ^= 1;
        }
        // bias + GELU, packed in place as the B fragments of phase 2: tile jt, k-step s <- registers 8s..8s+7
        Frag hf[4][2];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *(const f32x4*)(p.b1 + hc * 128 + jt * 32 + 8 * g + 4 * h);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(hacc[jt][4 * g + e] + bv[e]);
                hf[jt][g >> 1].u[(g & 1) * 2] = pack_bf16(v[0], v[1]);
                hf[jt][g >> 1].u[(g & 1) * 2 + 1] = pack_bf16(v[2], v[3]);
            }
        // ================= phase 2: O^T[C n][32 m] += W2[:, chunk] H^T =================
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            const bool last = (sb == 1) && (hc + 1 == nhc);
            if (sb == 0) load_w2(hc, 1); else if (!last) load_w1(hc + 1, 0);
            const unsigned char* base = ring + slot * SLOT_BYTES;
#pragma unroll
            for (int jl = 0; jl < 2; ++jl)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
#pragma unroll
                    for (int nt = 0; nt < NT2; ++nt) {
                        Frag a;
                        a.h2[0] = *(const u32x2*)(base + nt * 4096 + off0[jl][s]);
                        a.h2[1] = *(const u32x2*)(base + nt * 4096 + off1[jl][s]);
                        oacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b, hf[sb * 2 + jl][s].b, oacc[nt], 0, 0, 0);
                    }
                }
            if (sb == 0) store_w2(slot ^ 1); else if (!last) store_w1(slot ^ 1);
            __syncthreads();
            slot ^= 1;
        }
    }

    // ---- epilogue: O^T tile: lane = token m, registers = channels n = nt*32 + (e&3) + 8(e>>2) + 4h -----
    if (m < p.M) {
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = nt * 32 + 8 * g + 4 * h;
                const f32x4 bv = *(const f32x4*)(p.b2 + n);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = oacc[nt][4 * g + e] + bv[e];
                if (p.res) {
                    const u32x2 rr = *(const u32x2*)(p.res + (size_t)m * p.ldr + n);
                    v[0] += __uint_as_float(rr[0] << 16); v[1] += __uint_as_float(rr[0] & 0xffff0000u);
                    v[2] += __uint_as_float(rr[1] << 16); v[3] += __uint_as_float(rr[1] & 0xffff0000u);
                }
                u32x2 o; o[0] = pack_bf16(v[0], v[1]); o[1] = pack_bf16(v[2], v[3]);
                *(u32x2*)(p.out + (size_t)m * p.ldo + n) = o;
            }
    }
}

template <int C>
int launch(const MlpArgs& p, hipStream_t s) {
    const size_t lds = 2 * SLOT_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)mlp_bf16_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((mlp_bf16_kernel<C>), dim3((p.M + 127) / 128), dim3(256), lds, s, p);
    return master_check_launch("mlp_bf16");
}

}  // namespace

int mlp_bf16(const bf16_t* x, const float* ln_g, const float* ln_b, const bf16_t* w1, const float* b1, const bf16_t* w2,
             const float* b2, const bf16_t* res, bf16_t* out, int M, int C, int HID, int ldx, int ldr, int ldo, float eps,
             hipStream_t s) {
    MlpArgs p{x, ln_g, ln_b, w1, b1, w2, b2, res, out, M, HID, ldx, ldr, ldo, eps};
    if (HID % 128 != 0) { master_set_error("mlp: hidden width must be a multiple of 128 (got %d)", HID); return MASTER_ERR_INVALID; }
    if (C == 256) return launch<256>(p, s);
    if (C == 128) return launch<128>(p, s);
    master_set_error("mlp(bf16): channel width %d not built (128, 256)", C);
    return MASTER_ERR_UNSUPPORTED;
}
