// bf16 GEMM family (throughput mode): v_mfma_f32_32x32x16_bf16, fp32 accumulate, bf16 storage.
//
//   out[M,N] = r1 (.) act(A[M,K] . W[N,K]^T + bias) + r2          (bias fp32; A, W, r1, r2, out bf16)
//
// Same byte geometry as the fp32 kernel: a BK = 64 chunk is a 128-B row segment, staged
// global -> registers -> LDS rows padded to 144 B (conflict-free ds_read_b128), one barrier per chunk,
// next chunk's global loads in flight under the MFMAs.  Lane-half h reads bytes [64h, 64h+64) of its
// row; k-step s (of 4) uses the 16-B fragment 64h+16s of A and of W alike, so the k permutation
// cancels.  Epilogue: bias + activation in registers, tile staged through LDS in fp32, then every
// thread handles 16-B row pieces: r1/r2 loads and bf16 stores are whole 128/256-B row segments.
// CONV: implicit GEMM of the 3x3 reflect-pad (+nearest x2 upsample) decoder convolutions; K is padded
// to a multiple of 64 (tail weights read as zero) when Cin = 32, where a chunk spans two taps.
#include "common.h"

namespace {

constexpr int BK = 64;     // bf16 elements per chunk (128 B)
constexpr int LDK = 72;    // padded LDS row stride in bf16 elements (144 B)

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct GemmArgsB {
    const bf16_t* a; const bf16_t* w; const float* bias; const bf16_t* r1; const bf16_t* r2; void* out;
    int M, N, K, lda, ldo, ldr, act;
    int Hin, Win, Cin, Hout, Wout, up, out_nchw;
    int zero_pad;      // conv gather: out-of-image taps read 0 (nn.Conv2d padding=1) instead of the mirror pixel (padding_mode='reflect')
    int Kw;            // real length (and stride) of a weight row; K - Kw trailing columns are zero padding
    int mask_relu;     // r1 is a ReLU output that gates the result: out = r1 > 0 ? acc + r2 : 0  (data gradient chains of the loss network)
};

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__device__ __forceinline__ unsigned pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }

template <int BM, int BN, int WM, int WN, bool CONV>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void gemm_bf16_kernel(GemmArgsB p) {
    constexpr int NWN = BN / WN;
    constexpr int NT = (BM / WM) * NWN * 64;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int ACH = BM * 8 / NT, BCH = (BN * 8 + NT - 1) / NT;
    constexpr int LDO = BN + 4;             // fp32 staging stride of the epilogue tile
    static_assert(BM * 8 % NT == 0, "A chunks must divide evenly");
    static_assert(BM * LDO * 4 <= 2 * (BM + BN) * LDK * 2, "epilogue tile must fit in the staging buffers");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* As = (bf16_t*)smem_raw;                   // [2][BM][LDK]
    bf16_t* Ws = As + 2 * BM * LDK;                   // [2][BN][LDK]
    float* Os = (float*)smem_raw;                     // [BM][LDO] after the K loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / NWN) * WM, wn = (wave % NWN) * WN;
    const int r = lane & 31, h = lane >> 5;

    const int MT = (p.M + BM - 1) / BM, NTN = (p.N + BN - 1) / BN;
    const int id = blockIdx.x;
    const int j = id >> 3;
    const int mt = (j / NTN) * 8 + (id & 7), nt = j % NTN;     // XCD-aware: N-tiles of an M-tile share id%8
    if (mt >= MT) return;
    const int m0 = mt * BM, n0 = nt * BN;

    const int kc = (tid & 7) * 8;
    const bf16_t* a_ptr[ACH];
    int cy[ACH], cx[ACH], cb[ACH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        int m = m0 + (tid >> 3) + i * (NT / 8);
        m = m < p.M ? m : p.M - 1;
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            cb[i] = m / hw; const int pix = m - cb[i] * hw; cy[i] = pix / p.Wout; cx[i] = pix - cy[i] * p.Wout;
            a_ptr[i] = p.a;
        } else {
            a_ptr[i] = p.a + (size_t)m * p.lda + kc;
        }
    }
    const bf16_t* w_ptr[BCH];
    bool w_on[BCH];
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        const int row = (tid >> 3) + i * (NT / 8);
        w_on[i] = row < BN;
        int n = n0 + row; n = n < p.N ? n : p.N - 1;
        w_ptr[i] = p.w + (size_t)n * p.Kw + kc;
    }

    u32x4 a_st[ACH], w_st[BCH];
    auto stage_load = [&](int kt) {
        const int k0 = kt * BK;
        if (CONV) {
            const int kk = k0 + kc;
            int tap = kk / p.Cin; const int c = kk - tap * p.Cin;
            tap = tap < 8 ? tap : 8;                       // K padding (zero weights): stay in bounds
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                const int yy = cy[i] + dy, xx = cx[i] + dx;
                const int sy = reflect1(yy, p.Hout) >> p.up;
                const int sx = reflect1(xx, p.Wout) >> p.up;
                a_st[i] = *(const u32x4*)(p.a + ((size_t)(cb[i] * p.Hin + sy) * p.Win + sx) * p.Cin + c);
                if (p.zero_pad && ((unsigned)yy >= (unsigned)p.Hout || (unsigned)xx >= (unsigned)p.Wout)) a_st[i] = u32x4{0u, 0u, 0u, 0u};
            }
        } else {
#pragma unroll
            for (int i = 0; i < ACH; ++i) a_st[i] = *(const u32x4*)(a_ptr[i] + k0);
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) if (w_on[i])
            w_st[i] = (k0 + kc < p.Kw) ? *(const u32x4*)(w_ptr[i] + k0) : u32x4{0u, 0u, 0u, 0u};
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ACH; ++i)
            *(u32x4*)(As + (buf * BM + (tid >> 3) + i * (NT / 8)) * LDK + kc) = a_st[i];
#pragma unroll
        for (int i = 0; i < BCH; ++i) if (w_on[i])
            *(u32x4*)(Ws + (buf * BN + (tid >> 3) + i * (NT / 8)) * LDK + kc) = w_st[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.0f;

    const int nk = p.K / BK;
    stage_load(0);
    stage_store(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage_load(kt + 1);
        const bf16_t* ab = As + (buf * BM + wm + r) * LDK + 32 * h;
        const bf16_t* wb = Ws + (buf * BN + wn + r) * LDK + 32 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8_t av[TM], wv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *(const bf16x8_t*)(ab + i * 32 * LDK + 8 * s);
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) wv[jn] = *(const bf16x8_t*)(wb + jn * 32 * LDK + 8 * s);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], wv[jn], acc[i][jn], 0, 0, 0);
        }
        if (kt + 1 < nk) stage_store(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue 1: bias + activation, fp32 tile -> LDS (C/D map: col = lane&31, row = (e&3)+8*(e>>2)+4h)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int nl = wn + jn * 32 + r;
        const int n = n0 + nl;
        const float bv = (p.bias && n < p.N) ? p.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[i][jn][e] + bv;
                if (p.act == MASTER_ACT_GELU) v = gelu_erf(v);
                else if (p.act == MASTER_ACT_RELU) v = fmaxf(v, 0.0f);
                Os[(wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * LDO + nl] = v;
            }
    }
    __syncthreads();

    // ---- epilogue 2: row pieces of 4 columns per thread, coalesced r1/r2 loads and stores ----------
    if (CONV && p.out_nchw) {
        // the model output: fp32 NCHW, N = 3 channels; consecutive threads -> consecutive pixels
        float* out = (float*)p.out;
        const int hw = p.Hout * p.Wout;
        for (int ml = tid; ml < BM; ml += NT) {
            const int m = m0 + ml;
            if (m >= p.M) continue;
            const int b = m / hw, pix = m - b * hw;
            for (int nl = 0; nl < BN && n0 + nl < p.N; ++nl)
                out[((size_t)b * p.N + n0 + nl) * hw + pix] = Os[ml * LDO + nl];
        }
        return;
    }
    bf16_t* out = (bf16_t*)p.out;
    constexpr int PPR = BN / 4;                          // 4-column pieces per tile row
    const bool vec_ok = (p.ldo % 4 == 0) && (p.N % 4 == 0) && (p.ldr % 4 == 0);
    for (int pc = tid; pc < BM * PPR; pc += NT) {
        const int ml = pc / PPR, nl = (pc % PPR) * 4;
        const int m = m0 + ml, n = n0 + nl;
        if (m >= p.M || n >= p.N) continue;
        f32x4 v = *(const f32x4*)(Os + ml * LDO + nl);
        if (vec_ok) {
            if (p.r1 && !p.mask_relu) {
                const u32x2 u = *(const u32x2*)(p.r1 + (size_t)m * p.ldr + n);
                v[0] *= __uint_as_float(u[0] << 16); v[1] *= __uint_as_float(u[0] & 0xffff0000u);
                v[2] *= __uint_as_float(u[1] << 16); v[3] *= __uint_as_float(u[1] & 0xffff0000u);
            }
            if (p.r2) {
                const u32x2 u = *(const u32x2*)(p.r2 + (size_t)m * p.ldr + n);
                v[0] += __uint_as_float(u[0] << 16); v[1] += __uint_as_float(u[0] & 0xffff0000u);
                v[2] += __uint_as_float(u[1] << 16); v[3] += __uint_as_float(u[1] & 0xffff0000u);
            }
            if (p.mask_relu) {
                const u32x2 u = *(const u32x2*)(p.r1 + (size_t)m * p.ldr + n);
                if (!(__uint_as_float(u[0] << 16) > 0.f)) v[0] = 0.f;
                if (!(__uint_as_float(u[0] & 0xffff0000u) > 0.f)) v[1] = 0.f;
                if (!(__uint_as_float(u[1] << 16) > 0.f)) v[2] = 0.f;
                if (!(__uint_as_float(u[1] & 0xffff0000u) > 0.f)) v[3] = 0.f;
            }
            u32x2 o; o[0] = pack2(v[0], v[1]); o[1] = pack2(v[2], v[3]);
            *(u32x2*)(out + (size_t)m * p.ldo + n) = o;
        } else {
            for (int e = 0; e < 4 && n + e < p.N; ++e) {
                float x = v[e];
                if (p.r1 && !p.mask_relu) x *= bf16_to_f32(p.r1[(size_t)m * p.ldr + n + e]);
                if (p.r2) x += bf16_to_f32(p.r2[(size_t)m * p.ldr + n + e]);
                if (p.mask_relu && !(bf16_to_f32(p.r1[(size_t)m * p.ldr + n + e]) > 0.f)) x = 0.f;
                out[(size_t)m * p.ldo + n + e] = f32_to_bf16(x);
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, bool CONV>
int launch(const GemmArgsB& p, hipStream_t s) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    const int MT = (p.M + BM - 1) / BM, NTN = (p.N + BN - 1) / BN;
    const int grid = ((MT + 7) / 8) * 8 * NTN;
    const size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(bf16_t);
    static LdsLimit lds_limit;
    lds_limit.ensure((const void*)gemm_bf16_kernel<BM, BN, WM, WN, CONV>, (int)((int)lds));
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, WM, WN, CONV>), dim3(grid), dim3(NT), lds, s, p);
    return master_check_launch("gemm_bf16");
}

template <bool CONV>
int dispatch(const GemmArgsB& p, hipStream_t s) {
    if (p.N > 64) return launch<128, 128, 64, 64, CONV>(p, s);
    if (p.N > 32) return launch<128, 64, 64, 32, CONV>(p, s);
    return launch<256, 32, 64, 32, CONV>(p, s);
}

}  // namespace

int gemm_bf16_linear(const bf16_t* a, const bf16_t* w, const float* bias, const bf16_t* r1, const bf16_t* r2,
                     bf16_t* out, int M, int N, int K, int lda, int ldo, int ldr, int act, hipStream_t s) {
    if (K % BK != 0) { master_set_error("linear(bf16): K must be a multiple of 64 (K=%d)", K); return MASTER_ERR_INVALID; }
    GemmArgsB p{a, w, bias, r1, r2, out, M, N, K, lda, ldo, ldr, act, 0, 0, 0, 1, 1, 0, 0, 0, K, 0};
    return dispatch<false>(p, s);
}

// w: [Cout][3][3][Cin]; the K loop runs to roundup(9*Cin, 64) with the tail treated as zero weights
int gemm_bf16_conv3x3(const bf16_t* x, const bf16_t* w, const float* bias, void* out, int B, int Hin, int Win, int Cin,
                      int Cout, int up, int relu, int out_nchw, hipStream_t s, int zero_pad = 0) {
    const int Hout = Hin << up, Wout = Win << up;
    const int Kpad = (9 * Cin + BK - 1) / BK * BK;
    GemmArgsB p{x, w, bias, nullptr, nullptr, out, B * Hout * Wout, Cout, Kpad, 0, Cout, 0,
                relu ? MASTER_ACT_RELU : MASTER_ACT_NONE, Hin, Win, Cin, Hout, Wout, up, out_nchw, zero_pad, 9 * Cin, 0};
    return dispatch<true>(p, s);
}

// Data gradient of a zero-padded 3x3 convolution in bf16 (the frozen loss network when it runs in bf16): the forward kernel on the
// output gradient g [B,H,W,Cg] with flipped, transposed weights wf [Cx][3][3][Cg]; out = relu_src > 0 ? acc + add : 0 (both optional).
int gemm_bf16_conv3x3_dgrad(const bf16_t* g, const bf16_t* wf, bf16_t* out, const bf16_t* relu_src, const bf16_t* add, int B, int H, int W,
                            int Cg, int Cx, hipStream_t s) {
    const int Kpad = (9 * Cg + BK - 1) / BK * BK;
    GemmArgsB p{g, wf, nullptr, relu_src, add, out, B * H * W, Cx, Kpad, 0, Cx, Cx, MASTER_ACT_NONE, H, W, Cg, H, W, 0, 0, 1, 9 * Cg,
                relu_src != nullptr};
    return dispatch<true>(p, s);
}
