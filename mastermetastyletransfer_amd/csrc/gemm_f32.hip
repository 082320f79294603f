// fp32 GEMM family on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32), parity mode.
//
//   out[M,N] = r1 (.) act(A[M,K] . W[N,K]^T + bias) + r2
//
// A is either a plain row-major token matrix (Linear layers) or an implicit-GEMM gather of a
// 3x3 reflect-padded (optionally nearest-x2-upsampled) NHWC image (decoder convolutions).
// Both operands are K-contiguous, so one staging path serves A and W:
//   global (16 B / lane, 8 lanes per 128-B row segment) -> registers -> LDS rows padded to 36 floats
//   (stride 9 x 16 B: any 16 distinct rows hit 16 distinct 16-B slots -> conflict-free ds_read_b128).
// Each wave owns a WM x WN sub-tile as (WM/32) x (WN/32) accumulators of 32x32.  For one BK=32
// chunk lane-half h reads floats [16h, 16h+16) of its row with four ds_read_b128; MFMA k-step s then
// pairs column s (half 0) with column 16+s (half 1).  A and W use the same k permutation, so the sum
// is unchanged.  One barrier per K chunk; the next chunk's global loads are issued before the MFMAs.
#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = 36;   // padded LDS row stride (floats)

struct GemmArgs {
    const float* a; const float* w; const float* bias; const float* r1; const float* r2; float* out;
    int M, N, K, lda, ldo, ldr, act;
    int Hin, Win, Cin, Hout, Wout, up, out_nchw, zero_pad;   // conv gather only (zero_pad: out-of-image taps read 0 instead of the mirror pixel)
    // training extras.  aux: the pre-activation acc + bias is also stored there (same layout as out; saved for the GELU backward).
    // off = 1 (zero_pad only): "full" correlation - the output domain is (Hsrc+2) x (Wsrc+2) and output pixel (y, x) reads source
    // pixels (y - 2 + ty, x - 2 + tx): the data-gradient of a padded 3x3 convolution on the padded domain.
    // mask_relu: r1 is not a multiplier but a ReLU output whose sign gates the result:  out = r1 > 0 ? acc + r2 : 0
    float* aux; int off, mask_relu;
};

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

template <int BM, int BN, int WM, int WN, bool CONV>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void gemm_f32_kernel(GemmArgs p) {
    constexpr int NWM = BM / WM, NWN = BN / WN;
    constexpr int NT = NWM * NWN * 64;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int ACH = BM * 8 / NT, BCH = (BN * 8 + NT - 1) / NT;
    static_assert(BM * 8 % NT == 0, "A chunks must divide evenly");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][BM][LDK]
    float* Ws = smem + 2 * BM * LDK;        // [2][BN][LDK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / NWN) * WM, wn = (wave % NWN) * WN;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware tile order: the N-tiles of one M-tile get ids that are congruent mod 8 (same XCD
    // under round-robin placement) and adjacent in time, so the A panel is fetched into one L2 once.
    const int MT = (p.M + BM - 1) / BM, NTN = (p.N + BN - 1) / BN;
    const int id = blockIdx.x;
    const int j = id >> 3;
    const int mt = (j / NTN) * 8 + (id & 7), nt = j % NTN;
    if (mt >= MT) return;
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- per-thread staging descriptors -------------------------------------------------------
    const int kc = (tid & 7) * 4;
    const float* a_ptr[ACH];
    int cy[ACH], cx[ACH], cb[ACH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        int m = m0 + (tid >> 3) + i * (NT / 8);
        m = m < p.M ? m : p.M - 1;
        if (CONV) {
            int hw = p.Hout * p.Wout;
            cb[i] = m / hw; int pix = m - cb[i] * hw; cy[i] = pix / p.Wout; cx[i] = pix - cy[i] * p.Wout;
            a_ptr[i] = p.a;
        } else {
            a_ptr[i] = p.a + (size_t)m * p.lda + kc;
        }
    }
    const float* w_ptr[BCH];
    bool w_on[BCH];
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
        int row = (tid >> 3) + i * (NT / 8);
        w_on[i] = row < BN;
        int n = n0 + row; n = n < p.N ? n : p.N - 1;
        w_ptr[i] = p.w + (size_t)n * p.K + kc;
    }

    f32x4 a_st[ACH], w_st[BCH];
    auto stage_load = [&](int kt) {
        const int k0 = kt * BK;
        if (CONV) {
            const int tap = k0 / p.Cin, c0 = k0 - tap * p.Cin + kc;
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                const int yy = cy[i] + dy - p.off, xx = cx[i] + dx - p.off;
                const int Hs = p.Hin << p.up, Ws = p.Win << p.up;            // source domain (= output domain unless off)
                int sy = reflect1(yy, Hs) >> p.up;
                int sx = reflect1(xx, Ws) >> p.up;
                if (p.zero_pad) {                                            // clamp: the value is discarded below
                    sy = sy < 0 ? 0 : (sy >= p.Hin ? p.Hin - 1 : sy);
                    sx = sx < 0 ? 0 : (sx >= p.Win ? p.Win - 1 : sx);
                }
                a_st[i] = *(const f32x4*)(p.a + ((size_t)(cb[i] * p.Hin + sy) * p.Win + sx) * p.Cin + c0);
                if (p.zero_pad && ((unsigned)yy >= (unsigned)Hs || (unsigned)xx >= (unsigned)Ws)) a_st[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        } else {
#pragma unroll
            for (int i = 0; i < ACH; ++i) a_st[i] = *(const f32x4*)(a_ptr[i] + k0);
        }
#pragma unroll
        for (int i = 0; i < BCH; ++i) if (w_on[i]) w_st[i] = *(const f32x4*)(w_ptr[i] + k0);
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ACH; ++i)
            *(f32x4*)(As + (buf * BM + (tid >> 3) + i * (NT / 8)) * LDK + kc) = a_st[i];
#pragma unroll
        for (int i = 0; i < BCH; ++i) if (w_on[i])
            *(f32x4*)(Ws + (buf * BN + (tid >> 3) + i * (NT / 8)) * LDK + kc) = w_st[i];
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
        const float* ab = As + (buf * BM + wm + r) * LDK + 16 * h;
        const float* wb = Ws + (buf * BN + wn + r) * LDK + 16 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 av[TM], wv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *(const f32x4*)(ab + i * 32 * LDK + 4 * q);
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) wv[jn] = *(const f32x4*)(wb + jn * 32 * LDK + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], wv[jn][e], acc[i][jn], 0, 0, 0);
        }
        if (kt + 1 < nk) stage_store(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D map col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) -------------
    const int hw_out = p.Hout * p.Wout;
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int n = n0 + wn + jn * 32 + r;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m >= p.M) continue;
                float v = acc[i][jn][e] + bv;
                if (p.aux) p.aux[(size_t)m * p.ldo + n] = v;
                if (p.act == MASTER_ACT_GELU) v = gelu_erf(v);
                else if (p.act == MASTER_ACT_RELU) v = fmaxf(v, 0.0f);
                if (p.mask_relu) {
                    if (p.r2) v += p.r2[(size_t)m * p.ldr + n];
                    if (!(p.r1[(size_t)m * p.ldr + n] > 0.0f)) v = 0.0f;
                } else {
                    if (p.r1) v *= p.r1[(size_t)m * p.ldr + n];
                    if (p.r2) v += p.r2[(size_t)m * p.ldr + n];
                }
                if (CONV && p.out_nchw) {
                    const int b = m / hw_out, pix = m - b * hw_out;
                    p.out[((size_t)b * p.N + n) * hw_out + pix] = v;
                } else {
                    p.out[(size_t)m * p.ldo + n] = v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, bool CONV>
int launch(const GemmArgs& p, hipStream_t s) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    const int MT = (p.M + BM - 1) / BM, NTN = (p.N + BN - 1) / BN;
    const int grid = ((MT + 7) / 8) * 8 * NTN;
    const size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(float);
    static LdsLimit lds_limit;
    lds_limit.ensure((const void*)gemm_f32_kernel<BM, BN, WM, WN, CONV>, (int)((int)lds));
    hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WM, WN, CONV>), dim3(grid), dim3(NT), lds, s, p);
    return master_check_launch("gemm_f32");
}

template <bool CONV>
int dispatch(const GemmArgs& p, hipStream_t s) {
    // fewer than one 128x128 tile per CU (train step: 8 192-token matrices): 64-row tiles double the number of workgroups
    // (forcing 64-row tiles on the larger inference shapes was measured 5-10 % slower: profiles/r02_ops_fp32_gemm_{base,tile64}.txt)
    if (p.N > 64 && (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) < 256) return launch<64, 128, 32, 64, CONV>(p, s);
    if (p.N > 64) return launch<128, 128, 64, 64, CONV>(p, s);
    if (p.N > 32) return launch<128, 64, 64, 32, CONV>(p, s);
    return launch<256, 32, 64, 32, CONV>(p, s);
}

}  // namespace

int gemm_f32_linear(const float* a, const float* w, const float* bias, const float* r1, const float* r2, float* out,
                    int M, int N, int K, int lda, int ldo, int ldr, int act, hipStream_t s, float* aux = nullptr) {
    GemmArgs p{a, w, bias, r1, r2, out, M, N, K, lda, ldo, ldr, act, 0, 0, 0, 1, 1, 0, 0, 0, aux, 0, 0};
    return dispatch<false>(p, s);
}

int gemm_f32_conv3x3(const float* x, const float* w, const float* bias, float* out, int B, int Hin, int Win, int Cin,
                     int Cout, int up, int relu, int out_nchw, hipStream_t s, int zero_pad = 0) {
    const int Hout = Hin << up, Wout = Win << up;
    GemmArgs p{x, w, bias, nullptr, nullptr, out, B * Hout * Wout, Cout, 9 * Cin, 0, Cout, 0,
               relu ? MASTER_ACT_RELU : MASTER_ACT_NONE, Hin, Win, Cin, Hout, Wout, up, out_nchw, zero_pad, nullptr, 0, 0};
    return dispatch<true>(p, s);
}

// Data gradient of a 3x3 convolution as a zero-padded 3x3 correlation of the output gradient g [B,H,W,Cg] with the flipped,
// transposed weights wf [Cx][3][3][Cg]:  full = 0: out [B,H,W,Cx] (the convolution had zero padding: VGG);  full = 1: out
// [B,H+2,W+2,Cx] = the gradient on the padded domain (reflect-padded decoder convolutions; folded back by fold_reflect).
// Optional epilogue  out = relu_src > 0 ? acc + add : 0  (the ReLU backward of the layer below and a tap gradient joining there).
int gemm_f32_conv3x3_dgrad(const float* g, const float* wf, float* out, const float* relu_src, const float* add, int B, int H,
                           int W, int Cg, int Cx, int full, hipStream_t s) {
    const int Ho = H + 2 * full, Wo = W + 2 * full;
    GemmArgs p{g, wf, nullptr, relu_src, add, out, B * Ho * Wo, Cx, 9 * Cg, 0, Cx, Cx, MASTER_ACT_NONE, H, W, Cg, Ho, Wo, 0, 0, 1,
               nullptr, full, relu_src != nullptr};
    return dispatch<true>(p, s);
}
