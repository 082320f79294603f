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
#include <stdlib.h>

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
    // rowscale: out = rowscale[m / rs_rows] * act(acc + bias) + r2 - a residual branch under torchvision StochasticDepth(p, "row")
    // (style_transformer.py:390-397: one factor per image, rs_rows = tokens per image), r03: was a separate elementwise launch
    const float* rowscale; int rs_rows; unsigned rs_magic;      // rs_magic = ceil(2^32 / rs_rows) when M * rs_rows < 2^32: m / rs_rows = umulhi(m, magic)
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
    float tile_sc = 1.0f; bool tile_sc_ok = false;                      // per-image row scale of a tile whose rows belong to one image: fetched up front
    if (!CONV && p.rowscale) {
        const int mlast = (m0 + BM - 1 < p.M ? m0 + BM - 1 : p.M - 1);
        const int i0 = p.rs_magic ? (int)__umulhi((unsigned)m0, p.rs_magic) : m0 / p.rs_rows;
        const int i1 = p.rs_magic ? (int)__umulhi((unsigned)mlast, p.rs_magic) : mlast / p.rs_rows;
        tile_sc_ok = i0 == i1;
        tile_sc = p.rowscale[i0];
    }

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
                else if (p.act == MASTER_ACT_RELU) v = relu_np(v);
                if (p.mask_relu) {
                    if (p.r2) v += p.r2[(size_t)m * p.ldr + n];
                    if (!(p.r1[(size_t)m * p.ldr + n] > 0.0f)) v = 0.0f;
                } else {
                    if (p.r1) v *= p.r1[(size_t)m * p.ldr + n];
                    if (!CONV && p.rowscale) v *= tile_sc_ok ? tile_sc : p.rowscale[p.rs_magic ? (int)__umulhi((unsigned)m, p.rs_magic) : m / p.rs_rows];
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


// ---------------------------------------------------------------------------------------------------------------------------------
// r03: the same tile computed from an LDS ring filled by LDS-DMA (global_load_lds, 16 B per lane, 1-KB pieces = 8 rows x 128 B).
// Nothing is staged through registers any more: the old form held a chunk in 8 VGPR quads, waited for it mid-iteration
// (vmcnt(0) before its ds_write_b128s) and paid the tap / reflect arithmetic of the convolution gather on every 32-wide chunk.
//   * LDS rows are 128 B (BK = 32 floats), un-padded (a DMA piece lands contiguously); 16-B chunk position c of row R holds source
//     chunk c ^ ((R >> 1) & 7): the 16 rows of a ds_read_b128 lane group ({0-3,12-15,20-27} / {4-11,16-19,28-31}) alternate
//     between the two bank halves and their (R >> 1) & 7 are distinct, so the fragment reads stay conflict-free.
//   * NSTAGE-deep ring, one s_barrier per chunk: at step kt every wave waits for ITS pieces of chunk kt (counted vmcnt: the NSTAGE - 2
//     younger chunks stay in flight), the barrier makes the chunk visible and frees the slot read at step kt - 1, which is refilled
//     at once with chunk kt + NSTAGE - 1.
//   * convolution gather: a lane's source pixel moves only when the TAP changes (every Cin / 32 chunks), so the reflect / upsample /
//     zero-pad arithmetic runs 9 times per tile instead of 9 Cin / 32 times; out-of-image taps of a zero-padded convolution read a
//     512-float zero block.
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) float g_zero_row[512];          // never written: zero-initialised by the loader

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int BM, int BN, int WM, int WN, bool CONV, int NSTAGE>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void gemm_f32_dma_kernel(GemmArgs p) {
    constexpr int NWN = BN / WN;
    constexpr int NWAVES = (BM / WM) * NWN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int APW = BM / 8 / NWAVES, WPW = BN / 8 / NWAVES;          // 1-KB pieces per wave and chunk
    constexpr int PPW = APW + WPW;
    constexpr int STAGE = (BM + BN) * 128;                                // bytes per ring slot: A rows, then W rows
    static_assert((BM / 8) % NWAVES == 0 && (BN / 8) % NWAVES == 0, "pieces must divide over the waves");
    static_assert(NSTAGE == 2 || NSTAGE == 3, "ring depth");

    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / NWN) * WM, wn = (wave % NWN) * WN;
    const int r = lane & 31, h = lane >> 5;

    const int MT = (p.M + BM - 1) / BM, NTN = (p.N + BN - 1) / BN;
    const int id = blockIdx.x;
    const int j8 = id >> 3;
    const int mt = (j8 / NTN) * 8 + (id & 7), nt = j8 % NTN;           // XCD-aware order, as above
    if (mt >= MT) return;
    const int m0 = mt * BM, n0 = nt * BN;
    float tile_sc = 1.0f; bool tile_sc_ok = false;                      // per-image row scale of a tile whose rows belong to one image: fetched up front
    if (!CONV && p.rowscale) {
        const int mlast = (m0 + BM - 1 < p.M ? m0 + BM - 1 : p.M - 1);
        const int i0 = p.rs_magic ? (int)__umulhi((unsigned)m0, p.rs_magic) : m0 / p.rs_rows;
        const int i1 = p.rs_magic ? (int)__umulhi((unsigned)mlast, p.rs_magic) : mlast / p.rs_rows;
        tile_sc_ok = i0 == i1;
        tile_sc = p.rowscale[i0];
    }

    // ---- per-lane DMA sources: piece j of this wave = tile rows (wave * XPW + j) * 8 + (lane >> 3), LDS chunk position lane & 7 ------
    const int prow = lane >> 3, ppos = lane & 7;
    const float* a_src[APW];
    int cy[CONV ? APW : 1], cx[CONV ? APW : 1], cimg[CONV ? APW : 1], csw[CONV ? APW : 1];
#pragma unroll
    for (int j = 0; j < APW; ++j) {
        const int row = (wave * APW + j) * 8 + prow;
        const int sw = (ppos ^ ((row >> 1) & 7)) * 4;                  // float offset of the lane's source chunk within the 32-wide k chunk
        int m = m0 + row; m = m < p.M ? m : p.M - 1;
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, pix = m - b * hw;
            cy[j] = pix / p.Wout; cx[j] = pix - cy[j] * p.Wout; cimg[j] = b * p.Hin * p.Win; csw[j] = sw;
            a_src[j] = p.a;
        } else {
            a_src[j] = p.a + (size_t)m * p.lda + sw;
        }
    }
    const float* w_src[WPW];
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        const int row = (wave * WPW + j) * 8 + prow;
        int n = n0 + row; n = n < p.N ? n : p.N - 1;
        w_src[j] = p.w + (size_t)n * p.K + (ppos ^ ((row >> 1) & 7)) * 4;
    }
    auto set_tap = [&](int tap) __attribute__((always_inline)) {       // CONV: the lane's source pixel of every A piece for this tap
        const int dy = tap / 3 - 1 - p.off, dx = tap % 3 - 1 - p.off;
        const int Hs = p.Hin << p.up, Ws = p.Win << p.up;
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            const int yy = cy[j] + dy, xx = cx[j] + dx;
            const int sy = reflect1(yy, Hs) >> p.up, sx = reflect1(xx, Ws) >> p.up;
            const bool outside = p.zero_pad && ((unsigned)yy >= (unsigned)Hs || (unsigned)xx >= (unsigned)Ws);
            a_src[j] = outside ? (const float*)g_zero_row + csw[j]
                               : p.a + ((size_t)(cimg[j] + sy * p.Win + sx)) * p.Cin + csw[j];
        }
    };
    int i_tap = 0, i_c0 = 0;                                           // (tap, channel offset) of the next chunk to issue (CONV)
    if (CONV) set_tap(0);
    auto issue = [&](int kt) __attribute__((always_inline)) {
        unsigned char* slot = ring + (kt % NSTAGE) * STAGE;
        const int ka = CONV ? i_c0 : kt * BK;
#pragma unroll
        for (int j = 0; j < APW; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[j] + ka),
                                             (__attribute__((address_space(3))) void*)(slot + (wave * APW + j) * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < WPW; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[j] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(slot + BM * 128 + (wave * WPW + j) * 1024), 16, 0, 0);
        if (CONV) {
            i_c0 += BK;
            if (i_c0 == p.Cin) { i_c0 = 0; ++i_tap; if (i_tap < 9) set_tap(i_tap); }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.0f;

    const int nk = p.K / BK;
#pragma unroll
    for (int t = 0; t < NSTAGE - 1; ++t) if (t < nk) issue(t);

    const int fsw = (r >> 1) & 7;                                      // swizzle of this lane's fragment rows (tile rows are multiples of 32 apart)
    for (int kt = 0; kt < nk; ++kt) {
        if (NSTAGE == 3 && kt + 1 < nk) wait_vmcnt<PPW>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + NSTAGE - 1 < nk) issue(kt + NSTAGE - 1);
        const unsigned char* ab = ring + (kt % NSTAGE) * STAGE + (wm + r) * 128;
        const unsigned char* wb = ring + (kt % NSTAGE) * STAGE + BM * 128 + (wn + r) * 128;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int off = ((4 * h + q) ^ fsw) << 4;
            f32x4 av[TM], wv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = *(const f32x4*)(ab + i * 32 * 128 + off);
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) wv[jn] = *(const f32x4*)(wb + jn * 32 * 128 + off);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], wv[jn][e], acc[i][jn], 0, 0, 0);
        }
        asm volatile("" ::: "memory");                                 // the slot is refilled behind the next barrier: no read may sink below it
    }

    // ---- epilogue --------------------------------------------------------------------------------------------------------------------
    // r02's element-wise form (one 4-byte load / store per accumulator register behind per-element branches; hipcc serialised it into
    // load -> s_waitcnt vmcnt(0) -> store chains, ~35 k clocks per 128 x 128 tile: it, not the main loop, bounded the Linear shapes)
    // is kept only for outputs it alone can address (NCHW image, N % 4 != 0, unaligned operands).  Everything else goes through the
    // idle ring: the accumulators are written as the [BM][BN] fp32 tile (32 consecutive floats per half-wave: conflict-free), then
    // every thread takes whole 16-byte pieces of rows: coalesced float4 loads of bias / r1 / r2, float4 stores, no per-element branch.
    const bool vec_ok = (p.N & 3) == 0 && !(CONV && p.out_nchw) && (p.ldo & 3) == 0 && (((uintptr_t)p.out) & 15) == 0 &&
                        (!p.aux || (((uintptr_t)p.aux) & 15) == 0) && (!p.bias || (((uintptr_t)p.bias) & 15) == 0) &&
                        ((!p.r1 && !p.r2) || ((p.ldr & 3) == 0 && ((((uintptr_t)p.r1) | ((uintptr_t)p.r2)) & 15) == 0));
    if (vec_ok) {
        constexpr int NT = NWAVES * 64;
        constexpr int VPR = BN / 4;                                       // float4 pieces per tile row
        constexpr int PIECES = BM * VPR / NT;
        static_assert((BM * VPR) % NT == 0, "row pieces must divide over the threads");
        static_assert(BM * BN * 4 <= NSTAGE * STAGE, "the output tile must fit in the ring");
        float* tile = (float*)ring;
        __builtin_amdgcn_s_barrier();                                      // every wave has left the main loop: the ring is free
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    tile[(wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * BN + wn + jn * 32 + r] = acc[i][jn][e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int c4 = tid % VPR, row0 = tid / VPR;
        int n = n0 + c4 * 4;
        const bool n_ok = n < p.N;
        n = n_ok ? n : p.N - 4;                                            // clamped: loads stay in bounds, the store is predicated
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *(const f32x4*)(p.bias + n);
        constexpr int BATCH = PIECES < 4 ? PIECES : 4;                    // pieces in flight per thread: 4 x (tile + r1 + r2) = 48 registers
#pragma unroll
        for (int b0 = 0; b0 < PIECES; b0 += BATCH) {
            f32x4 v[BATCH], a1[BATCH], a2[BATCH];
            size_t orow[BATCH];
            bool ok[BATCH];
#pragma unroll
            for (int it = 0; it < BATCH; ++it) {
                const int row = row0 + (b0 + it) * (NT / VPR);
                int m = m0 + row;
                ok[it] = n_ok && m < p.M;
                m = m < p.M ? m : p.M - 1;
                orow[it] = (size_t)m;
                v[it] = *(const f32x4*)(tile + row * BN + c4 * 4);
                if (p.r1) a1[it] = *(const f32x4*)(p.r1 + (size_t)m * p.ldr + n);
                if (p.r2) a2[it] = *(const f32x4*)(p.r2 + (size_t)m * p.ldr + n);
            }
#pragma unroll
            for (int it = 0; it < BATCH; ++it) {
                f32x4 x = v[it] + bv;
                if (p.aux && ok[it]) *(f32x4*)(p.aux + orow[it] * p.ldo + n) = x;
                if (p.act == MASTER_ACT_GELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[c] = gelu_erf(x[c]);
                } else if (p.act == MASTER_ACT_RELU) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[c] = relu_np(x[c]);
                }
                if (p.mask_relu) {
                    if (p.r2) x += a2[it];
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[c] = a1[it][c] > 0.0f ? x[c] : 0.0f;
                } else {
                    if (p.r1) x *= a1[it];
                    if (!CONV && p.rowscale) x *= tile_sc_ok ? tile_sc : p.rowscale[p.rs_magic ? (int)__umulhi((unsigned)orow[it], p.rs_magic) : (int)orow[it] / p.rs_rows];
                    if (p.r2) x += a2[it];
                }
                if (ok[it]) *(f32x4*)(p.out + orow[it] * p.ldo + n) = x;
            }
            __builtin_amdgcn_sched_barrier(0);                             // keep the batches apart (max-ilp would hoist every load: 246 VGPRs)
        }
        return;
    }
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
                else if (p.act == MASTER_ACT_RELU) v = relu_np(v);
                if (p.mask_relu) {
                    if (p.r2) v += p.r2[(size_t)m * p.ldr + n];
                    if (!(p.r1[(size_t)m * p.ldr + n] > 0.0f)) v = 0.0f;
                } else {
                    if (p.r1) v *= p.r1[(size_t)m * p.ldr + n];
                    if (!CONV && p.rowscale) v *= p.rowscale[p.rs_magic ? (int)__umulhi((unsigned)m, p.rs_magic) : m / p.rs_rows];
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

template <int BM, int BN, int WM, int WN, bool CONV, int NSTAGE>
int launch_dma(const GemmArgs& p, hipStream_t s) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    const int MT = (p.M + BM - 1) / BM, NTN = (p.N + BN - 1) / BN;
    const int grid = ((MT + 7) / 8) * 8 * NTN;
    const size_t lds = (size_t)NSTAGE * (BM + BN) * 128;
    static LdsLimit lds_limit;
    lds_limit.ensure((const void*)gemm_f32_dma_kernel<BM, BN, WM, WN, CONV, NSTAGE>, (int)lds);
    hipLaunchKernelGGL((gemm_f32_dma_kernel<BM, BN, WM, WN, CONV, NSTAGE>), dim3(grid), dim3(NT), lds, s, p);
    return master_check_launch("gemm_f32_dma");
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

// ---------------------------------------------------------------------------------------------------------------------------------
// r03: 3x3 convolution with a handful of output channels (the decoder's last layer, decoder.py:54: 32 -> 3) as a direct
// convolution on the vector ALUs.  On the implicit-GEMM kernel the 3 outputs occupy a 32-wide N tile (10x wasted matrix work:
// 555 us at bs = 32 for 3.6 GFLOP); the layer is a stream of 128-byte pixels, so one thread takes one output pixel, walks its
// nine taps with 16-byte loads (neighbouring lanes share six of them through L1 / L2) and keeps COUT accumulators; the weights
// (COUT x 288 floats) sit in LDS and are read as wave-wide broadcasts.  Same reflect / upsample / zero-pad index rules as the gather
// above; fp32 FMA chains (exact products, a different summation order than the MFMA's: ~1e-6).
// ---------------------------------------------------------------------------------------------------------------------------------
// Tile form: a workgroup owns a 16 x 16 block of output pixels; the (16 + 2) x (16 + 2) source pixels it needs (after reflect / upsample
// index math; zero rows for out-of-image taps of a zero-padded convolution) are copied ONCE into LDS with coalesced 16-byte loads -
// pixel rows padded to Cin + 4 floats, so that the 16 lanes of a ds_read_b128 group (16 neighbouring pixels) hit 16 distinct bank
// groups - and every thread then reads its nine taps from LDS instead of 72 strided 16-byte global loads (267 us as a direct
// global-memory convolution: the texture path, not HBM, bounded it).
template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_small_cout_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];            // halo tile [18][18][Cin + 4]
    const int K = 9 * p.Cin, PS = p.Cin + 4;
    float* tile = sm;
    const int tiles_x = (p.Wout + 15) >> 4, tiles_y = (p.Hout + 15) >> 4;
    const int b = blockIdx.x / (tiles_x * tiles_y), t = blockIdx.x - b * (tiles_x * tiles_y);
    const int y0 = (t / tiles_x) << 4, x0 = (t % tiles_x) << 4;
    const int Hs = p.Hin << p.up, Ws = p.Win << p.up;
    const int c4n = p.Cin >> 2;                                           // 16-byte chunks per pixel
    for (int i = threadIdx.x; i < 18 * 18 * c4n; i += 256) {
        const int pix = i / c4n, c4 = i - pix * c4n;
        const int yy = y0 + pix / 18 - 1, xx = x0 + pix % 18 - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const bool outside = (unsigned)yy >= (unsigned)Hs || (unsigned)xx >= (unsigned)Ws;
        if (!(p.zero_pad && outside) && yy >= -1 && xx >= -1 && yy <= Hs && xx <= Ws) {
            const int sy = reflect1(yy, Hs) >> p.up, sx = reflect1(xx, Ws) >> p.up;
            if ((unsigned)sy < (unsigned)p.Hin && (unsigned)sx < (unsigned)p.Win)
                v = *(const f32x4*)(p.a + ((size_t)(b * p.Hin + sy) * p.Win + sx) * p.Cin + c4 * 4);
        }
        *(f32x4*)(tile + pix * PS + c4 * 4) = v;
    }
    __syncthreads();
    const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
    const int y = y0 + ly, x = x0 + lx;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = p.bias ? p.bias[co] : 0.0f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const float* src = tile + ((ly + tap / 3) * 18 + lx + tap % 3) * PS;
        const float* wt = p.w + tap * p.Cin;                              // wave-uniform addresses: scalar loads, the weight is an SGPR operand of the FMA
#pragma unroll 8
        for (int c = 0; c < p.Cin; c += 4) {
            const f32x4 xv = *(const f32x4*)(src + c);
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const f32x4 wv = *(const f32x4*)(wt + co * K + c);
                acc[co] = fmaf(xv[0], wv[0], acc[co]); acc[co] = fmaf(xv[1], wv[1], acc[co]);
                acc[co] = fmaf(xv[2], wv[2], acc[co]); acc[co] = fmaf(xv[3], wv[3], acc[co]);
            }
        }
    }
    if (y >= p.Hout || x >= p.Wout) return;
    const int hw = p.Hout * p.Wout, pix = y * p.Wout + x;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        float v = acc[co];
        if (p.act == MASTER_ACT_GELU) v = gelu_erf(v);
        else if (p.act == MASTER_ACT_RELU) v = relu_np(v);
        if (p.out_nchw) p.out[((size_t)b * COUT + co) * hw + pix] = v;
        else p.out[((size_t)b * hw + pix) * p.ldo + co] = v;
    }
}

template <int COUT>
int launch_small_cout(const GemmArgs& p, hipStream_t s) {
    const size_t lds = (size_t)18 * 18 * (p.Cin + 4) * sizeof(float);
    const int B = p.M / (p.Hout * p.Wout);
    static LdsLimit lds_limit;
    lds_limit.ensure((const void*)conv3x3_small_cout_kernel<COUT>, (int)lds);
    hipLaunchKernelGGL(conv3x3_small_cout_kernel<COUT>, dim3(B * ((p.Hout + 15) / 16) * ((p.Wout + 15) / 16)), dim3(256), lds, s, p);
    return master_check_launch("conv3x3_small_cout");
}

// MASTER_GEMM_F32 = "reg": the register-staged r01 / r02 kernel; "dma" (default): the LDS-DMA ring.  MASTER_GEMM_F32_SMALL = 0 / 1:
// 64x64 tiles (3-deep ring, 4-5 workgroups per CU) for problems with fewer than 256 tiles of 128x128 (train step: 8 192-token matrices).
static int gemm_variant() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MASTER_GEMM_F32"); v = (e && e[0] == 'r') ? 0 : 1; }
    return v;
}
static int gemm_small64() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MASTER_GEMM_F32_SMALL"); v = e ? atoi(e) : 1; }
    return v;
}

template <bool CONV>
int dispatch(const GemmArgs& p, hipStream_t s) {
    if (CONV && p.N <= 4 && p.N >= 1 && gemm_variant() == 1 && !p.r1 && !p.r2 && !p.aux && !p.mask_relu && p.off == 0 && (p.Cin & 3) == 0 &&
        p.Cin <= 96 && (((uintptr_t)p.a) & 15) == 0) {
        switch (p.N) {
            case 1: return launch_small_cout<1>(p, s);
            case 2: return launch_small_cout<2>(p, s);
            case 3: return launch_small_cout<3>(p, s);
            default: return launch_small_cout<4>(p, s);
        }
    }
    const bool few_tiles = p.N > 64 && (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) < 256;
    const bool dma_ok = (p.K % BK) == 0 && (!CONV || (p.Cin % BK == 0 && p.Cin <= 512)) &&
                        (((uintptr_t)p.a | (uintptr_t)p.w) & 15) == 0 && (CONV || (p.lda % 4) == 0);
    if (gemm_variant() == 1 && dma_ok) {
        // a short contraction into a narrow output (K <= 128, N <= 128: the Swin stage-1 proj) is all prologue and epilogue on 128x128 tiles:
        // 64x64 tiles (3-deep ring, three to five workgroups per CU) take 135 us against 154 at M = 262 144; wider or deeper shapes lose
        // 4-8 % on them (measured), so only there.  MASTER_GEMM_F32_SHORTK=0 switches it off.
        static int shortk = -1;
        if (shortk < 0) { const char* e = getenv("MASTER_GEMM_F32_SHORTK"); shortk = e ? atoi(e) : 1; }
        if (shortk && !CONV && p.K <= 128 && p.N > 32 && p.N <= 128) return launch_dma<64, 64, 32, 32, CONV, 3>(p, s);
        if (few_tiles) {
            if (gemm_small64()) return launch_dma<64, 64, 32, 32, CONV, 3>(p, s);
            return launch_dma<64, 128, 32, 64, CONV, 2>(p, s);
        }
        if (p.N > 64) return launch_dma<128, 128, 64, 64, CONV, 2>(p, s);
        if (p.N > 32) return launch_dma<128, 64, 64, 32, CONV, 2>(p, s);
        return launch_dma<256, 32, 64, 32, CONV, 2>(p, s);
    }
    // fewer than one 128x128 tile per CU (train step: 8 192-token matrices): 64-row tiles double the number of workgroups
    // (forcing 64-row tiles on the larger inference shapes was measured 5-10 % slower: profiles/r02_ops_fp32_gemm_{base,tile64}.txt)
    if (few_tiles) return launch<64, 128, 32, 64, CONV>(p, s);
    if (p.N > 64) return launch<128, 128, 64, 64, CONV>(p, s);
    if (p.N > 32) return launch<128, 64, 64, 32, CONV>(p, s);
    return launch<256, 32, 64, 32, CONV>(p, s);
}

}  // namespace

int gemm_f32_linear(const float* a, const float* w, const float* bias, const float* r1, const float* r2, float* out,
                    int M, int N, int K, int lda, int ldo, int ldr, int act, hipStream_t s, float* aux = nullptr, const float* rowscale = nullptr,
                    int rs_rows = 1) {
    const unsigned long long d = (unsigned long long)(rs_rows > 0 ? rs_rows : 1);
    const unsigned magic = (rowscale && d > 1 && (unsigned long long)M * d < (1ull << 32)) ? (unsigned)(((1ull << 32) + d - 1) / d) : 0u;
    GemmArgs p{a, w, bias, r1, r2, out, M, N, K, lda, ldo, ldr, act, 0, 0, 0, 1, 1, 0, 0, 0, aux, 0, 0, rowscale, rs_rows > 0 ? rs_rows : 1, magic};
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
