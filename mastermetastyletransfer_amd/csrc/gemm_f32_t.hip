// fp32 GEMMs of the backward pass on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32): operands that are NOT contraction-contiguous.
//
//   data gradient of a Linear    dX[M,K'] = G[M,N'] . W[N',K']          A = G  row-major (contraction contiguous), B = W  [contraction][n]
//   weight gradient of a Linear  dW[N',K'] = G[M,N']^T . X[M,K']        A = G  [contraction][m],                   B = X  [contraction][n]
//   weight gradient of a 3x3 conv dW[Cout][tap][Cin] = sum_pixels G[pix][Cout] . gather(X)[pix][tap][Cin]          B = implicit gather
//
// (the forward kernel gemm_f32.hip has both operands contraction-contiguous.)  A contraction-major operand is staged the way it lies in
// memory - 16-byte loads along m (or n), LDS tile [32 k][BM + 4] - and its MFMA fragments are single-float LDS reads with the lane on
// m: 32 consecutive words per half-wave, conflict-free.  MFMA k-step s pairs contraction index s (lanes 0-31) with 16 + s (lanes 32-63),
// the pairing of gemm_f32.hip, so a row-major A keeps its ds_read_b128 path.
//
// Weight gradients contract over all tokens / pixels of the batch (8 192 ... 524 288) into a small output, so the contraction is split
// over blockIdx.y: every split writes its partial tile to a workspace slab and reduce_slabs_kernel (train_misc.hip) adds the slabs in a fixed order
// (deterministic, no atomics).
//
// Bias gradients ride along (r03): the weight-gradient forms already stream G through the workgroups, so the workgroups of the first
// column tile also add up the G values they stage (one VALU add per 16-byte load, next to the MFMAs) and write  rowsum[m] = sum over the
// split's contraction rows of G[k][m]  - the column sums of G, i.e. the bias gradient - to a second slab that the same reduce launch
// folds.  This replaced one colsum launch + one reduce launch per Linear / convolution with a bias (88 + 88 launches of a train step).
#include "common.h"
#include <stdlib.h>

int reduce_slabs_launch(const float* part, float* out, size_t n, size_t slab, int splits, hipStream_t s, int accumulate = 0);   // train_misc.hip
int reduce_slabs2_launch(const float* part, float* out, size_t n, size_t slab, const float* part2, float* out2, size_t n2, size_t slab2, int splits,
                         hipStream_t s, int accumulate);                                                                           // train_misc.hip

namespace {

constexpr int BK = 32;
constexpr int LDK = 36;     // row-major A: padded LDS row (floats)
constexpr int PADT = 4;     // contraction-major tiles: row = BM + PADT floats

struct TArgs {
    const float* a; const float* w; float* out;
    int M, N, K;                 // out is [M][N]; K = contraction length
    int lda, ldw, ldo;
    int chunks_per_split;        // BK-chunks per blockIdx.y
    size_t slab;                 // elements between split slabs of `out` (0: one split, final result)
    // epilogue of the data-gradient form (one split only):  out *= gelu'(aux)   (aux = pre-activation of the layer below)
    const float* aux; int ld_aux;
    // conv weight gradient: w = x NHWC [B,Hin,Win,Cin]; contraction index = output pixel of the Hout x Wout map; column n = tap*Cin + ci
    int Hin, Win, Cin, Hout, Wout, up, zero_pad;
    // contraction-major A only: rowsum[blockIdx.y * rowsum_slab + m] = sum over this split's contraction rows of a[k][m]  (NULL: not wanted)
    float* rowsum; size_t rowsum_slab;
    // data-gradient form (one split): out += add[m][n] after the GELU factor - the gradient arriving over a skip connection (r03: autograd
    // added it with one elementwise launch per residual block)
    const float* add; int ld_add;
    int xcd_remap;               // set by the launcher when gridDim.x * gridDim.y is a multiple of 8
    // The upstream gradient G belongs to a residual branch under StochasticDepth(p, "row"): its rows carry one factor per image,
    // rowscale[row / rs_rows] (r03: was a separate elementwise launch producing a scaled copy of G).
    //   data-gradient form (A = G row-major):  out[m] *= rowscale[m / rs_rows] in the epilogue, before `add`;
    //   weight-gradient form (A = G with the contraction index on the rows): staged rows k are scaled by rowscale[k / rs_rows].
    const float* rowscale; int rs_rows; unsigned rs_magic;      // rs_magic = ceil(2^32 / rs_rows) when rows * rs_rows < 2^32: row / rs_rows = umulhi(row, magic)
};

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// row -> image index of a per-image row scale: one v_mul_hi instead of an integer division (~25 instructions) per staged piece
__device__ __forceinline__ int rs_index(unsigned row, int rows, unsigned magic) { return magic ? (int)__umulhi(row, magic) : (int)(row / (unsigned)rows); }

__device__ __forceinline__ float gelu_grad(float x) {
    // d/dx [ x Phi(x) ] = Phi(x) + x phi(x)
    const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752440f));
    const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.5f * x * x * 1.4426950408889634f);
    return fmaf(x, pdf, cdf);
}

template <int BM, int BN, int WM, int WN, bool AT, bool GATHER>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64) void gemm_f32_t_kernel(TArgs p) {
    constexpr int NWN = BN / WN;
    constexpr int NT = (BM / WM) * NWN * 64;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int LDA_T = BM + PADT, LDW_T = BN + PADT;
    constexpr int A_ELEMS = AT ? BK * LDA_T : BM * LDK;
    constexpr int W_ELEMS = BK * LDW_T;
    constexpr int ACH = AT ? (BK * BM / 4 + NT - 1) / NT : BM * 8 / NT;     // 16-byte chunks per thread and stage
    constexpr int WCH = (BK * BN / 4 + NT - 1) / NT;
    static_assert(AT || (BM * 8) % NT == 0, "A chunks must divide evenly");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][A_ELEMS]
    float* Ws = smem + 2 * A_ELEMS;         // [2][W_ELEMS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / NWN) * WM, wn = (wave % NWN) * WN;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware order (r03).  Workgroups go to the 8 XCDs round-robin in dispatch order (x fastest), and each XCD has its own L2: with the
    // plain order the N-tiles of one M-tile (data gradient: same rows of G) or the tiles of one split (weight gradient: same token range
    // of G and X) sat on different XCDs and every one of them fetched those rows again - the [8192,256].[256,256] data gradient read
    // 73 MB for 16 MB of operands (profiles/r03_train_pmc_traffic.txt).  Remapped so that runs of total/8 consecutive logical ids share an XCD.
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_remap) {
        const int lin = bx + by * (int)gridDim.x, per = ((int)gridDim.x * (int)gridDim.y) >> 3;
        const int logical = (lin & 7) * per + (lin >> 3);
        by = logical / (int)gridDim.x; bx = logical - by * (int)gridDim.x;
    }
    const int NTN = (p.N + BN - 1) / BN;
    const int mt = bx / NTN, nt = bx % NTN;
    const int m0 = mt * BM, n0 = nt * BN;
    const int nk_all = (p.K + BK - 1) / BK;
    const int kt0 = by * p.chunks_per_split;
    int kt1 = kt0 + p.chunks_per_split; if (kt1 > nk_all) kt1 = nk_all;
    float* outp = p.out + (size_t)by * p.slab;

    // data-gradient form with a per-image row scale: when the tile's rows belong to ONE image (always, for maps of a multiple of BM tokens)
    // the factor is fetched here, long before the epilogue needs it; a tile that straddles two images looks its rows up one by one
    float tile_sc = 1.0f; bool tile_sc_ok = false;
    if (!AT && p.rowscale) {
        const int mlast = (m0 + BM - 1 < p.M ? m0 + BM - 1 : p.M - 1);
        const int i0 = rs_index((unsigned)m0, p.rs_rows, p.rs_magic), i1 = rs_index((unsigned)mlast, p.rs_rows, p.rs_magic);
        tile_sc_ok = i0 == i1;
        tile_sc = p.rowscale[i0];
    }
    const bool a_vec = (p.lda & 3) == 0 && (((uintptr_t)p.a) & 15) == 0;      // else scalar loads (e.g. a 3-channel gradient)

    // ---- W^T gather descriptors (conv weight gradient): column group -> (tap, ci) is fixed per thread and chunk slot ----------------
    int g_dy[WCH], g_dx[WCH], g_ci[WCH];
    if (GATHER) {
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int c = tid + i * NT;
            int n = n0 + (c % (BN / 4)) * 4;
            if (n >= p.N) n = 0;
            const int tap = n / p.Cin;
            g_ci[i] = n - tap * p.Cin; g_dy[i] = tap / 3 - 1; g_dx[i] = tap % 3 - 1;
        }
    }

    f32x4 a_st[ACH], w_st[WCH];
    float a_sc[ACH];                                             // per staged A piece: the row scale of its contraction row (weight-gradient form)
#pragma unroll
    for (int i = 0; i < ACH; ++i) a_sc[i] = 1.0f;
    // bias gradient: thread c = tid + i NT stages contraction row c / (BM/4), columns (c % (BM/4)) * 4 .. +3 of every chunk; NT is a multiple
    // of BM/4, so all of a thread's pieces are the same four columns: one running f32x4, added where the staged values are consumed
    // (stage_store) - next to the load it would put a wait for the global loads in front of the MFMAs they are meant to hide under
    static_assert(!AT || NT % (BM / 4) == 0, "a thread's A pieces must share their columns");
    f32x4 a_sum = {0.f, 0.f, 0.f, 0.f};
    const bool do_rowsum = AT && p.rowsum != nullptr && nt == 0;               // workgroup-uniform
    auto stage_load = [&](int kt) {
        const int k0 = kt * BK;
        if (AT) {
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                const int c = tid + i * NT;
                const int kr = c / (BM / 4), m = m0 + (c % (BM / 4)) * 4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (c < BK * BM / 4 && k0 + kr < p.K) {
                    const float* src = p.a + (size_t)(k0 + kr) * p.lda + m;
                    if (a_vec && m + 3 < p.M) v = *(const f32x4*)src;
                    else
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (m + e < p.M) v[e] = src[e];
                    // the row's StochasticDepth factor is fetched here but applied where the staged values are consumed (stage_store): a
                    // multiply next to the load would put a wait for the global loads in front of the MFMAs they are meant to hide under
                    if (p.rowscale) a_sc[i] = p.rowscale[rs_index((unsigned)(k0 + kr), p.rs_rows, p.rs_magic)];
                }
                a_st[i] = v;
            }
        } else {
            const int kc = (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                int m = m0 + (tid >> 3) + i * (NT / 8);
                m = m < p.M ? m : p.M - 1;
                a_st[i] = *(const f32x4*)(p.a + (size_t)m * p.lda + k0 + kc);        // K % 32 == 0 in this form
            }
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int c = tid + i * NT;
            const int kr = c / (BN / 4), n = n0 + (c % (BN / 4)) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (c < BK * BN / 4 && k0 + kr < p.K && n < p.N) {             // N % 4 == 0 (checked by the launcher)
                if (GATHER) {
                    const int pix = k0 + kr, hw = p.Hout * p.Wout;
                    const int b = pix / hw, rem = pix - b * hw, y = rem / p.Wout, x = rem - y * p.Wout;
                    const int yy = y + g_dy[i], xx = x + g_dx[i];
                    const bool inside = (unsigned)yy < (unsigned)p.Hout && (unsigned)xx < (unsigned)p.Wout;
                    if (inside || !p.zero_pad) {
                        const int sy = reflect1(yy, p.Hout) >> p.up, sx = reflect1(xx, p.Wout) >> p.up;
                        v = *(const f32x4*)(p.w + ((size_t)(b * p.Hin + sy) * p.Win + sx) * p.Cin + g_ci[i]);
                    }
                } else {
                    v = *(const f32x4*)(p.w + (size_t)(k0 + kr) * p.ldw + n);
                }
            }
            w_st[i] = v;
        }
    };
    auto stage_store = [&](int buf) {
        if (AT) {
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                const int c = tid + i * NT;
                if (p.rowscale) a_st[i] *= a_sc[i];
                if (c < BK * BM / 4) *(f32x4*)(As + buf * A_ELEMS + (c / (BM / 4)) * LDA_T + (c % (BM / 4)) * 4) = a_st[i];
                if (do_rowsum) a_sum += a_st[i];                               // zeros where c or the row is out of range; every chunk passes here once
            }
        } else {
            const int kc = (tid & 7) * 4;
#pragma unroll
            for (int i = 0; i < ACH; ++i) *(f32x4*)(As + buf * A_ELEMS + ((tid >> 3) + i * (NT / 8)) * LDK + kc) = a_st[i];
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int c = tid + i * NT;
            if (c < BK * BN / 4) *(f32x4*)(Ws + buf * W_ELEMS + (c / (BN / 4)) * LDW_T + (c % (BN / 4)) * 4) = w_st[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.0f;

    if (kt0 < kt1) {
        stage_load(kt0);
        stage_store(0);
        __syncthreads();
        for (int kt = kt0; kt < kt1; ++kt) {
            const int buf = (kt - kt0) & 1;
            if (kt + 1 < kt1) stage_load(kt + 1);
            const float* ab = As + buf * A_ELEMS;
            const float* wb = Ws + buf * W_ELEMS;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 av[TM];
                if (!AT) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) av[i] = *(const f32x4*)(ab + (wm + i * 32 + r) * LDK + 16 * h + 4 * q);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kk = 16 * h + 4 * q + e;
                    float a1[TM], w1[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a1[i] = AT ? ab[kk * LDA_T + wm + i * 32 + r] : av[i][e];
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) w1[jn] = wb[kk * LDW_T + wn + jn * 32 + r];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int jn = 0; jn < TN; ++jn)
                            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], w1[jn], acc[i][jn], 0, 0, 0);
                }
            }
            if (kt + 1 < kt1) stage_store(buf ^ 1);
            __syncthreads();
        }
    }

    if (AT && do_rowsum) {
        constexpr int GROUPS = NT / (BM / 4);                                  // threads holding partial sums of the same four columns
        float* red = smem;                                                     // [GROUPS][BM]; the main loop's last barrier has passed
        *(f32x4*)(red + (tid / (BM / 4)) * BM + (tid % (BM / 4)) * 4) = a_sum;
        __syncthreads();
        if (tid < BM && m0 + tid < p.M) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < GROUPS; ++q) t += red[q * BM + tid];
            p.rowsum[(size_t)by * p.rowsum_slab + m0 + tid] = t;
        }
        // the epilogue's own first barrier orders these reads before the tile goes into the same memory
    }

    // ---- epilogue --------------------------------------------------------------------------------------------------------------------
    // r03: through the idle staging buffers as the [BM][BN] fp32 tile, then whole 16-byte row pieces per thread (coalesced float4
    // loads of the GELU pre-activation, float4 stores); the element-wise form below (4-byte accesses behind per-element branches,
    // serialised by hipcc into load -> wait -> store chains) stays for outputs it alone can address.
    const bool vec_ok = (p.N & 3) == 0 && (p.ldo & 3) == 0 && (((uintptr_t)outp) & 15) == 0 &&
                        (!p.aux || ((p.ld_aux & 3) == 0 && (((uintptr_t)p.aux) & 15) == 0)) &&
                        (!p.add || ((p.ld_add & 3) == 0 && (((uintptr_t)p.add) & 15) == 0));
    if (vec_ok) {
        constexpr int VPR = BN / 4;
        constexpr int PIECES = BM * VPR / NT;
        static_assert((BM * VPR) % NT == 0, "row pieces must divide over the threads");
        static_assert(BM * BN <= 2 * (A_ELEMS + W_ELEMS), "the output tile must fit in the staging buffers");
        float* tile = smem;
        __syncthreads();                                                   // every wave has left the main loop
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    tile[(wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * BN + wn + jn * 32 + r] = acc[i][jn][e];
        __syncthreads();
        const int c4 = tid % VPR, row0 = tid / VPR;
        int n = n0 + c4 * 4;
        const bool n_ok = n < p.N;
        n = n_ok ? n : p.N - 4;
        constexpr int BATCH = PIECES < 4 ? PIECES : 4;
#pragma unroll
        for (int b0 = 0; b0 < PIECES; b0 += BATCH) {
            f32x4 v[BATCH], ax[BATCH], ad[BATCH];
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
                if (p.aux) ax[it] = *(const f32x4*)(p.aux + (size_t)m * p.ld_aux + n);
                if (p.add) ad[it] = *(const f32x4*)(p.add + (size_t)m * p.ld_add + n);
            }
#pragma unroll
            for (int it = 0; it < BATCH; ++it) {
                f32x4 x = v[it];
                if (p.aux) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[c] *= gelu_grad(ax[it][c]);
                }
                if (!AT && p.rowscale) x *= tile_sc_ok ? tile_sc : p.rowscale[rs_index((unsigned)orow[it], p.rs_rows, p.rs_magic)];
                if (p.add) x += ad[it];
                if (ok[it]) *(f32x4*)(outp + orow[it] * p.ldo + n) = x;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
        const int n = n0 + wn + jn * 32 + r;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m >= p.M) continue;
                float v = acc[i][jn][e];
                if (p.aux) v *= gelu_grad(p.aux[(size_t)m * p.ld_aux + n]);
                if (!AT && p.rowscale) v *= tile_sc_ok ? tile_sc : p.rowscale[rs_index((unsigned)m, p.rs_rows, p.rs_magic)];
                if (p.add) v += p.add[(size_t)m * p.ld_add + n];
                outp[(size_t)m * p.ldo + n] = v;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, bool AT, bool GATHER>
int launch_t(TArgs p, float* final_out, float* ws, size_t ws_floats, int want_splits, hipStream_t s, int accumulate = 0, float* rowsum_out = nullptr) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    constexpr int A_ELEMS = AT ? BK * (BM + PADT) : BM * LDK;
    constexpr int W_ELEMS = BK * (BN + PADT);
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const int nk = (p.K + BK - 1) / BK;
    if (rowsum_out && !AT) { master_set_error("gemm_f32_t: row sums need the contraction-major A form"); return MASTER_ERR_INVALID; }
    int splits = want_splits;
    if (splits <= 0) {                                       // two workgroups per CU
        splits = (512 + tiles - 1) / tiles;
        if (splits > nk / 4) splits = nk / 4;                // keep >= 4 chunks per split
    }
    if (splits < 1) splits = 1;
    const size_t slab = ((size_t)p.M * p.ldo + 3) & ~(size_t)3;
    const size_t rs_slab = rowsum_out ? (((size_t)p.M + 3) & ~(size_t)3) : 0;        // the row-sum slabs follow the output slabs in ws
    if (splits > 1 && (ws == nullptr || ws_floats < (slab + rs_slab) * splits)) {
        splits = ws ? (int)(ws_floats / (slab + rs_slab)) : 1;
        if (splits < 1) splits = 1;
    }
    p.chunks_per_split = (nk + splits - 1) / splits;
    splits = (nk + p.chunks_per_split - 1) / p.chunks_per_split;
    // accumulate (final_out += result): always through the workspace and the slab reduce, which adds to final_out
    const bool via_ws = splits > 1 || accumulate;
    if (accumulate && (ws == nullptr || ws_floats < slab + rs_slab)) { master_set_error("gemm_f32_t: accumulating needs a workspace of at least one slab"); return MASTER_ERR_INVALID; }
    if (via_ws) { p.out = ws; p.slab = slab; p.rowsum = rowsum_out ? ws + slab * splits : nullptr; p.rowsum_slab = rs_slab; }
    else { p.out = final_out; p.slab = 0; p.rowsum = rowsum_out; p.rowsum_slab = 0; }
    const size_t lds = (size_t)2 * (A_ELEMS + W_ELEMS) * sizeof(float);
    static int xcd = -1;
    if (xcd < 0) { const char* e = getenv("MASTER_GEMM_T_XCD"); xcd = e ? atoi(e) : 1; }
    p.xcd_remap = xcd && ((long long)tiles * splits) % 8 == 0 && (long long)tiles * splits >= 16;
    {   // exact for every row < 2^32 / rs_rows
        const unsigned long long rows = AT ? (unsigned long long)p.K : (unsigned long long)p.M, d = (unsigned long long)(p.rs_rows > 0 ? p.rs_rows : 1);
        p.rs_magic = (p.rowscale && d > 1 && rows * d < (1ull << 32)) ? (unsigned)(((1ull << 32) + d - 1) / d) : 0u;
    }
    static LdsLimit lds_limit;
    lds_limit.ensure((const void*)gemm_f32_t_kernel<BM, BN, WM, WN, AT, GATHER>, (int)lds);
    hipLaunchKernelGGL((gemm_f32_t_kernel<BM, BN, WM, WN, AT, GATHER>), dim3(tiles, splits), dim3(NT), lds, s, p);
    if (via_ws) {
        if (rowsum_out) return reduce_slabs2_launch(ws, final_out, (size_t)p.M * p.ldo, slab, p.rowsum, rowsum_out, (size_t)p.M, rs_slab, splits, s, accumulate);
        return reduce_slabs_launch(ws, final_out, (size_t)p.M * p.ldo, slab, splits, s, accumulate);
    }
    return master_check_launch("gemm_f32_t");
}

}  // namespace

// number of floats of workspace that lets a weight-gradient GEMM use its preferred split count
size_t gemm_f32_t_workspace(int M, int N) {
    const size_t slab = (((size_t)M * N + 3) & ~(size_t)3) + (((size_t)M + 3) & ~(size_t)3);     // output tile slab + the bias-gradient (row sum) slab
    size_t n = slab * 512, cap = (size_t)8 << 20;            // up to 512 splits, at most 32 MB - but never fewer than 8 slabs
    if (n > cap) n = cap;
    return n < slab * 8 ? slab * 8 : n;
}

// dX[M,Kc] = G[M,Nc] . W[Nc,Kc] (* gelu'(aux) when aux != NULL)
int gemm_f32_dgrad(const float* g, const float* w, float* out, const float* aux, const float* add, int M, int Nc, int Kc, int ldg, int ldo, int ld_aux,
                   int ld_add, hipStream_t s, const float* rowscale, int rs_rows) {
    TArgs p{g, w, out, M, Kc, Nc, ldg, Kc, ldo, 0, 0, aux, ld_aux, 0, 0, 0, 0, 0, 0, 0, nullptr, 0, add, ld_add, 0, rowscale, rs_rows > 0 ? rs_rows : 1};
    if (Kc > 64) {
        // fewer than one 128x128 tile per CU (the StyleTransformer's 8 192-token matrices): 64-row tiles double the workgroups
        // (r03: 64x64 tiles, four workgroups per CU, for the 8 192-token matrices: 512+ workgroups instead of 256 one-per-CU ones)
        if ((long long)((M + 127) / 128) * ((Kc + 127) / 128) < 256) {
            static int small64 = -1;
            if (small64 < 0) { const char* e = getenv("MASTER_GEMM_F32_SMALL"); small64 = e ? atoi(e) : 1; }
            if (small64) return launch_t<64, 64, 32, 32, false, false>(p, out, nullptr, 0, 1, s);
            return launch_t<64, 128, 32, 64, false, false>(p, out, nullptr, 0, 1, s);
        }
        return launch_t<128, 128, 64, 64, false, false>(p, out, nullptr, 0, 1, s);
    }
    return launch_t<128, 64, 64, 32, false, false>(p, out, nullptr, 0, 1, s);
}

// dW[Nc,Kc] = G[M,Nc]^T . X[M,Kc]      (dense output, ldo = Kc);  dbias[Nc] = column sums of G when dbias != NULL
int gemm_f32_wgrad(const float* g, const float* x, float* out, float* dbias, int M, int Nc, int Kc, int ldg, int ldx, float* ws, size_t ws_floats,
                   hipStream_t s, int accumulate, const float* rowscale, int rs_rows) {
    TArgs p{g, x, out, Nc, Kc, M, ldg, ldx, Kc, 0, 0, nullptr, 0, 0, 0, 0, 0, 0, 0, 0, nullptr, 0, nullptr, 0, 0, rowscale, rs_rows > 0 ? rs_rows : 1};
    if (Nc > 64) return launch_t<128, 128, 64, 64, true, false>(p, out, ws, ws_floats, 0, s, accumulate, dbias);
    return launch_t<32, 256, 32, 64, true, false>(p, out, ws, ws_floats, 0, s, accumulate, dbias);
}

// dWpacked[Cout][3][3][Cin] = sum over the B*Hout*Wout output pixels of G[pix][Cout] (x) gather(X)[pix][tap][Cin];  dbias[Cout] = column sums of G
int gemm_f32_conv3x3_wgrad(const float* g, const float* x, float* out, float* dbias, int B, int Hin, int Win, int Cin, int Cout, int ldg, int up,
                           int zero_pad, float* ws, size_t ws_floats, hipStream_t s) {
    const int Hout = Hin << up, Wout = Win << up;
    TArgs p{g, x, out, Cout, 9 * Cin, B * Hout * Wout, ldg, 0, 9 * Cin, 0, 0, nullptr, 0, Hin, Win, Cin, Hout, Wout, up, zero_pad, nullptr, 0, nullptr, 0};
    if (Cout > 64) return launch_t<128, 128, 64, 64, true, true>(p, out, ws, ws_floats, 0, s, 0, dbias);
    if (Cout > 32) return launch_t<64, 128, 32, 64, true, true>(p, out, ws, ws_floats, 0, s, 0, dbias);
    return launch_t<32, 256, 32, 64, true, true>(p, out, ws, ws_floats, 0, s, 0, dbias);
}
