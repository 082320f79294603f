// Split-bf16 ("bf16x3") GEMM family: the parity-grade fast mode (MASTER_F32X3).
//
//   out[M,N] = r1 (.) act(A[M,K] . W[N,K]^T + bias) + r2          A, r1, r2, out: fp32 in HBM;  W: pre-split weights
//
// Every fp32 operand x is written as x = hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi), and a product a*w is
// evaluated as  a_lo*w_hi + a_hi*w_lo + a_hi*w_hi  on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: three matrix-core
// instructions of 32 clocks where the exact-fp32 form (v_mfma_f32_32x32x2_f32) needs eight of 64, i.e. 833 TFLOP/s of peak
// instead of 157; the dropped lo*lo term and the two residuals are ~2^-16 relative per product (measured end to end:
// 1.2e-4 max-abs on the stylized image against the fp32 CPU path, bar 1e-3 - profiles/r04_x3_numerics_sim.txt).
//
//   * W arrives PRE-SPLIT (master_split_weights, once per weight): per row and 32-wide k chunk 64 B of hi then 64 B of lo =
//     the 128-byte row segment of the fp32 kernel (gemm_f32.hip), so the weight tile reaches LDS by the same LDS-DMA pieces
//     (8 rows x 128 B, source-swizzled 16-byte units: unit u of row R lands at position u ^ ((R >> 1) & 7)).
//   * A stays fp32 in HBM (norms, attention and residual adds read it as such).  A thread loads 8 consecutive floats of a row
//     (two global_load_dwordx4, issued two chunks ahead), splits them in registers (v_cvt_pk_bf16_f32, shift / mask, subtract,
//     v_cvt_pk_bf16_f32: 24 VALU per 8 values, once per element and workgroup) and stores the hi and lo units into the same
//     swizzled 128-byte row image (two ds_write_b128).
//   * 3-slot ring, ONE s_barrier per chunk, counted vmcnt: at step kt a wave waits for the pieces of chunk kt (chunk kt+1's stay
//     in flight), the barrier publishes W(kt) and everybody's A(kt) stores, chunk kt+2 is requested, the 6 MFMAs per
//     (32x32 tile, chunk) run from slot kt % 3, and A(kt+1) is split into slot (kt+1) % 3.
//   * k order: MFMA k-step s (of 2), lane half h takes unit 2s + h of the hi half and unit 4 + 2s + h of the lo half, for A and W
//     alike, so the permutation cancels.
//   * epilogue: the row form of gemm_f32.hip (accumulators through the idle ring, whole 16-byte row pieces out).
// CONV: implicit GEMM of the decoder's 3x3 reflect-pad (+ nearest x2 upsample) convolutions (decoder.py:24-54): the A gather moves
// its source pixel when the TAP changes.
#include "common.h"

namespace {

constexpr int BK = 32;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct X3Args {
    const float* a; const float* w; const float* bias; const float* r1; const float* r2; float* out;      // w: split weights, addressed in 4-byte units
    int M, N, K, lda, ldo, ldr, act;
    int Hin, Win, Cin, Hout, Wout, up, out_nchw, zero_pad;
    int mask_relu;      // r1 is a ReLU output whose sign gates the result: out = r1 > 0 ? acc + r2 : 0 (data-gradient chain of the loss network)
    int out_split;      // the output goes out in the split layout (per row and 32-wide chunk 32 bf16 hi | 32 bf16 lo: what the next GEMM's A operand takes by LDS-DMA); N % 32 == 0
};

__device__ __attribute__((aligned(16))) float g_zero_row_x3[512];          // never written: zero-initialised by the loader

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
template <int V> struct template_int { static constexpr int value = V; };
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// 8 fp32 -> 8 bf16 hi + 8 bf16 lo (round-to-nearest-even both times; x - hi is exact in fp32)
template <int FMT>
__device__ __forceinline__ void split8(const f32x4& x0, const f32x4& x1, u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        unsigned ph, pl, qh, ql;
        split_x2<FMT>(x0[2 * j], x0[2 * j + 1], ph, pl);
        split_x2<FMT>(x1[2 * j], x1[2 * j + 1], qh, ql);
        hi[j] = ph; lo[j] = pl; hi[2 + j] = qh; lo[2 + j] = ql;
    }
}

#ifdef X3_CLOCK_PROBE
#define XPROBE_MARK() do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); probe_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define XPROBE_ADD(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); probe_sec[i] += t_ - probe_last; probe_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define XPROBE_MARK() do {} while (0)
#define XPROBE_ADD(i) do {} while (0)
#endif

// ---- epilogue shared by the two kernels: the row form of gemm_f32.hip (accumulators through the idle ring as the [BM][BN] fp32 tile, whole
// 16-byte row pieces out: coalesced bias / r1 / r2 loads, float4 stores), or - out_split - the same pieces split into bf16 hi | lo and
// stored in the split layout (8 + 8 bytes per piece); element-wise fallback for outputs the row form cannot address (NCHW, N % 4).
template <int BM, int BN, int WM, int WN, bool CONV, int RING_BYTES, int FMT>
__device__ __forceinline__ void x3_epilogue(const X3Args& p, f32x16 (&acc)[WM / 32][WN / 32], unsigned char* ring, int m0, int n0, int wm, int wn) {
    constexpr int NWN = BN / WN;
    constexpr int NWAVES = (BM / WM) * NWN;
    constexpr int NT = NWAVES * 64;
    constexpr int TM = WM / 32, TN = WN / 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    // ---- epilogue (the row form of gemm_f32.hip) ---------------------------------------------------------------------------------------
    const bool vec_ok = (p.N & 3) == 0 && !(CONV && p.out_nchw) && (p.ldo & 3) == 0 && (((uintptr_t)p.out) & 15) == 0 &&
                        (!p.bias || (((uintptr_t)p.bias) & 15) == 0) &&
                        ((!p.r1 && !p.r2) || ((p.ldr & 3) == 0 && ((((uintptr_t)p.r1) | ((uintptr_t)p.r2)) & 15) == 0));
    if (vec_ok) {
        constexpr int VPR = BN / 4;                                       // float4 pieces per tile row
        constexpr int PIECES = BM * VPR / NT;
        static_assert((BM * VPR) % NT == 0, "row pieces must divide over the threads");
        static_assert(BM * BN * 4 <= RING_BYTES, "the output tile must fit in the ring");
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
        asm volatile("" ::: "memory");                                    // nothing that reads the tile (or the LayerNorm statistics) moves above the barrier
        const int c4 = tid % VPR, row0 = tid / VPR;
        int n = n0 + c4 * 4;
        const bool n_ok = n < p.N;
        n = n_ok ? n : p.N - 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *(const f32x4*)(p.bias + n);
        constexpr int BATCH = PIECES < 4 ? PIECES : 4;
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
                    if (p.r2) x += a2[it];
                }
                if (p.out_split) {
                    if (ok[it]) {
                        unsigned h0, h1, l0, l1;
                        split_x2<FMT>(x[0], x[1], h0, l0);
                        split_x2<FMT>(x[2], x[3], h1, l1);
                        // row of N values = N / 32 chunks of 128 B; value n of the row: hi at (n >> 5) * 128 + (n & 31) * 2, lo 64 B further
                        unsigned char* dst = (unsigned char*)(p.out + orow[it] * p.ldo) + (n >> 5) * 128 + (n & 31) * 2;
                        typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
                        *(u32x2_t*)dst = u32x2_t{h0, h1};
                        *(u32x2_t*)(dst + 64) = u32x2_t{l0, l1};
                    }
                } else if (ok[it]) *(f32x4*)(p.out + orow[it] * p.ldo + n) = x;
            }
            __builtin_amdgcn_sched_barrier(0);
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
                if (p.act == MASTER_ACT_GELU) v = gelu_erf(v);
                else if (p.act == MASTER_ACT_RELU) v = relu_np(v);
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

// NSTAGE = 3 (default): the ring described above.  NSTAGE = 2 (r05 experiment, MASTER_X3_TILE=7: 128 x 128 tile, four waves, 64 KB of LDS ->
// TWO workgroups per CU, whose prologues / epilogues / step waits interleave): the weight pieces of chunk kt + 1 go out behind the
// barrier of step kt into the slot that step kt - 1 read (every wave has left it: it passed the barrier) - one step of look-ahead
// instead of two; queue per step [W(kt + 1)] [A(kt + 2)], so the step-head wait leaves the 2 APASS loads of A(kt + 1) in flight and the
// register wait the PER operations of this step.  A is staged through registers two chunks ahead in both forms.
template <int BM, int BN, int WM, int WN, bool CONV, int FMT, int NSTAGE = 3>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, NSTAGE == 2 ? 2 : 1) void gemm_x3_kernel(X3Args p) {
    constexpr int NWN = BN / WN;
    constexpr int NWAVES = (BM / WM) * NWN;
    constexpr int NT = NWAVES * 64;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int RP = NT / 4;                                            // A rows staged per pass (4 threads per 128-byte row)
    constexpr int APASS = BM / RP;
    constexpr int WPIECES = BN / 8;                                       // 1-KB LDS-DMA pieces of the weight tile per chunk
    constexpr int WPW = (WPIECES + NWAVES - 1) / NWAVES;
    constexpr int PER = 2 * APASS + WPW;                                  // vector-memory operations a (DMA-issuing) wave starts per chunk
    static_assert(NSTAGE == 2 || NSTAGE == 3, "two or three ring slots");
    constexpr int STAGE = (BM + BN) * 128;                                // bytes per ring slot: A rows (hi | lo), then W rows
    static_assert(BM % RP == 0 && APASS >= 1, "A rows must divide over the threads");
    static_assert(WPIECES % WPW == 0, "weight pieces must divide over the issuing waves");

    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / NWN) * WM, wn = (wave % NWN) * WN;
    const int r = lane & 31, h = lane >> 5;
    static_assert(WPIECES >= NWAVES, "every wave issues WPW weight pieces per chunk");

    // XCD-aware order (workgroup id % 8 = XCD under round-robin placement).  Linear: the N tiles of an M tile share id % 8 and are adjacent
    // in time, so the A panel is fetched into one L2 once.  Convolution: an XCD takes a CONTIGUOUS eighth of the M tiles, in order - its 32
    // concurrent workgroups then cover ~2 whole images whose rows (each read by 3 vertical taps of up to 3 neighbouring tiles) stay in
    // that XCD's 4-MB L2; with consecutive tiles dealt round-robin over the XCDs every tap re-read came from the Infinity Cache at
    // ~30 GB/s per CU (r04: the 128 -> 128 layer ran at 3 300 clocks per chunk for 1 536 of MFMA with either form of the A path).
    const int MT = (p.M + BM - 1) / BM, NTN = (p.N + BN - 1) / BN;
    const int id = blockIdx.x;
    const int j8 = id >> 3;
    const int per_xcd = (MT + 7) >> 3;
    const int mt = CONV ? (id & 7) * per_xcd + j8 / NTN : (j8 / NTN) * 8 + (id & 7), nt = j8 % NTN;
    if (mt >= MT || (CONV && j8 / NTN >= per_xcd)) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int nk = p.K / BK;

    // ---- A staging: thread = (row arow + i * RP, 8-float unit k8) ------------------------------------------------------------------
    const int arow = tid >> 2, k8 = tid & 3;
    const float* a_src[APASS];
    int cy[CONV ? APASS : 1], cx[CONV ? APASS : 1], cimg[CONV ? APASS : 1];
    int a_dst[APASS];                                                    // byte offset of the thread's hi unit inside a slot
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int row = arow + i * RP;
        a_dst[i] = row * 128 + ((k8 ^ ((row >> 1) & 7)) << 4);            // lo unit: same row, position (4 + k8) ^ sw = hi position ^ 4
        int m = m0 + row; m = m < p.M ? m : p.M - 1;
        if (CONV) {
            const int hw = p.Hout * p.Wout;
            const int b = m / hw, pix = m - b * hw;
            cy[i] = pix / p.Wout; cx[i] = pix - cy[i] * p.Wout; cimg[i] = b * p.Hin * p.Win;
            a_src[i] = p.a;
        } else {
            a_src[i] = p.a + (size_t)m * p.lda + k8 * 8;
        }
    }
    auto set_tap = [&](int tap) __attribute__((always_inline)) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int Hs = p.Hin << p.up, Ws = p.Win << p.up;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const int yy = cy[i] + dy, xx = cx[i] + dx;
            const int sy = reflect1(yy, Hs) >> p.up, sx = reflect1(xx, Ws) >> p.up;
            const bool outside = p.zero_pad && ((unsigned)yy >= (unsigned)Hs || (unsigned)xx >= (unsigned)Ws);
            a_src[i] = outside ? (const float*)g_zero_row_x3 + k8 * 8
                               : p.a + ((size_t)(cimg[i] + sy * p.Win + sx)) * p.Cin + k8 * 8;
        }
    };
    int i_tap = 0, i_c0 = 0;
    if (CONV) set_tap(0);
    // The A loads are inline asm: hipcc's own wait insertion answers a register load that is consumed two loop iterations later (across
    // the back edge, next to LDS-DMA it cannot count) with s_waitcnt vmcnt(0) in front of the NEXT group of loads, which serialises the
    // pipeline.  Written this way the vector-memory queue holds exactly the operations of this source in program order, and the waits
    // below are counted by hand: wait_regs<N>() leaves the N youngest operations in flight and ties the destination registers to the wait.
    auto load_a = [&](int kt, f32x4 (&ra)[APASS][2]) __attribute__((always_inline)) {
        // chunks past the end re-read the last one (the pipeline below requests two chunks ahead in EVERY step, so that the number of
        // operations in flight is the same in every step and each wait is one instruction with a constant count)
        const int ka = CONV ? i_c0 : (kt < nk ? kt : nk - 1) * BK;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            const float* src = a_src[i] + ka;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[i][0]) : "v"(src) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(ra[i][1]) : "v"(src) : "memory");
        }
        if (CONV && !(i_tap == 8 && i_c0 == p.Cin - BK)) {
            i_c0 += BK;
            if (i_c0 == p.Cin) { i_c0 = 0; ++i_tap; set_tap(i_tap); }
        }
    };
    auto store_a = [&](int slot_i, const f32x4 (&ra)[APASS][2]) __attribute__((always_inline)) {
        unsigned char* slot = ring + slot_i * STAGE;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            u32x4 hi, lo;
            split8<FMT>(ra[i][0], ra[i][1], hi, lo);
            *(u32x4*)(slot + a_dst[i]) = hi;
            *(u32x4*)(slot + (a_dst[i] ^ 64)) = lo;
        }
    };

    // ---- W pieces by LDS-DMA: piece j of this wave = tile rows (wave * WPW + j) * 8 + (lane >> 3), LDS unit position lane & 7 ----------
    const int prow = lane >> 3, ppos = lane & 7;
    const float* w_src[WPW];
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        const int row = (wave * WPW + j) * 8 + prow;
        int n = n0 + row; n = n < p.N ? n : p.N - 1;
        w_src[j] = p.w + (size_t)n * p.K + (ppos ^ ((row >> 1) & 7)) * 4;
    }
    auto issue_w = [&](int kt, int slot_i) __attribute__((always_inline)) {
        unsigned char* slot = ring + slot_i * STAGE + BM * 128;
        const int kw = (kt < nk ? kt : nk - 1) * BK;
#pragma unroll
        for (int j = 0; j < WPW; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[j] + kw),
                                             (__attribute__((address_space(3))) void*)(slot + (wave * WPW + j) * 1024), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.0f;

    const int fsw = (r >> 1) & 7;

    auto compute = [&](int slot) __attribute__((always_inline)) {
        const unsigned char* ab = ring + slot * STAGE + (wm + r) * 128;
        const unsigned char* wb = ring + slot * STAGE + BM * 128 + (wn + r) * 128;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ohi = ((2 * s + h) ^ fsw) << 4, olo = ohi ^ 64;
            x3_frag ah[TM], al[TM], wh[TN], wl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *(const x3_frag*)(ab + i * 32 * 128 + ohi);
                al[i] = *(const x3_frag*)(ab + i * 32 * 128 + olo);
            }
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                wh[jn] = *(const x3_frag*)(wb + jn * 32 * 128 + ohi);
                wl[jn] = *(const x3_frag*)(wb + jn * 32 * 128 + olo);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    acc[i][jn] = mfma_x3_32<FMT>(al[i], wh[jn], acc[i][jn]);
                    acc[i][jn] = mfma_x3_32<FMT>(ah[i], wl[jn], acc[i][jn]);
                    acc[i][jn] = mfma_x3_32<FMT>(ah[i], wh[jn], acc[i][jn]);
                }
        }
    };

    // wait until all but the N youngest vector-memory operations of this wave are done; the A registers are in/out operands so that
    // nothing that reads them can be scheduled above the wait (ONE such statement per use: two alternatives behind a branch make
    // hipcc copy the still-loading registers in front of the wait)
    auto wait_regs = [&](f32x4 (&ra)[APASS][2]) __attribute__((always_inline)) {
        constexpr int WCNT = NSTAGE == 3 ? PER + WPW : PER;               // operations issued behind the loads being retired
        // the registers are printed behind the instruction ("; landed v[..] .."): profiles/tools/isa_audit.py reads them from the generated
        // code and checks that no compiler copy / spill touches a register between the asm load that targets it and this wait
        if constexpr (APASS == 1) asm volatile("s_waitcnt vmcnt(%2) ; landed %0 %1" : "+v"(ra[0][0]), "+v"(ra[0][1]) : "n"(WCNT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%4) ; landed %0 %1 %2 %3" : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]) : "n"(WCNT) : "memory");
    };
    static_assert(APASS == 1 || APASS == 2, "wait_regs covers one or two A passes");

    // queue order: [A(0)] [W(0)] [A(1)] [W(1)] | step kt: [A(kt+2)] [W(kt+2)] - 2 APASS loads and WPW pieces per chunk
    f32x4 ra0[APASS][2], ra1[APASS][2];
    load_a(0, ra0); issue_w(0, 0);
    load_a(1, ra1); if constexpr (NSTAGE == 3) issue_w(1, 1);
    wait_regs(ra0);                                                        // A(0): W(0) and the group of chunk 1 stay in flight
    store_a(0, ra0);

    int s_cur = 0, s_nxt = 1, s_nn = 2;                                    // ring slots of chunks kt, kt + 1, kt + 2
#ifdef X3_CLOCK_PROBE     // diagnostic build only (profiles/probes/gemm_x3_probe.hip): s_memtime sections of every wave 0, written to a buffer of their own
    unsigned long long probe_sec[6] = {0, 0, 0, 0, 0, 0}, probe_last = 0;            // vmcnt wait, barrier, issue, compute, A wait, split + store
    const unsigned long long probe_t0 = __builtin_amdgcn_s_memtime();
#endif
    auto body = [&](int kt, f32x4 (&nxt)[APASS][2], f32x4 (&fre)[APASS][2]) __attribute__((always_inline)) {
        XPROBE_MARK();
        wait_vmcnt<NSTAGE == 3 ? PER : 2 * APASS>();                       // W(kt) has landed: only the group of chunk kt + 1 (two slots: the loads of A(kt + 1)) is still in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // this wave's A(kt) stores
        XPROBE_ADD(0);
        __builtin_amdgcn_s_barrier();
        XPROBE_ADD(1);
        if constexpr (NSTAGE == 3) { load_a(kt + 2, fre); issue_w(kt + 2, s_nn); }
        else { issue_w(kt + 1, s_nxt); load_a(kt + 2, fre); }
        XPROBE_ADD(2);
        compute(s_cur);
        XPROBE_ADD(3);
        wait_regs(nxt);                                                    // A(kt + 1): W(kt + 1) and the group of chunk kt + 2 stay in flight
        XPROBE_ADD(4);
        store_a(s_nxt, nxt);
        XPROBE_ADD(5);
        asm volatile("" ::: "memory");
        if constexpr (NSTAGE == 3) { const int t = s_cur; s_cur = s_nxt; s_nxt = s_nn; s_nn = t; }
        else { const int t = s_cur; s_cur = s_nxt; s_nxt = t; }
    };
    for (int kt = 0; kt < nk; kt += 2) {
        body(kt, ra1, ra0);
        if (kt + 1 >= nk) break;          // (a break, not a guarded second half: the flow graph then has no path from a skipped half back to the loop head, which the ISA audit would have to assume)
        body(kt + 1, ra0, ra1);
    }
    // the two look-ahead groups past the end and this wave's last (look-ahead) A stores must have landed before the barrier below lets
    // other waves write the output tile over the ring (a store still in flight would land on top of it)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef X3_CLOCK_PROBE
    if (tid == 0 && g_probe) {
        unsigned long long* dst = g_probe + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 6; ++i) dst[i] = probe_sec[i];
        dst[6] = __builtin_amdgcn_s_memtime() - probe_t0;
    }
#endif

    x3_epilogue<BM, BN, WM, WN, CONV, NSTAGE * STAGE, FMT>(p, acc, ring, m0, n0, wm, wn);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same product with the A operand ALREADY in the split layout (written by the previous kernel's out_split epilogue): both tiles
// reach LDS by LDS-DMA, nothing is staged through registers, no split arithmetic and no ds_write in the loop.  In-kernel clocks of the
// fp32-input form (profiles/r04_gemm_x3_probe_mi355x.txt, 128 -> 128 convolution at 64^2): per chunk and wave 525 clocks to issue the
// look-ahead (4 loads, tap arithmetic, 2 pieces), 856 to split and store 16 values per thread (the 32 ds_write_b128 of the eight waves
// queue on the LDS store path) next to 768 of MFMA - with two waves per SIMD in lockstep the vector work, not the matrix core, bounded it.
// Ring, k order, waits and epilogue as above; per chunk a wave issues BM / 8 / NWAVES pieces of A (8 rows x 128 B, per-lane row
// pointers: the convolution gather moves them when the tap changes) and BN / 8 / NWAVES of W.
// ---------------------------------------------------------------------------------------------------------------------------------
// NSTAGE = 2 (r05): two ring slots, the pieces of chunk kt + 1 dealt between the MFMA groups of step kt into the slot step kt - 1 read, all of
// them landed (vmcnt(0)) at the head of step kt + 1: half the look-ahead, but a 128 x 128 tile then takes 64 KB and two workgroups share a CU.
template <int BM, int BN, int WM, int WN, bool CONV, int FMT, int NSTAGE = 3>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, NSTAGE == 2 ? 2 : 1) void gemm_x3s_kernel(X3Args p) {
    constexpr int NWN = BN / WN;
    constexpr int NWAVES = (BM / WM) * NWN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int APW = BM / 8 / NWAVES, WPW = BN / 8 / NWAVES;          // 1-KB pieces per wave and chunk
    constexpr int PER = APW + WPW;
    static_assert(NSTAGE == 2 || NSTAGE == 3, "two or three ring slots");
    constexpr int STAGE = (BM + BN) * 128;
    static_assert((BM / 8) % NWAVES == 0 && (BN / 8) % NWAVES == 0 && WPW >= 1, "pieces must divide over the waves");

    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / NWN) * WM, wn = (wave % NWN) * WN;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware order (workgroup id % 8 = XCD under round-robin placement).  Linear: the N tiles of an M tile share id % 8 and are adjacent
    // in time, so the A panel is fetched into one L2 once.  Convolution: an XCD takes a CONTIGUOUS eighth of the M tiles, in order - its 32
    // concurrent workgroups then cover ~2 whole images whose rows (each read by 3 vertical taps of up to 3 neighbouring tiles) stay in
    // that XCD's 4-MB L2; with consecutive tiles dealt round-robin over the XCDs every tap re-read came from the Infinity Cache at
    // ~30 GB/s per CU (r04: the 128 -> 128 layer ran at 3 300 clocks per chunk for 1 536 of MFMA with either form of the A path).
    const int MT = (p.M + BM - 1) / BM, NTN = (p.N + BN - 1) / BN;
    const int id = blockIdx.x;
    const int j8 = id >> 3;
    const int per_xcd = (MT + 7) >> 3;
    const int mt = CONV ? (id & 7) * per_xcd + j8 / NTN : (j8 / NTN) * 8 + (id & 7), nt = j8 % NTN;
    if (mt >= MT || (CONV && j8 / NTN >= per_xcd)) return;
    const int m0 = mt * BM, n0 = nt * BN;
    const int nk = p.K / BK;

    const int prow = lane >> 3, ppos = lane & 7;
    const float* a_src[APW];
    int cy[CONV ? APW : 1], cx[CONV ? APW : 1], cimg[CONV ? APW : 1], csw[CONV ? APW : 1];
#pragma unroll
    for (int j = 0; j < APW; ++j) {
        const int row = (wave * APW + j) * 8 + prow;
        const int sw = (ppos ^ ((row >> 1) & 7)) * 4;                  // 4-byte offset of the lane's source unit within the 128-byte chunk
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
    auto set_tap = [&](int tap) __attribute__((always_inline)) {
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
        const int Hs = p.Hin << p.up, Ws = p.Win << p.up;
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            const int yy = cy[j] + dy, xx = cx[j] + dx;
            const int sy = reflect1(yy, Hs) >> p.up, sx = reflect1(xx, Ws) >> p.up;
            const bool outside = p.zero_pad && ((unsigned)yy >= (unsigned)Hs || (unsigned)xx >= (unsigned)Ws);
            a_src[j] = outside ? (const float*)g_zero_row_x3 + csw[j]
                               : p.a + ((size_t)(cimg[j] + sy * p.Win + sx)) * p.Cin + csw[j];
        }
    };
    int i_tap = 0, i_c0 = 0;
    if (CONV) set_tap(0);
#ifdef X3_ABLATE_SAMEDATA
#pragma unroll
    for (int j = 0; j < APW; ++j) a_src[j] = p.a + (size_t)((wave * APW + j) * 8 + prow) * 32 + (ppos ^ ((((wave * APW + j) * 8 + prow) >> 1) & 7)) * 4;
#endif
    // One look-ahead chunk = APW pieces of A and WPW of W per wave.  They are NOT issued as a burst behind the barrier (every wave of the
    // CU then sits in the memory-issue queue at the same time, ~150 clocks per piece, with the matrix cores idle - measured on the MLP
    // kernel: 72 k of 264 k clocks) but one at a time between the MFMA groups of the chunk being computed: begin_issue() fixes the chunk's
    // slot and k offset, piece(q) issues piece q = 0 .. PER - 1 (A pieces first), end_issue() advances the convolution's tap state.
    unsigned char* is_slot = nullptr; int is_kc = 0, is_ka = 0;
    auto begin_issue = [&](int kt, int slot_i) __attribute__((always_inline)) {       // chunks past the end re-read the last one (uniform vmcnt bookkeeping)
        is_slot = ring + slot_i * STAGE;
        is_kc = (kt < nk ? kt : nk - 1) * BK;
        is_ka = CONV ? i_c0 : is_kc;
#ifdef X3_ABLATE_SAMEDATA       // probe builds: every chunk of every workgroup re-reads the same 48 KB (L2-resident): what is left is the CU's own time
        is_kc = 0; is_ka = 0;
#endif
    };
    auto piece = [&](int q) __attribute__((always_inline)) {
        if (q < APW)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[q] + is_ka),
                                             (__attribute__((address_space(3))) void*)(is_slot + (wave * APW + q) * 1024), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_src[q - APW] + is_kc),
                                             (__attribute__((address_space(3))) void*)(is_slot + BM * 128 + (wave * WPW + q - APW) * 1024), 16, 0, 0);
    };
    auto end_issue = [&]() __attribute__((always_inline)) {
#ifdef X3_ABLATE_SAMEDATA
        return;
#endif
        if (CONV && !(i_tap == 8 && i_c0 == p.Cin - BK)) {
            i_c0 += BK;
            if (i_c0 == p.Cin) { i_c0 = 0; ++i_tap; set_tap(i_tap); }
        }
    };
    auto issue = [&](int kt, int slot_i) __attribute__((always_inline)) {
        begin_issue(kt, slot_i);
#pragma unroll
        for (int q = 0; q < PER; ++q) piece(q);
        end_issue();
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jn][e] = 0.0f;

    const int fsw = (r >> 1) & 7;
    issue(0, 0); if constexpr (NSTAGE == 3) issue(1, 1);
    int s_cur = 0, s_nxt = 1, s_nn = 2;
    constexpr int NG = 2 * TM * TN;                                        // MFMA groups (3 MFMAs on one accumulator tile) per chunk
#ifdef X3_CLOCK_PROBE
    unsigned long long probe_sec[6] = {0, 0, 0, 0, 0, 0}, probe_last = 0;            // vmcnt wait, barrier, -, compute + pieces, -, -
    const unsigned long long probe_t0 = __builtin_amdgcn_s_memtime();
#endif
    for (int kt = 0; kt < nk; ++kt) {
        XPROBE_MARK();
        wait_vmcnt<NSTAGE == 3 ? PER : 0>();                               // chunk kt has landed (this wave's pieces): only chunk kt + 1's are in flight (two slots: nothing is)
        XPROBE_ADD(0);
        __builtin_amdgcn_s_barrier();
        XPROBE_ADD(1);
        if constexpr (NSTAGE == 3) begin_issue(kt + 2, s_nn); else begin_issue(kt + 1, s_nxt);
        const unsigned char* ab = ring + s_cur * STAGE + (wm + r) * 128;
        const unsigned char* wb = ring + s_cur * STAGE + BM * 128 + (wn + r) * 128;
        int g = 0, q = 0;                                                  // compile-time after unrolling
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int ohi = ((2 * s + h) ^ fsw) << 4, olo = ohi ^ 64;
            x3_frag ah[TM], al[TM], wh[TN], wl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *(const x3_frag*)(ab + i * 32 * 128 + ohi);
                al[i] = *(const x3_frag*)(ab + i * 32 * 128 + olo);
            }
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
                wh[jn] = *(const x3_frag*)(wb + jn * 32 * 128 + ohi);
                wl[jn] = *(const x3_frag*)(wb + jn * 32 * 128 + olo);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn) {
                    acc[i][jn] = mfma_x3_32<FMT>(al[i], wh[jn], acc[i][jn]);
                    acc[i][jn] = mfma_x3_32<FMT>(ah[i], wl[jn], acc[i][jn]);
                    acc[i][jn] = mfma_x3_32<FMT>(ah[i], wh[jn], acc[i][jn]);
                    // pieces due after this group: piece q goes behind group floor(q * NG / PER)
                    ++g;
#pragma unroll
                    for (int z = 0; z < PER; ++z)
                        if (q < PER && (q * NG) / PER < g) { __builtin_amdgcn_sched_barrier(0); piece(q); __builtin_amdgcn_sched_barrier(0); ++q; }
                }
        }
        end_issue();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // this wave's fragment reads of slot s_cur are done before it passes the next barrier
        XPROBE_ADD(3);
        if constexpr (NSTAGE == 3) { const int t = s_cur; s_cur = s_nxt; s_nxt = s_nn; s_nn = t; }
        else { const int t = s_cur; s_cur = s_nxt; s_nxt = t; }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");            // the two look-ahead groups past the end: the ring is reused below
#ifdef X3_CLOCK_PROBE
    if (tid == 0 && g_probe) {
        unsigned long long* dst = g_probe + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 6; ++i) dst[i] = probe_sec[i];
        dst[6] = __builtin_amdgcn_s_memtime() - probe_t0;
    }
#endif
    x3_epilogue<BM, BN, WM, WN, CONV, NSTAGE * STAGE, FMT>(p, acc, ring, m0, n0, wm, wn);
}

template <int BM, int BN, int WM, int WN, bool CONV, int FMT, int NSTAGE = 3>
int launch_s(const X3Args& p, hipStream_t s) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    const int MT = (p.M + BM - 1) / BM, NTN = (p.N + BN - 1) / BN;
    const int grid = ((MT + 7) / 8) * 8 * NTN;
    const size_t lds = (size_t)NSTAGE * (BM + BN) * 128;
    static LdsLimit lds_limit;
    lds_limit.ensure((const void*)gemm_x3s_kernel<BM, BN, WM, WN, CONV, FMT, NSTAGE>, (int)lds);
    hipLaunchKernelGGL((gemm_x3s_kernel<BM, BN, WM, WN, CONV, FMT, NSTAGE>), dim3(grid), dim3(NT), lds, s, p);
    return master_check_launch("gemm_x3s");
}

template <int BM, int BN, int WM, int WN, bool CONV, int FMT, int NSTAGE = 3>
int launch(const X3Args& p, hipStream_t s) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    const int MT = (p.M + BM - 1) / BM, NTN = (p.N + BN - 1) / BN;
    const int grid = ((MT + 7) / 8) * 8 * NTN;
    const size_t lds = (size_t)NSTAGE * (BM + BN) * 128;
    static LdsLimit lds_limit;
    lds_limit.ensure((const void*)gemm_x3_kernel<BM, BN, WM, WN, CONV, FMT, NSTAGE>, (int)lds);
    hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, WM, WN, CONV, FMT, NSTAGE>), dim3(grid), dim3(NT), lds, s, p);
    return master_check_launch("gemm_x3");
}

// MASTER_X3_TILE: 0 = by shape (default), 1 = 128x128 / 4 waves, 2 = 128x128 / 8 waves, 3 = 256x128 / 8 waves, 5 / 6 = 128x64 with 4 / 8 waves,
// 7 = 128x128 / 4 waves on the two-slot ring, 8 = the by-shape choice of r04 (no two-slot form), 9 = the shipped choice + two-slot forms of the
// split-input kernel (measured: no gain, profiles/r05_x3_ring2_ab.txt section 5)  (A/B runs)
static int x3_tile() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MASTER_X3_TILE"); v = e ? atoi(e) : 0; }
    return v;
}

template <bool CONV, int FMT>
int dispatch(const X3Args& p, hipStream_t s) {
    const int force = x3_tile();
    if (p.N > 64) {
        if (force == 1) return launch<128, 128, 64, 64, CONV, FMT>(p, s);
        if (force == 2) return launch<128, 128, 32, 64, CONV, FMT>(p, s);
        if (force == 3) return launch<256, 128, 64, 64, CONV, FMT>(p, s);
        if (force == 5) return launch<128, 64, 64, 32, CONV, FMT>(p, s);        // 72 KB of LDS, four waves: two workgroups per CU
        if (force == 6) return launch<128, 64, 32, 32, CONV, FMT>(p, s);        // eight waves
        if (force == 7) return launch<128, 128, 64, 64, CONV, FMT, 2>(p, s);    // r05: two-slot ring, 64 KB, four waves: two workgroups per CU
        // one N tile of 128 with a short K (the stage-1 projection, 262 144 x 128 x 128 + residual): 128 x 64 tiles, eight waves, two
        // workgroups per CU - one tile's epilogue runs under the other's loop (117.6 -> 100.3 us; with more N tiles the re-split of A
        // per N tile costs more than the overlap gives: profiles/r04 A/B of MASTER_X3_TILE=5/6)
        if (!CONV && (force == 0 || force >= 8) && p.N <= 128 && p.K <= 128 && p.M >= 65536) return launch<128, 64, 32, 32, CONV, FMT>(p, s);
        // fewer than two 256-row tiles per CU: 128-row tiles (eight waves: two per SIMD) fill the chip better
        const long long t256 = (long long)((p.M + 255) / 256) * ((p.N + 127) / 128);
        // every other Linear layer: the two-slot ring, two workgroups per CU.  Same-box kernel traces of the bs = 32 forward (r05,
        // profiles/r05_x3_ring2_ab.txt), three-slot form by shape -> this one: 32 768 x 256 x 256 (9 launches) 31.8 -> 28.3 us, 65 536 x 256 x K
        // 62.5 -> 58.8, 65 536 x 768 x 256 120 -> 114.5, 262 144 x 384 x 128 171 -> 168; the convolutions (+8 %) and the stage-1 projection
        // above (+10 %) lose with it and keep the three-slot forms
        if (!CONV && (force == 0 || force == 9)) return launch<128, 128, 64, 64, CONV, FMT, 2>(p, s);
        if (t256 < 512) return launch<128, 128, 32, 64, CONV, FMT>(p, s);
        return launch<256, 128, 64, 64, CONV, FMT>(p, s);
    }
    if (p.N > 32 || force == 4) return launch<256, 64, 64, 32, CONV, FMT>(p, s);
    return launch<128, 32, 32, 32, CONV, FMT>(p, s);
}

template <bool CONV, int FMT>
int dispatch_s(const X3Args& p, hipStream_t s) {
    const int force = x3_tile();
    if (p.N > 64) {
        if (force == 9) return launch_s<128, 128, 64, 64, CONV, FMT, 2>(p, s);     // r05 A/B: two-slot ring, two workgroups per CU
        const long long t256 = (long long)((p.M + 255) / 256) * ((p.N + 127) / 128);
        if (t256 < 512) return launch_s<128, 128, 32, 64, CONV, FMT>(p, s);
        return launch_s<256, 128, 64, 64, CONV, FMT>(p, s);
    }
    if (p.N > 32 && force == 9) return launch_s<128, 64, 64, 32, CONV, FMT, 2>(p, s);   // 48 KB: three workgroups per CU by LDS, two by registers
    if (p.N > 32) return launch_s<256, 64, 64, 32, CONV, FMT>(p, s);
    return launch_s<128, 32, 32, 32, CONV, FMT>(p, s);
}

// ---- weight split: fp32 [rows, K] -> per row and 32-wide chunk: 32 bf16 hi, 32 bf16 lo (same byte size and row stride as the fp32 matrix) ----
// layout 1 (the second Linear of the fused MLP, mlp_x3.hip): inside every group of 16 k the order is 0-3, 8-11, 4-7, 12-15 - the k order
// of a packed 32x32 accumulator tile used as an MFMA operand (lane half h holds rows 4h + 0..3 and 8 + 4h + 0..3 of each group).
template <int FMT>
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, unsigned* __restrict__ out, size_t n8, int layout) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;            // one 8-element unit of the OUTPUT per thread
    if (i >= n8) return;
    const size_t chunk = i >> 2, u = i & 3;                               // 4 units per 32-wide chunk
    const float* src = layout == 1 ? w + chunk * 32 + (u >> 1) * 16 + (u & 1) * 4 : w + i * 8;
    const f32x4 x0 = *(const f32x4*)src, x1 = *(const f32x4*)(src + (layout == 1 ? 8 : 4));
    u32x4 hi, lo;
    split8<FMT>(x0, x1, hi, lo);
    *(u32x4*)(out + chunk * 32 + u * 4) = hi;
    *(u32x4*)(out + chunk * 32 + 16 + u * 4) = lo;
}

}  // namespace

// fmt: X3_BF16 (MASTER_F32X3) or X3_F16 (MASTER_F16X3) halves, for the weights, the A split in registers and the out_split epilogue alike
int gemm_x3_split_weights(const float* w, void* out, size_t rows, size_t K, int layout, hipStream_t s, int fmt) {
    const size_t n8 = rows * K / 8;
    if (fmt == X3_F16) hipLaunchKernelGGL(split_weights_kernel<X3_F16>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, w, (unsigned*)out, n8, layout);
    else hipLaunchKernelGGL(split_weights_kernel<X3_BF16>, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, w, (unsigned*)out, n8, layout);
    return master_check_launch("split_weights");
}

int gemm_x3_linear(const float* a, const void* w, const float* bias, const float* r1, const float* r2, float* out,
                   int M, int N, int K, int lda, int ldo, int ldr, int act, hipStream_t s, int a_split, int out_split, int fmt) {
    X3Args p{a, (const float*)w, bias, r1, r2, out, M, N, K, lda, ldo, ldr, act, 0, 0, 0, 1, 1, 0, 0, 0, 0, out_split};
    if (fmt == X3_F16) return a_split ? dispatch_s<false, X3_F16>(p, s) : dispatch<false, X3_F16>(p, s);
    return a_split ? dispatch_s<false, X3_BF16>(p, s) : dispatch<false, X3_BF16>(p, s);
}

int gemm_x3_conv3x3(const float* x, const void* w, const float* bias, float* out, int B, int Hin, int Win, int Cin,
                    int Cout, int up, int relu, int out_nchw, hipStream_t s, int zero_pad, int a_split, int out_split, int fmt) {
    const int Hout = Hin << up, Wout = Win << up;
    X3Args p{x, (const float*)w, bias, nullptr, nullptr, out, B * Hout * Wout, Cout, 9 * Cin, 0, Cout, 0,
             relu ? MASTER_ACT_RELU : MASTER_ACT_NONE, Hin, Win, Cin, Hout, Wout, up, out_nchw, zero_pad, 0, out_split};
    if (fmt == X3_F16) return a_split ? dispatch_s<true, X3_F16>(p, s) : dispatch<true, X3_F16>(p, s);
    return a_split ? dispatch_s<true, X3_BF16>(p, s) : dispatch<true, X3_BF16>(p, s);
}

// data gradient of a zero-padded 3x3 convolution (the frozen loss network's chain, gemm_f32.hip's gemm_f32_conv3x3_dgrad in split-bf16):
// g [B,H,W,Cg] fp32, wf = split [Cx][3][3][Cg] flipped weights; out = relu_src > 0 ? conv(g, wf) + add : 0  (relu_src / add optional).
// bf16 halves only: gradients span fp32's exponent range (1e-9 ... 1), which half halves cannot represent.
int gemm_x3_conv3x3_dgrad(const float* g, const void* wf, float* out, const float* relu_src, const float* add, int B, int H, int W,
                          int Cg, int Cx, hipStream_t s) {
    X3Args p{g, (const float*)wf, nullptr, relu_src, add, out, B * H * W, Cx, 9 * Cg, 0, Cx, Cx, MASTER_ACT_NONE, H, W, Cg, H, W, 0, 0, 1,
             relu_src != nullptr, 0};
    return dispatch<true, X3_BF16>(p, s);
}
