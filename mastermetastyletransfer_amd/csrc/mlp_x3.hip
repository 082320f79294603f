// Fused MLP block of the split-bf16 fast parity mode (MASTER_F32X3):   out = res + W2 . gelu(W1 . LN?(x) + b1) + b2
// (torchvision MLP = Linear, GELU (exact erf), Linear and the residual add around it, optional LayerNorm on x: reference
// style_transformer.py:366,:392,:839-841,:865-882,:991,:1125 and the Swin blocks).  x, res, out: fp32 token matrices in HBM; the
// hidden activations [M, 4C] - 537 MB written and read again per Swin stage-1 block at bs = 32 in the two-GEMM form - never leave the CU.
//
// The shape of mlp_bf16.hip (token-stationary, everything transposed so that the token sits on the MFMA lane), with every product
// as three bf16 MFMAs on split operands (gemm_x3.hip: x = hi + lo, a*w = a_lo*w_hi + a_hi*w_lo + a_hi*w_hi, fp32 accumulation):
//   * a wave owns 32 tokens for the whole kernel.  Their rows are loaded once, LayerNorm'd in fp32, split, and stay in registers as
//     the B operands xh / xl (lane = token, half h holds k = 16s + 8h .. +8 of k-step s).
//   * hidden units go in blocks of 64:  phase 1  H^T[j][m] = b1[j] + sum_k W1[j][k] X[m][k]   (A = W1 rows from LDS, two 32x32 tiles),
//     exact-erf GELU on the accumulators, split into hi / lo and packed pairwise: a packed accumulator tile IS a B fragment whose k
//     order inside a group of 16 hidden units is (4h + 0..3, 8 + 4h + 0..3) - W2 is stored in that order by master_split_weights
//     (layout 1), so no lane exchange is needed;  phase 2  O^T[n][m] += sum_j W2[n][j] H^T[j][m]  (A = W2 rows from LDS, C / 32 tiles).
//   * weights: pre-split (hi | lo per 32-wide k chunk = 128-byte row segments), streamed in 32-KB steps through a 4-slot LDS ring by
//     LDS-DMA three steps ahead, counted vmcnt + ONE s_barrier per step (48 MFMAs per wave and step); images of [rows][128 B] with
//     source-swizzled 16-byte units (unit u of row R at position u ^ ((R >> 1) & 7)) as in gemm_x3.hip: conflict-free ds_read_b128.
//   * epilogue: each wave transposes its [32 tokens][C] fp32 tile through the idle ring and walks it row by row: bias, residual
//     (re-read from memory: exact) and stores are whole rows.
// C = 256: 4 waves (one per SIMD, ~450 registers), 128 tokens per workgroup.  C = 128: 8 waves (two per SIMD), 256 tokens.
#include "common.h"
#include <stdlib.h>

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct Frag { u32x4 u; };        // one MFMA operand fragment: 8 sixteen-bit halves (bf16 or IEEE half, by FMT)

struct MlpX3Args {
    const float* x; const float* ln_g; const float* ln_b;
    const unsigned char* w1; const float* b1; const unsigned char* w2; const float* b2;       // w1 [HID][C], w2 [C][HID]: split weights (w2: layout 1)
    const float* res; float* out;
    int M, HID, ldx, ldr, ldo; float eps;
};

constexpr int SLOT = 32768;      // bytes per ring slot = one step of weights
constexpr int HB = 64;           // hidden units per block

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// PIPE (C = 256, one wave per SIMD, 512 registers): fragment reads pipelined by hand, the GELU of block hb dealt out between the MFMAs of
// phase 1 of block hb + 1 (a second accumulator set).  !PIPE (C = 128, two waves per SIMD, 256 registers: no room for either): the
// compiler schedules the reads, the GELU runs between the two phases, and the partner wave's MFMAs cover what that exposes.
// RING slots of 32 KB, AHEAD = RING - 1 steps of DMA in flight.  C = 256: 4 slots, one workgroup per CU.  C = 128 (r04, second form): TWO
// slots, four waves, 72 KB of LDS and 200 registers: two independent 128-token workgroups per CU (WGPC = 2) - nothing synchronises them, so
// one's GELU burst, prologue (row loads, LayerNorm, split) and epilogue (transpose, residual, stores) run under the other's MFMAs,
// which eight waves of ONE workgroup in lockstep at every step barrier could not do (probe: 28.7 k of 126 k clocks in the GELU with the
// matrix cores idle, 27 k in prologue + epilogue).
template <int C, int NW, bool PIPE, int RING, int WGPC, int FMT>
__global__ __launch_bounds__(NW * 64, WGPC) void mlp_x3_kernel(MlpX3Args p) {
    constexpr int AHEAD = RING - 1;             // steps of DMA in flight ahead of the one being consumed
    constexpr int KS1 = C / 16;                 // k-steps of phase 1
    constexpr int NT2 = C / 32;                 // output-channel tiles of phase 2
    constexpr int P1 = C / 128;                 // phase-1 steps per hidden block: [64 rows][4 chunks] each
    constexpr int P2 = C / 128;                 // phase-2 steps per hidden block: C = 256: [256 rows][1 chunk] x 2;  C = 128: [128 rows][2 chunks]
    constexpr int SPH = P1 + P2;
    constexpr int PPW = 32 / NW;                // 1-KB DMA pieces per wave and step
    constexpr int R2 = C == 256 ? 256 : 128;    // rows of a phase-2 image
    static_assert(C == 128 || C == 256, "channel widths of swin_B");

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [RING][SLOT] | b1 [HID] | b2 [C] | ln gamma, beta [C] | dummy 1 KB per wave
    float* b1s = (float*)(lds + RING * SLOT);
    float* b2s = b1s + p.HID;
    float* lng = b2s + C;
    float* lnb = lng + C;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MLPX3_CLOCK_PROBE   // diagnostic build only (profiles/probes/mlp_x3_probe.hip): s_memtime sections of wave 0, written to a buffer of their own
    const unsigned long long probe_t0 = __builtin_amdgcn_s_memtime(), probe_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long probe_sec[7] = {0, 0, 0, 0, 0, 0, 0}, probe_last = 0;     // wait+barrier, DMA issue, phase 1, GELU, phase 2, prologue, epilogue
#define PROBE_MARK() do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); probe_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define PROBE_ADD(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); probe_sec[i] += t_ - probe_last; probe_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PROBE_MARK() do {} while (0)
#define PROBE_ADD(i) do {} while (0)
#endif
    const int r = lane & 31, h = lane >> 5;
    const int m = blockIdx.x * (NW * 32) + wave * 32 + r;
    const int mc = m < p.M ? m : p.M - 1;
    const int nhb = p.HID / HB;
    const int T = nhb * SPH;

    // ---- LDS-DMA of stream step t into ring slot t % RING -----------------------------------------------------------------------
    // piece q = wave * PPW + i of a step: phase 1: image b = q >> 3 (k chunk), rows 8 (q & 7) + (lane >> 3) of the block's 64;
    // phase 2: C = 256: rows 8 q + (lane >> 3) of 256;  C = 128: image b = q >> 4, rows 8 (q & 15) + (lane >> 3) of 128.
    const int prow = lane >> 3, ppos = lane & 7;
    unsigned vo1[PPW], vo2[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wave * PPW + i;
        const int row1 = 8 * (q & 7) + prow;
        vo1[i] = (unsigned)(row1 * C * 4 + (q >> 3) * 128 + ((ppos ^ ((row1 >> 1) & 7)) << 4));
        const int row2 = C == 256 ? 8 * q + prow : 8 * (q & 15) + prow;
        vo2[i] = (unsigned)(row2 * p.HID * 4 + (C == 256 ? 0 : (q >> 4) * 128) + ((ppos ^ ((row2 >> 1) & 7)) << 4));
    }
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds;
    const unsigned dummy_lds = lds0 + RING * SLOT + (p.HID + 3 * C) * 4 + wave * 1024;
    // Stream order (as mlp_bf16.hip): W1(0) | W1(1) W2(0) | W1(2) W2(1) | ... | W2(nhb-1): phase 1 of block hb + 1 runs BEFORE phase 2 of
    // block hb, so that the GELU of block hb can be dealt out between its MFMAs.  set_dma_step() works out the wave-uniform source and
    // LDS address of step t once; dma_piece(i) issues ONE 1-KB piece - the pieces of step t + AHEAD go out one at a time between the MFMA
    // groups of step t (a burst of 8 per wave at the top of a step kept the waves in the memory-issue queue for ~1 100 clocks per step:
    // 72 k of 264 k clocks per workgroup, profiles/r04_mlp_x3_probe_mi355x.txt).  Steps past the end are not skipped (no branch, the
    // vmcnt bookkeeping stays uniform): they re-read the first KB of W1 into a 1-KB dummy region per wave.
    const unsigned char* d_base = nullptr; unsigned d_dst = 0, d_dstep = 0; bool d_w1 = true, d_valid = false;
    auto set_dma_step = [&](int t) __attribute__((always_inline)) {
        d_valid = t < T;
        int hb = 0, idx = t; d_w1 = true;
        if constexpr (!PIPE) {                                  // sequential order: W1(0) W2(0) | W1(1) W2(1) | ...
            hb = t / SPH; const int o = t - hb * SPH;
            d_w1 = o < P1; idx = d_w1 ? o : o - P1;
        } else if (t >= P1) {
            const int u = t - P1, bq = u / SPH, o = u - bq * SPH;
            if (bq < nhb - 1 && o < P1) { hb = bq + 1; idx = o; }
            else { d_w1 = false; hb = bq; idx = bq < nhb - 1 ? o - P1 : o; }
        }
        if (!d_valid) { d_w1 = true; hb = 0; idx = 0; }
        d_base = d_w1 ? p.w1 + ((size_t)hb * HB * C + idx * 128) * 4 : p.w2 + (size_t)(2 * hb + (C == 256 ? idx : 0)) * 128;
        d_dst = d_valid ? lds0 + (t % RING) * SLOT + wave * PPW * 1024 : dummy_lds;
        d_dstep = d_valid ? 1024u : 0u;
    };
    auto dma_piece = [&](int i) __attribute__((always_inline)) {
        const unsigned vo = !d_valid ? (unsigned)(lane * 16) : (d_w1 ? vo1[i] : vo2[i]);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(d_base + vo),
                                         (__attribute__((address_space(3))) void*)(size_t)(d_dst + i * d_dstep), 16, 0, 0);
    };
    auto issue = [&](int t) __attribute__((always_inline)) {
        set_dma_step(t);
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma_piece(i);
    };

    // ---- X rows -> registers (fp32), requested before the first weight steps ------------------------------------------------------
    f32x4 xv[KS1][2];
    {
        const float* xp = p.x + (size_t)mc * p.ldx + 8 * h;
#pragma unroll
        for (int s = 0; s < KS1; ++s) { xv[s][0] = *(const f32x4*)(xp + 16 * s); xv[s][1] = *(const f32x4*)(xp + 16 * s + 4); }
    }
    for (int i = tid; i < p.HID; i += NW * 64) b1s[i] = p.b1[i];
    if (tid < C) {
        b2s[tid] = p.b2[tid];
        if (p.ln_g) { lng[tid] = p.ln_g[tid]; lnb[tid] = p.ln_b[tid]; }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // parameter vectors visible to every wave; nothing of this wave is in flight
    int t_issue = 0;
    for (; t_issue < AHEAD; ++t_issue) issue(t_issue);

    Frag xh[KS1], xl[KS1];
    {
        if (p.ln_g) {                                     // LayerNorm over the C channels of token m: this lane holds half of them, lane ^ 32 the rest
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < KS1; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) sum += xv[s][0][e] + xv[s][1][e];
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum / (float)C;
            float sq = 0.f;
#pragma unroll
            for (int s = 0; s < KS1; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d0 = xv[s][0][e] - mean, d1 = xv[s][1][e] - mean;
                    sq += d0 * d0 + d1 * d1;
                }
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = rsqrtf(sq / (float)C + p.eps);
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                const float* gp = lng + 16 * s + 8 * h; const float* bp = lnb + 16 * s + 8 * h;
                const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 4);
                const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xv[s][0][e] = (xv[s][0][e] - mean) * rstd * g0[e] + b0[e];
                    xv[s][1][e] = (xv[s][1][e] - mean) * rstd * g1[e] + b1[e];
                }
            }
        }
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                unsigned ah, al, bh, bl;
                split_x2<FMT>(xv[s][0][2 * j], xv[s][0][2 * j + 1], ah, al);
                split_x2<FMT>(xv[s][1][2 * j], xv[s][1][2 * j + 1], bh, bl);
                xh[s].u[j] = ah; xl[s].u[j] = al; xh[s].u[2 + j] = bh; xl[s].u[2 + j] = bl;
            }
        }
    }

    // ---- per-step synchronisation: my pieces of step t have landed -> barrier -> everyone's have, and everyone is done reading step
    //      t - 1, whose slot is the one step t + AHEAD goes into ---------------------------------------------------------------------
    int t_cons = 0;
    auto begin_step = [&]() __attribute__((always_inline)) {
        PROBE_MARK();
        wait_vmcnt<(AHEAD - 1) * PPW>();                    // two later steps (real or dummy) may still be in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        PROBE_ADD(0);
        set_dma_step(t_issue); ++t_issue;                   // its pieces go out between this step's MFMA groups
        const unsigned slot = lds0 + (t_cons % RING) * SLOT;
        ++t_cons;
        return slot;
    };

    f32x16 hacc[2], hprev[2], oacc[NT2];
    Frag hfh[2][2], hfl[2][2];
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[t][e] = 0.f;

    // lane part of the fragment addresses: row r of a 32-row tile, unit (2 sk + h) of the hi half (lo: ^ 64), swizzled by (r >> 1) & 7
    const int fsw = (r >> 1) & 7;
    const unsigned fa_hi0 = r * 128 + (((0 + h) ^ fsw) << 4), fa_hi1 = r * 128 + (((2 + h) ^ fsw) << 4);
#ifdef MLPX3_CLOCK_PROBE
    probe_sec[5] = __builtin_amdgcn_s_memtime() - probe_t0;
#endif

    // Fragment reads are software-pipelined by hand (as in mlp_bf16.hip): with one wave per SIMD nothing else hides the LDS latency, and
    // hipcc schedules read -> wait -> 3 MFMAs with only two fragment buffers (2 130 clocks per step for 1 536 of MFMA, probe).  The reads
    // are inline asm (the compiler neither waits for nor moves them): the 4 fragments of group g + 1 are requested, a counted lgkmcnt
    // retires exactly the 4 of group g (LDS returns in order), a sched_barrier keeps the 6 MFMAs of the group below it.
#define LDS_READ128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
// the wait names the four fragment registers it retires as in/out operands: nothing that reads them can be placed above it, and the
// operand list is printed behind the instruction so that the build-time ISA audit (profiles/tools/isa_audit.py) can check that no compiler
// copy / spill touches a register between the asm read that targets it and this wait
#define LGKM_LANDED(n, B) do { asm volatile("s_waitcnt lgkmcnt(%4) ; landed %0 %1 %2 %3" : "+v"((B)[0]), "+v"((B)[1]), "+v"((B)[2]), "+v"((B)[3]) : "n"(n) : "memory"); \
                               __builtin_amdgcn_sched_barrier(0); } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)

    // GELU (exact erf) + split of the previous block, one hidden value at a time between the MFMA groups of the next block's phase 1:
    // value v = 0..31 of the lane: tile jt = v >> 4, k-step sg = (v >> 3) & 1, pair i = (v >> 1) & 3, element z = v & 1.  A packed
    // accumulator tile is the B fragment of phase 2 (k order of split_weights layout 1).
    // r05 experiment, -DMLPX3_GELU_7128: the pair (z = 0, 1) through the packed form of common.h (A&S 7.1.28: one transcendental per value,
    // v_pk_fma / v_pk_mul for the rest; 12.5 issue slots per value instead of 21) in two halves - the 16th power behind group z = 0,
    // reciprocal / sign / product / split behind group z = 1.  Measured (profiles/r05_mlp_x3_packed_gelu_ab.txt): C = 128 form -5.5 %, C = 256
    // form unchanged (its GELU already runs under the MFMAs), the step -0.05 %; its 8.8e-7 GELU error (7.1.26: 4.6e-7) moved the smoke pair's
    // image error from 1.28e-4 to 1.33e-4 (bf16 halves 6.7e-4 -> 8.0e-4): not worth it, the default stays 7.1.26.
#ifndef MLPX3_GELU_7128
    float gfine[2];
    auto gelu_fine = [&](int v) __attribute__((always_inline)) {
        const int jt = v >> 4, sg = (v >> 3) & 1, i = (v >> 1) & 3, z = v & 1;
        gfine[z] = gelu_erf(PIPE ? hprev[jt][8 * sg + 2 * i + z] : hacc[jt][8 * sg + 2 * i + z]);
        if (z == 1) {
            unsigned a, l;
            split_x2<FMT>(gfine[0], gfine[1], a, l);
            hfh[jt][sg].u[i] = a;
            hfl[jt][sg].u[i] = l;
            asm volatile("" :: "v"(hfl[jt][sg].u[i]));          // pin: LLVM otherwise sinks the arithmetic below the MFMAs
        } else {
            asm volatile("" :: "v"(gfine[0]));
        }
    };
#else
    f32x2 gx, gp16;
    auto gelu_fine = [&](int v) __attribute__((always_inline)) {
        const int jt = v >> 4, sg = (v >> 3) & 1, i = (v >> 1) & 3, z = v & 1;
        if (z == 0) {
            gx.x = PIPE ? hprev[jt][8 * sg + 2 * i] : hacc[jt][8 * sg + 2 * i];
            gx.y = PIPE ? hprev[jt][8 * sg + 2 * i + 1] : hacc[jt][8 * sg + 2 * i + 1];
            gp16 = gelu_erf2_pow(gx);
            asm volatile("" :: "v"(gp16));                      // pin: LLVM otherwise sinks the arithmetic below the MFMAs
        } else {
            const f32x2 g = gelu_erf2_finish(gx, gp16);
            unsigned a, l;
            split_x2<FMT>(g.x, g.y, a, l);
            hfh[jt][sg].u[i] = a;
            hfl[jt][sg].u[i] = l;
            asm volatile("" :: "v"(hfl[jt][sg].u[i]));
        }
    };
#endif

    // !PIPE forms: the pieces of the look-ahead step behind MFMA group `it` of 8.  With a two-slot ring (one step of look-ahead) a piece
    // issued late in the step has not landed when the next step begins, so they go out two at a time in the first half of the step.
    auto early_pieces = [&](int it) __attribute__((always_inline)) {
        if (RING == 2) { if (2 * it < PPW) dma_piece(2 * it); if (2 * it + 1 < PPW) dma_piece(2 * it + 1); }
        else if (PPW == 8) dma_piece(it);
        else if ((it & 1) == 0) dma_piece(it >> 1);
    };
    auto lds_ptr = [&](unsigned off) { return (const unsigned char*)(__attribute__((address_space(3))) unsigned char*)(size_t)off; };
    auto phase1_step = [&](unsigned slot, int i1, bool pipe) __attribute__((always_inline)) {
        if constexpr (!PIPE) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int b = it >> 1, sk = it & 1, s = (i1 * 4 + b) * 2 + sk;
                const unsigned ad = slot + (sk ? fa_hi1 : fa_hi0) + b * (HB * 128);
                Frag wh[2], wl[2];
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
                    wh[jt].u = *(const u32x4*)lds_ptr(ad + jt * 4096);
                    wl[jt].u = *(const u32x4*)lds_ptr((ad ^ 64u) + jt * 4096);
                }
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) {
                    hacc[jt] = mfma_x3_32<FMT>(wh[jt].u, xl[s].u, hacc[jt]);
                    hacc[jt] = mfma_x3_32<FMT>(wl[jt].u, xh[s].u, hacc[jt]);
                    hacc[jt] = mfma_x3_32<FMT>(wh[jt].u, xh[s].u, hacc[jt]);
                    if (jt == 0) early_pieces(it);
                }
            }
            return;
        }
        const unsigned ahi[2] = {slot + fa_hi0, slot + fa_hi1};
        const unsigned alo[2] = {(slot + fa_hi0) ^ 64u, (slot + fa_hi1) ^ 64u};
        u32x4 a[2][4];                                       // [buffer][wh0, wl0, wh1, wl1]
        auto rd = [&](int buf, int it) __attribute__((always_inline)) {
            const int b = it >> 1, sk = it & 1;
            if (b == 0) { LDS_READ128(a[buf][0], ahi[sk], 0); LDS_READ128(a[buf][1], alo[sk], 0); LDS_READ128(a[buf][2], ahi[sk], 4096); LDS_READ128(a[buf][3], alo[sk], 4096); }
            else if (b == 1) { LDS_READ128(a[buf][0], ahi[sk], 8192); LDS_READ128(a[buf][1], alo[sk], 8192); LDS_READ128(a[buf][2], ahi[sk], 12288); LDS_READ128(a[buf][3], alo[sk], 12288); }
            else if (b == 2) { LDS_READ128(a[buf][0], ahi[sk], 16384); LDS_READ128(a[buf][1], alo[sk], 16384); LDS_READ128(a[buf][2], ahi[sk], 20480); LDS_READ128(a[buf][3], alo[sk], 20480); }
            else { LDS_READ128(a[buf][0], ahi[sk], 24576); LDS_READ128(a[buf][1], alo[sk], 24576); LDS_READ128(a[buf][2], ahi[sk], 28672); LDS_READ128(a[buf][3], alo[sk], 28672); }
        };
        rd(0, 0);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            if (it + 1 < 8) { rd((it + 1) & 1, it + 1); LGKM_LANDED(4, a[it & 1]); } else { LGKM_LANDED(0, a[it & 1]); }
            const int s = (i1 * 4 + (it >> 1)) * 2 + (it & 1);             // k-step of x
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                Frag wh, wl; wh.u = a[it & 1][2 * jt]; wl.u = a[it & 1][2 * jt + 1];
                hacc[jt] = mfma_x3_32<FMT>(wh.u, xl[s].u, hacc[jt]);
                hacc[jt] = mfma_x3_32<FMT>(wl.u, xh[s].u, hacc[jt]);
                hacc[jt] = mfma_x3_32<FMT>(wh.u, xh[s].u, hacc[jt]);
                if (jt == 0) { SB(); if (PPW == 8) dma_piece(it); else if ((it & 1) == 0) dma_piece(it >> 1); SB(); }
                if (pipe) {
                    constexpr int VPG = 32 / (P1 * 16);                    // hidden values per MFMA group: 1 (C = 256) or 2 (C = 128)
                    SB();
#pragma unroll
                    for (int u = 0; u < VPG; ++u) gelu_fine(((i1 * 8 + it) * 2 + jt) * VPG + u);
                    SB();
                }
            }
        }
    };
    auto phase2_step = [&](unsigned slot, int i2) __attribute__((always_inline)) {
        if constexpr (!PIPE) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int sg = C == 256 ? it >> 2 : (it >> 1) & 1;
                const int nt0 = C == 256 ? 2 * (it & 3) : 2 * (it & 1);
                const int jt = C == 256 ? i2 : (it >> 2);
                const unsigned ad = slot + (sg ? fa_hi1 : fa_hi0) + (C == 256 ? 0 : (it >> 2)) * (R2 * 128) + nt0 * 4096;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    Frag wh, wl;
                    wh.u = *(const u32x4*)lds_ptr(ad + q * 4096);
                    wl.u = *(const u32x4*)lds_ptr((ad ^ 64u) + q * 4096);
                    oacc[nt0 + q] = mfma_x3_32<FMT>(wh.u, hfl[jt][sg].u, oacc[nt0 + q]);
                    oacc[nt0 + q] = mfma_x3_32<FMT>(wl.u, hfh[jt][sg].u, oacc[nt0 + q]);
                    oacc[nt0 + q] = mfma_x3_32<FMT>(wh.u, hfh[jt][sg].u, oacc[nt0 + q]);
                    if (q == 0) early_pieces(it);
                }
            }
            return;
        }
        const unsigned ahi[2] = {slot + fa_hi0, slot + fa_hi1};
        const unsigned alo[2] = {(slot + fa_hi0) ^ 64u, (slot + fa_hi1) ^ 64u};
        u32x4 a[2][4];                                       // [buffer][wh(nt0), wl(nt0), wh(nt1), wl(nt1)]
        // group it = 0..7: C = 256: k-step sg = it >> 2, output tiles 2 (it & 3), +1 of the one image;  C = 128: image jb = it >> 2,
        // sg = (it >> 1) & 1, output tiles 2 (it & 1), +1
        auto rd = [&](int buf, int it) __attribute__((always_inline)) {
            const int sg = C == 256 ? it >> 2 : (it >> 1) & 1;
            const int nt0 = C == 256 ? 2 * (it & 3) : 2 * (it & 1);
            const int img = C == 256 ? 0 : (it >> 2);
            const int o0 = img * (R2 * 128) + nt0 * 4096;
            // immediates must be literal constants per call site: enumerate the (at most eight) offsets
#define RD4(OFF) do { LDS_READ128(a[buf][0], ahi[sg], OFF); LDS_READ128(a[buf][1], alo[sg], OFF); LDS_READ128(a[buf][2], ahi[sg], OFF + 4096); LDS_READ128(a[buf][3], alo[sg], OFF + 4096); } while (0)
            if (o0 == 0) RD4(0); else if (o0 == 8192) RD4(8192); else if (o0 == 16384) RD4(16384); else RD4(24576);
#undef RD4
        };
        rd(0, 0);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            if (it + 1 < 8) { rd((it + 1) & 1, it + 1); LGKM_LANDED(4, a[it & 1]); } else { LGKM_LANDED(0, a[it & 1]); }
            const int sg = C == 256 ? it >> 2 : (it >> 1) & 1;
            const int nt0 = C == 256 ? 2 * (it & 3) : 2 * (it & 1);
            const int jt = C == 256 ? i2 : (it >> 2);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                Frag wh, wl; wh.u = a[it & 1][2 * q]; wl.u = a[it & 1][2 * q + 1];
                oacc[nt0 + q] = mfma_x3_32<FMT>(wh.u, hfl[jt][sg].u, oacc[nt0 + q]);
                oacc[nt0 + q] = mfma_x3_32<FMT>(wl.u, hfh[jt][sg].u, oacc[nt0 + q]);
                oacc[nt0 + q] = mfma_x3_32<FMT>(wh.u, hfh[jt][sg].u, oacc[nt0 + q]);
                if (q == 0) { SB(); if (PPW == 8) dma_piece(it); else if ((it & 1) == 0) dma_piece(it >> 1); SB(); }
            }
        }
    };
    // accumulators of a hidden block start from its bias b1: rows = hidden units 64 hb + 32 jt + 8 g + 4 h + (0..3)
    auto init_hacc = [&](int hb) __attribute__((always_inline)) {
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *(const f32x4*)(b1s + hb * HB + jt * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) hacc[jt][4 * g + e] = bv[e];
            }
    };

    if constexpr (PIPE) {
        // prologue: phase 1 of block 0
        init_hacc(0);
#pragma unroll
        for (int i1 = 0; i1 < P1; ++i1) { const unsigned slot = begin_step(); phase1_step(slot, i1, false); PROBE_ADD(2); }
        for (int hb = 0; hb < nhb; ++hb) {
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) hprev[jt] = hacc[jt];
            if (hb + 1 < nhb) {
                init_hacc(hb + 1);
#pragma unroll
                for (int i1 = 0; i1 < P1; ++i1) { const unsigned slot = begin_step(); phase1_step(slot, i1, true); PROBE_ADD(2); }     // + GELU(hb) in the gaps
            } else {
                PROBE_MARK();
#pragma unroll
                for (int v = 0; v < 32; ++v) gelu_fine(v);
                PROBE_ADD(3);
            }
#pragma unroll
            for (int i2 = 0; i2 < P2; ++i2) { const unsigned slot = begin_step(); phase2_step(slot, i2); PROBE_ADD(4); }
        }
    } else {
        // same stream order (phase 1 of block hb + 1 before phase 2 of block hb), so the accumulators of block hb + 1 would overwrite those
        // of block hb: without a second set the GELU of block hb runs right after its phase 1, and phase 1 of block hb + 1 follows phase 2
        // of block hb - the stream is therefore consumed in the order W1(0) | W2(0)... NOT the pipelined one: see set_dma_step (SEQ order)
        for (int hb = 0; hb < nhb; ++hb) {
            init_hacc(hb);
#pragma unroll
            for (int i1 = 0; i1 < P1; ++i1) { const unsigned slot = begin_step(); phase1_step(slot, i1, false); PROBE_ADD(2); }
            PROBE_MARK();
#pragma unroll
            for (int v = 0; v < 32; ++v) gelu_fine(v);
            PROBE_ADD(3);
#pragma unroll
            for (int i2 = 0; i2 < P2; ++i2) { const unsigned slot = begin_step(); phase2_step(slot, i2); PROBE_ADD(4); }
        }
    }
    PROBE_MARK();

    // ---- epilogue: transpose the wave's [32 tokens][C] tile through the idle ring, then whole rows: + b2 + residual -> out -----------------
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // the dummy look-ahead pieces too
    __builtin_amdgcn_s_barrier();                                    // every wave is done reading the last weight slot
    {
        float* stage = (float*)(lds + wave * (32 * C * 4));          // [32][C] fp32, 16-byte chunks swizzled by token & 15
        constexpr int CHUNKS = C / 4;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = nt * 8 + 2 * g + h;                    // chunk of channels nt*32 + 8g + 4h .. +3
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = oacc[nt][4 * g + e];
                *(f32x4*)(stage + r * C + ((c ^ (r & 15)) << 2)) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // wave-private tile: LDS operations of a wave complete in order
        constexpr int RPI = 64 / CHUNKS;                             // rows per wave instruction (1 for C = 256, 2 for C = 128)
        const int q = lane % CHUNKS, rsub = lane / CHUNKS;
        const int m0w = blockIdx.x * (NW * 32) + wave * 32;
        const f32x4 bv = *(const f32x4*)(b2s + 4 * q);
        constexpr int BATCH = 8;
#pragma unroll
        for (int it0 = 0; it0 < 32 / RPI; it0 += BATCH) {
            f32x4 rr[BATCH];
            if (p.res) {
#pragma unroll
                for (int it = 0; it < BATCH; ++it) {
                    int mm = m0w + (it0 + it) * RPI + rsub; mm = mm < p.M ? mm : p.M - 1;
                    rr[it] = *(const f32x4*)(p.res + (size_t)mm * p.ldr + 4 * q);
                }
            }
#pragma unroll
            for (int it = 0; it < BATCH; ++it) {
                const int row = (it0 + it) * RPI + rsub, mm = m0w + row;
                f32x4 v = *(const f32x4*)(stage + row * C + ((q ^ (row & 15)) << 2));
                v += bv;
                if (p.res) v += rr[it];
                if (mm < p.M) *(f32x4*)(p.out + (size_t)mm * p.ldo + 4 * q) = v;
            }
        }
    }
#ifdef MLPX3_CLOCK_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PROBE_ADD(6);
    if (tid == 0 && g_probe) {
        g_probe[blockIdx.x * 4] = probe_t0; g_probe[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
        g_probe[blockIdx.x * 4 + 2] = probe_r0; g_probe[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 7; ++i) g_probe[gridDim.x * 4 + blockIdx.x * 7 + i] = probe_sec[i];
    }
#endif
}

template <int C, int NW, bool PIPE, int RING, int WGPC, int FMT>
int launch(const MlpX3Args& p, hipStream_t s) {
    static_assert(NW * 32 * C * 4 <= RING * SLOT, "the epilogue's per-wave [32][C] fp32 tiles must fit in the ring");
    const size_t lds = (size_t)RING * SLOT + (size_t)(p.HID + 3 * C) * sizeof(float) + NW * 1024;
    if (lds > 160 * 1024) { master_set_error("mlp(x3): hidden width %d needs %zu B of LDS", p.HID, lds); return MASTER_ERR_UNSUPPORTED; }
    static LdsLimit lds_limit;
    lds_limit.ensure((const void*)mlp_x3_kernel<C, NW, PIPE, RING, WGPC, FMT>, (int)lds);
    hipLaunchKernelGGL((mlp_x3_kernel<C, NW, PIPE, RING, WGPC, FMT>), dim3((p.M + NW * 32 - 1) / (NW * 32)), dim3(NW * 64), lds, s, p);
    return master_check_launch("mlp_x3");
}

}  // namespace

int mlp_x3(const float* x, const float* ln_g, const float* ln_b, const void* w1, const float* b1, const void* w2, const float* b2,
           const float* res, float* out, int M, int C, int HID, int ldx, int ldr, int ldo, float eps, hipStream_t s, int fmt) {
    MlpX3Args p{x, ln_g, ln_b, (const unsigned char*)w1, b1, (const unsigned char*)w2, b2, res, out, M, HID, ldx, ldr, ldo, eps};
    if (HID % HB != 0) { master_set_error("mlp(x3): hidden width must be a multiple of 64 (got %d)", HID); return MASTER_ERR_UNSUPPORTED; }
    if (C == 256) return fmt == X3_F16 ? launch<256, 4, true, 4, 1, X3_F16>(p, s) : launch<256, 4, true, 4, 1, X3_BF16>(p, s);
    if (C == 128) {
        static int form = -1;                     // MASTER_MLP_X3_C128 = 8w: the first form (eight waves, 256 tokens, one workgroup per CU) for A/B runs
        if (form < 0) { const char* e = getenv("MASTER_MLP_X3_C128"); form = (e && e[0] == '8') ? 1 : 0; }
        if (fmt == X3_F16) return form ? launch<128, 8, false, 4, 1, X3_F16>(p, s) : launch<128, 4, false, 2, 2, X3_F16>(p, s);
        return form ? launch<128, 8, false, 4, 1, X3_BF16>(p, s) : launch<128, 4, false, 2, 2, X3_BF16>(p, s);
    }
    master_set_error("mlp(x3): channel width %d not built (128, 256)", C);
    return MASTER_ERR_UNSUPPORTED;
}
