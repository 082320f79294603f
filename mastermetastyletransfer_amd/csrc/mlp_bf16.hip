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
//                                                     (bias + GELU, packed to bf16 in registers)
// so the hidden activations never touch LDS or HBM.  A 32x32 accumulator tile used as B operand has the k
// order j = 16s + 8(e>>2) + 4*half + (e&3); two v_permlane32_swap per k-step exchange the middle quarters
// between the half-waves and restore the natural order j = 16s + 8*half + e, so W2 is read exactly like W1.
//
// Weights stream through a 4-slot LDS ring in 32-KB steps by LDS-DMA (global_load_lds, 16 B per lane: no
// VGPR destination, so the compiler cannot sink the prefetch below the MFMAs), issued three steps ahead and
// retired with counted s_waitcnt vmcnt + a raw s_barrier (one barrier per step, 32 MFMAs per wave per step).
// The DMA writes LDS linearly (wave-uniform base + lane*16), so bank conflicts are avoided by swizzling the
// SOURCE: position p of a row holds source chunk p ^ f(row), and fragment reads apply the same XOR
// (f = row&15 for 256-B rows, (row>>1)&7 for 128-B rows: every 16-lane ds_read_b128 group hits 16 distinct
// 16-B slots).  The GELU of hidden chunk hc is sliced between the phase-1 MFMAs of chunk hc+1.
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

union Frag { u32x4 u; bf16x8_t b; };

constexpr int SLOT = 32768;      // bytes per ring slot = one step of weights
constexpr int RING = 4;          // slots
constexpr int AHEAD = 3;         // steps of DMA in flight ahead of the one being consumed
constexpr int DMA_PER_STEP = 8;  // global_load_lds instructions per wave per step (4 waves x 8 x 1 KB)

#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

template <int C>
__global__ __launch_bounds__(256, 1) void mlp_bf16_kernel(MlpArgs p) {
    constexpr int KS1 = C / 16;                 // k-steps of phase 1 (K1 = C)
    constexpr int KB = C / 128;                 // W1 steps per hidden chunk ([128 j][128 k] = 32 KB each)
    constexpr int NT2 = C / 32;                 // output-channel tiles of phase 2
    constexpr int S2 = C / 128;                 // W2 steps per hidden chunk: C=256 -> 2 x [256 n][64 j]; C=128 -> 1 x [128 n][128 j]
    constexpr int JB2 = 128 / S2;               // hidden units per W2 step (64 or 128)
    constexpr int ROW2 = JB2 * 2;               // bytes per W2 row in a slot (128 or 256)
    constexpr int SPH = KB + S2;                // steps per hidden chunk

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [RING][SLOT] weights, then b1 as fp32[HID]
    float* b1s = (float*)(lds + RING * SLOT);                              // [HID]
    float* b2s = b1s + p.HID;                                              // [C]
    float* lng = b2s + C;                                                  // [C] LayerNorm gamma
    float* lnb = lng + C;                                                  // [C] LayerNorm beta

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MLP_CLOCK_PROBE   // diagnostic build only (profiles/probes/clock_probe.hip)
    const unsigned long long probe_t0 = __builtin_amdgcn_s_memtime(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int r = lane & 31, h = lane >> 5;
    const int m = blockIdx.x * 128 + wave * 32 + r;
    const int mc = m < p.M ? m : p.M - 1;
    const int nhc = p.HID / 128;
    const int T = KB + (nhc - 1) * SPH + S2;    // stream: W1(0) | W1(1) W2(0) | ... | W1(nhc-1) W2(nhc-2) | W2(nhc-1)

    // ---- LDS-DMA of stream step t into ring slot t % RING ----------------------------------------------
    // Per-lane source offsets (elements) of the 8 one-KB pieces this wave moves per step are computed once; a step
    // then only adds a wave-uniform base.  256-B-row layout (W1, and W2 when C = 128): piece I = rows 4I..4I+3,
    // source chunk q = (lane&15) ^ (row&15).  128-B-row layout (W2 when C = 256): piece I = rows 8I..8I+7,
    // q = (lane&7) ^ ((row>>1)&7).
    const int row0 = wave * (DMA_PER_STEP * 4) + (lane >> 4), l15 = lane & 15;        // 256-B-row layout
    const int row0b = wave * (DMA_PER_STEP * 8) + (lane >> 3), l7 = lane & 7;         // 128-B-row layout
    // One stream step = DMA_PER_STEP one-KB pieces per wave, issued one at a time between the MFMAs of the step being computed (the
    // L2 -> LDS path moves 64 B/clk per CU: a burst of 4 x 8 KB at the top of a step kept all four waves, and with them the matrix
    // cores, waiting in the issue queue for ~580 cycles per step).  A wave issues in order and only ~2 instructions hide under an
    // MFMA, so a piece is kept to FIVE instructions (r02; it was ten with a branch: ~19 k of 108 k clocks per workgroup went to the
    // weight stream, profiles/r02_mlp_ablate.txt): set_dma_step() works out a wave-uniform running source pointer and LDS address
    // once per step; dma_piece() selects the lane's precomputed byte offset (one v_cndmask), issues  global_load_lds  with the
    // scalar base, and advances both by scalar adds.  Steps past the end of the stream are not skipped (no branch, the vmcnt
    // bookkeeping stays uniform): they re-read the first KB of W1 into a 1-KB dummy region per wave.
    // Per-lane offsets: row * row pitch + swizzled 16-B chunk; (row & 15) of piece i depends on i & 3 only, ((row >> 1) & 7) on
    // i & 1 only (waves own 32-/64-row bands).  Named scalars, not arrays: a select between array elements becomes scratch.
#define SW256(i) ((l15 ^ ((row0 + 4 * (i)) & 15)) << 4)      /* bytes */
#define SW128(i) ((l7 ^ (((row0b + 8 * (i)) >> 1) & 7)) << 4)
    const unsigned vo1_0 = row0 * (C * 2) + SW256(0), vo1_1 = row0 * (C * 2) + SW256(1), vo1_2 = row0 * (C * 2) + SW256(2),
                   vo1_3 = row0 * (C * 2) + SW256(3);                                              // W1 steps
    const unsigned vo2_0 = ROW2 == 128 ? row0b * (p.HID * 2) + SW128(0) : row0 * (p.HID * 2) + SW256(0),
                   vo2_1 = ROW2 == 128 ? row0b * (p.HID * 2) + SW128(1) : row0 * (p.HID * 2) + SW256(1),
                   vo2_2 = ROW2 == 128 ? vo2_0 : row0 * (p.HID * 2) + SW256(2), vo2_3 = ROW2 == 128 ? vo2_1 : row0 * (p.HID * 2) + SW256(3);   // W2 steps
    const int pstride_w1 = 4 * C * 2, pstride_w2 = (ROW2 == 128 ? 8 : 4) * p.HID * 2;
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds;
    const unsigned dummy_lds = lds0 + RING * SLOT + (p.HID + 3 * C) * 4 + wave * 1024;
    const char* d_ptr = nullptr; unsigned d_pstride = 0, d_lds = 0, d_lstride = 0, d_w1m = 0;
    auto set_dma_step = [&](int t) __attribute__((always_inline)) {
        const bool valid = t < T;
        bool is_w1; int hc, idx;
        if (t < KB) { is_w1 = true; hc = 0; idx = t; }
        else {
            const int u = t - KB, b = u / SPH, o = u - b * SPH;
            if (b < nhc - 1 && o < KB) { is_w1 = true; hc = b + 1; idx = o; }
            else { is_w1 = false; hc = b; idx = b < nhc - 1 ? o - KB : o; }
        }
        if (!valid) { is_w1 = true; hc = 0; idx = 0; }
        // mask arithmetic, not a ternary on two variables: hipcc turns such a select into a private table in scratch memory
        const int mk = -(int)is_w1, vm = -(int)valid;
        d_w1m = (unsigned)mk;
        d_pstride = (unsigned)(((pstride_w1 & mk) | (pstride_w2 & ~mk)) & vm);
        d_lstride = 1024u & (unsigned)vm;
        d_lds = ((lds0 + (t % RING) * SLOT + wave * DMA_PER_STEP * 1024) & (unsigned)vm) | (dummy_lds & ~(unsigned)vm);
        const char* b1p = (const char*)(p.w1 + (size_t)hc * 128 * C + idx * 128);
        const char* b2p = (const char*)(p.w2 + hc * 128 + (ROW2 == 128 ? idx * JB2 : 0));
        d_ptr = (const char*)(((size_t)b1p & (size_t)(long long)mk) | ((size_t)b2p & ~(size_t)(long long)mk));
    };
    auto dma_piece = [&](int i) __attribute__((always_inline)) {                                      // pieces of a step go out in order i = 0..7
        const unsigned a1 = (i & 3) == 0 ? vo1_0 : (i & 3) == 1 ? vo1_1 : (i & 3) == 2 ? vo1_2 : vo1_3;     // i is a constant here
        const unsigned a2 = (i & 3) == 0 ? vo2_0 : (i & 3) == 1 ? vo2_1 : (i & 3) == 2 ? vo2_2 : vo2_3;
        const unsigned voff = (a1 & d_w1m) | (a2 & ~d_w1m);                                           // one v_bfi_b32 with a scalar mask
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(d_ptr + voff),
                                         (__attribute__((address_space(3))) void*)(size_t)d_lds, 16, 0, 0);
        d_ptr += (size_t)d_pstride; d_lds += d_lstride;
    };
    auto issue = [&](int t) __attribute__((always_inline)) {
        set_dma_step(t);
#pragma unroll
        for (int i = 0; i < DMA_PER_STEP; ++i) dma_piece(i);
    };

    // ---- X fragments (B operand of phase 1): lane (token m, half h) holds x[m][16s + 8h .. +8].  Requested BEFORE the first DMA burst:
    //      loads return in order, so behind 96 KB of weights they would arrive ~1.5 k clocks later and the LayerNorm with them --------
    Frag xf[KS1];
    {
        const bf16_t* xp = p.x + (size_t)mc * p.ldx + 8 * h;
#pragma unroll
        for (int s = 0; s < KS1; ++s) xf[s].u = *(const u32x4*)(xp + 16 * s);
    }

    int t_issue = 0;
    for (; t_issue < AHEAD; ++t_issue) issue(t_issue);

    // small fp32 parameter vectors live in LDS: ordinary global loads inside the DMA pipeline would each be followed
    // by a full vmcnt(0) drain
    for (int i = tid; i < p.HID; i += 256) b1s[i] = p.b1[i];
    if (tid < C) {
        b2s[tid] = p.b2[tid];
        if (p.ln_g) { lng[tid] = p.ln_g[tid]; lnb[tid] = p.ln_b[tid]; }
    }

    {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // parameter vectors visible to every wave
#ifndef MLP_LN_SCALAR        // r05: packed math (common.h); -DMLP_LN_SCALAR keeps the r01-r04 form below for A/B runs
        if (p.ln_g) layernorm_frags_pk<KS1>(xf, lng, lnb, h, p.eps);
#else
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
                const float* gp = lng + 16 * s + 8 * h; const float* bp = lnb + 16 * s + 8 * h;
                const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 4);
                const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
                float o[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (v[s][e] - mean) * rstd * g0[e] + b0[e];
                    o[4 + e] = (v[s][4 + e] - mean) * rstd * g1[e] + b1[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) xf[s].u[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]);
            }
        }
#endif
    }

    // ---- per-step synchronisation: my DMAs of step t have landed -> barrier -> everyone's have, and everyone
    //      is done reading step t-1, whose slot is the one step t+AHEAD goes into -----------------------------
#ifdef MLP_CLOCK_PROBE
    unsigned long long probe_sec[6] = {0, 0, 0, 0, 0, 0}, probe_last = 0;     // wait+barrier, DMA issue, phase 1, phase 2, prologue, epilogue
#define PROBE_MARK() do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); probe_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define PROBE_ADD(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); probe_sec[i] += t_ - probe_last; probe_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PROBE_MARK() do {} while (0)
#define PROBE_ADD(i) do {} while (0)
#endif
    int t_cons = 0;
    auto begin_step = [&]() __attribute__((always_inline)) {
        PROBE_MARK();
        WAIT_VMCNT(16);                                     // two later steps (real or dummy: 8 pieces each) may still be in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        PROBE_ADD(0);
        set_dma_step(t_issue);                              // its pieces go out between this step's MFMA groups
        ++t_issue;
        PROBE_ADD(1);
        const unsigned char* slot = lds + (t_cons % RING) * SLOT;
        ++t_cons;
        return slot;
    };

    f32x16 hacc[4], hprev[4], oacc[NT2];
    Frag hf[4][2];
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[t][e] = 0.f;

    // GELU + pack of hidden chunk hc, sliced: a k-step of phase 2 (tile jt, half s) is registers 8s..8s+7 of hprev[jt] = four
    // pairs (gelu_pair, 10 VALU instructions each), then the half-wave exchange that turns the accumulator k order into the
    // natural one (gelu_swap).
    unsigned gu[4];
    auto gelu_pair = [&](int jt, int s, int q) __attribute__((always_inline)) {
        const int e0 = 4 * (2 * s + (q >> 1)) + 2 * (q & 1);
#if defined(MLP_X_CHEAPGELU)          // probe builds: identity instead of the polynomial (what does the GELU arithmetic cost?)
        const f32x2 v = f32x2{hprev[jt][e0], hprev[jt][e0 + 1]};
#elif defined(MLP_X_PLAINGELU)        // probe builds: the same polynomial in un-packed fp32 (18 instructions per pair instead of 9 packed)
        f32x2 v;
#pragma unroll
        for (int z = 0; z < 2; ++z) v[z] = gelu_poly1(hprev[jt][e0 + z]);
#else
        const f32x2 v = gelu_poly2(f32x2{hprev[jt][e0], hprev[jt][e0 + 1]});          // common.h; b1 is already in the accumulator
#endif
        gu[q] = pack_bf16x2(v[0], v[1]);
    };
    auto gelu_swap = [&](int jt, int s) __attribute__((always_inline)) {
        // half 0 keeps u0,u1 and takes the partner's u0,u1 as words 2,3; half 1 takes the partner's u2,u3 as words 0,1
        auto sw0 = __builtin_amdgcn_permlane32_swap(gu[0], gu[2], false, false);
        auto sw1 = __builtin_amdgcn_permlane32_swap(gu[1], gu[3], false, false);
        hf[jt][s].u[0] = sw0[0]; hf[jt][s].u[2] = sw0[1];
        hf[jt][s].u[1] = sw1[0]; hf[jt][s].u[3] = sw1[1];
    };
    // Fine-grained form (default, r02): the polynomial in un-packed fp32, ONE hidden value after every phase-1 MFMA of the next chunk.
    // Measured (profiles/r02_mlp_probe_*.txt, clocks per 128-token workgroup, C = 256): GELU arithmetic = 18.7 k of 111 k (identity
    // instead of the polynomial: 92.3 k); packed bursts 111.3 k, un-packed bursts 110.7 k, un-packed one value per MFMA 107.6 k, the
    // same dealt out four values at a time for instruction-level parallelism 107.3 k.  11 VALU instructions per 32-clock MFMA are
    // issue-bound (4 clocks each): only ~1/3 hides.  Hiding the rest means spreading chunk hc's GELU over phase 2 of chunk hc-1 as
    // well (5 instructions per MFMA), i.e. one more pipeline stage and a second hidden-fragment buffer - not done.
    float gfine[2];
    auto gelu_fine = [&](int v) __attribute__((always_inline)) {          // v = 0..63: value index inside the previous hidden chunk
        const int ks = v >> 3, jtg = ks >> 1, sg = ks & 1, q = (v >> 1) & 3, z = v & 1;
        const int e0 = 4 * (2 * sg + (q >> 1)) + 2 * (q & 1);
        gfine[z] = gelu_poly1(hprev[jtg][e0 + z]);                        // common.h
        if (z == 1) gu[q] = pack_bf16x2(gfine[0], gfine[1]);
        if ((v & 7) == 7) gelu_swap(jtg, sg);
        asm volatile("" :: "v"(gfine[z]));                                // pin: LLVM otherwise sinks the arithmetic below the MFMAs
    };
    auto gelu_kstep = [&](int jt, int s) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) gelu_pair(jt, s, q);
        gelu_swap(jt, s);
    };

    const int x256 = (h ^ (r & 15)) << 4;            // swizzle term for 256-B rows: chunk (2s+h) ^ (row&15)
    const int x128 = (h ^ ((r >> 1) & 7)) << 4;      //                  128-B rows: chunk (2g+h) ^ ((row>>1)&7)

    // Fragment reads are software-pipelined by hand.  With one wave per SIMD nothing else hides the LDS latency and
    // hipcc collapses a source-level double buffer back into read -> wait -> MFMA, so the reads are inline asm (the
    // compiler neither waits for nor moves them): the 4 (8) fragments of k-step s+1 are requested, a counted lgkmcnt
    // retires exactly the fragments of k-step s (LDS returns in order), and a sched_barrier keeps the MFMAs below it.
#define LDS_READ128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define LGKM_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    auto lds_off = [&](const unsigned char* ptr) { return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)ptr; };

    auto phase1_step = [&](const unsigned char* slot, int kb, bool pipe) __attribute__((always_inline)) {
        const unsigned base = lds_off(slot) + r * 256;
#ifdef MLP_PREFETCH2            // fragments requested TWO k-steps ahead (three register buffers): LDS latency under load > one k-step
        constexpr int NBUF = 3;
#else
        constexpr int NBUF = 2;
#endif
        u32x4 a[NBUF][4];
        auto rd = [&](int buf, int s) {
            const unsigned ad = base + ((32 * s) ^ x256);
            LDS_READ128(a[buf][0], ad, 0); LDS_READ128(a[buf][1], ad, 8192);
            LDS_READ128(a[buf][2], ad, 16384); LDS_READ128(a[buf][3], ad, 24576);
        };
#ifdef MLP_X_NORD1          // probe builds: no fragment reads in phase 1 (timing only, garbage operands)
#pragma unroll
        for (int i = 0; i < 4; ++i) { asm volatile("" : "=v"(a[0][i])); asm volatile("" : "=v"(a[1][i])); }
#else
        rd(0, 0);
        if (NBUF == 3) rd(1, 1);
#endif
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#ifndef MLP_X_NORD1
            if (NBUF == 3) {
                if (s + 2 < 8) { rd((s + 2) % 3, s + 2); LGKM_WAIT(8); } else if (s + 1 < 8) { LGKM_WAIT(4); } else { LGKM_WAIT(0); }
            } else {
                if (s + 1 < 8) { rd((s + 1) & 1, s + 1); LGKM_WAIT(4); } else { LGKM_WAIT(0); }
            }
#endif
#define SB() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                Frag f; f.u = a[s % NBUF][jt];
                hacc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b, xf[kb * 8 + s].b, hacc[jt], 0, 0, 0);
#ifndef MLP_X_NODMA1
                if (jt == 1) { SB(); dma_piece(s); SB(); }      // two VALU instructions: hides behind the MFMA
#endif
#ifndef MLP_X_BURSTGELU
                if (pipe) {
                    constexpr int VPM = 64 / (KB * 32);          // hidden values per MFMA: 1 (C = 256) or 2 (C = 128)
                    SB();
#pragma unroll
                    for (int u = 0; u < VPM; ++u) gelu_fine(((kb * 8 + s) * 4 + jt) * VPM + u);
                    SB();
                }
#endif
            }
#ifdef MLP_X_BURSTGELU
            if (pipe) {
                // 8 k-step slices of GELU(previous chunk) between the KB*8 MFMA groups of this chunk.  Each slice is one burst of
                // four independent packed-fp32 chains: v_pk_* does not overlap this wave's MFMAs (measured, issue_probe.hip:
                // 4 x (MFMA ; 4 pk_fma) = 260 clocks against 128 + 75), and a dependent chain issues only every 8 clocks, so
                // slicing finer than this costs more than it hides.
                const int i = kb * 8 + s;
                if (KB == 2) { if ((i & 1) == 0) gelu_kstep(i >> 2, (i >> 1) & 1); }
                else gelu_kstep(i >> 1, i & 1);
            }
#endif
        }
    };
    auto phase2_step = [&](const unsigned char* slot, int sb) __attribute__((always_inline)) {
        constexpr int NG = ROW2 == 128 ? 4 : 8;                 // k-steps of 16 hidden units in this slot
        const unsigned base = lds_off(slot) + r * ROW2;
        const int xs = ROW2 == 128 ? x128 : x256;
        u32x4 a[2][NT2];
        auto rd = [&](int buf, int g) {
            const unsigned ad = base + ((32 * g) ^ xs);
            if (ROW2 == 128) {          // C = 256: 8 tiles of 32 rows x 128 B
                LDS_READ128(a[buf][0], ad, 0); LDS_READ128(a[buf][1], ad, 4096); LDS_READ128(a[buf][2], ad, 8192);
                LDS_READ128(a[buf][3], ad, 12288);
                if (NT2 > 4) {
                    LDS_READ128(a[buf][NT2 > 4 ? 4 : 0], ad, 16384); LDS_READ128(a[buf][NT2 > 4 ? 5 : 0], ad, 20480);
                    LDS_READ128(a[buf][NT2 > 4 ? 6 : 0], ad, 24576); LDS_READ128(a[buf][NT2 > 4 ? 7 : 0], ad, 28672);
                }
            } else {                    // C = 128: 4 tiles of 32 rows x 256 B
                LDS_READ128(a[buf][0], ad, 0); LDS_READ128(a[buf][1], ad, 8192);
                LDS_READ128(a[buf][2], ad, 16384); LDS_READ128(a[buf][3], ad, 24576);
            }
        };
#ifdef MLP_X_NORD2          // probe builds: no fragment reads in phase 2
#pragma unroll
        for (int i = 0; i < NT2; ++i) { asm volatile("" : "=v"(a[0][i])); asm volatile("" : "=v"(a[1][i])); }
#else
        rd(0, 0);
#endif
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#ifndef MLP_X_NORD2
            if (g + 1 < NG) {
                rd((g + 1) & 1, g + 1);
                if (NT2 == 8) LGKM_WAIT(8); else LGKM_WAIT(4);
            } else { LGKM_WAIT(0); }
#endif
            const int jt = ROW2 == 128 ? sb * 2 + (g >> 1) : (g >> 1);
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) {
                Frag f; f.u = a[g & 1][nt];
                oacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b, hf[jt][g & 1].b, oacc[nt], 0, 0, 0);
#ifndef MLP_X_NODMA2
                if (NG == 4) { if (nt == 1 || nt == 5) { SB(); dma_piece(2 * g + (nt > 1)); SB(); } }
                else if (nt == 1) { SB(); dma_piece(g); SB(); }
#endif
            }
        }
    };
    // accumulators of hidden chunk hc start from its bias b1 (rows = hidden units: 16 per tile per lane, one batch of LDS reads)
    auto init_hacc = [&](int hc) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *(const f32x4*)(b1s + hc * 128 + t * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) hacc[t][4 * g + e] = bv[e];
            }
    };

#ifdef MLP_CLOCK_PROBE
    probe_sec[4] = __builtin_amdgcn_s_memtime() - probe_t0;
#endif
    // prologue: phase 1 of chunk 0
    init_hacc(0);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) { phase1_step(begin_step(), kb, false); PROBE_ADD(2); }

    for (int hc = 0; hc < nhc; ++hc) {
#pragma unroll
        for (int t = 0; t < 4; ++t) hprev[t] = hacc[t];
        if (hc + 1 < nhc) {
            init_hacc(hc + 1);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) { phase1_step(begin_step(), kb, true); PROBE_ADD(2); }    // + GELU(hc) in the gaps
        } else {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int s = 0; s < 2; ++s) gelu_kstep(jt, s);
        }
#pragma unroll
        for (int sb = 0; sb < S2; ++sb) { phase2_step(begin_step(), sb); PROBE_ADD(3); }
    }

    // ---- epilogue.  O^T tile: lane = token, registers = channels, which would mean 8-byte stores strided by the row
    //      pitch (64 per lane, each touching 32 rows).  Instead each wave transposes its [32 tokens][C] fp32 tile through
    //      the now idle weight ring (own 32-KB region, 16-B chunks swizzled by token&15) and walks it row by row:
    //      residual loads and bf16 stores are whole contiguous rows.
    PROBE_MARK();
    // residual = x without a LayerNorm (every MLP of the StyleEncoder and the last MLP: style_transformer.py:865-882,:1125): the raw X
    // fragments are still in registers.  Fragment s holds channels 16s + 8h .. +7 of the lane's token, the accumulators hold
    // 8g + 4h .. +3: the same half-wave exchange as the GELU path (two v_permlane32_swap per 16 channels) lines them up - no second
    // read of x from HBM and no load latency in the epilogue.
    const bool res_in_regs = p.res == p.x && p.ldr == p.ldx && p.ln_g == nullptr;
    if (res_in_regs) {
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            auto swa = __builtin_amdgcn_permlane32_swap(xf[s].u[0], xf[s].u[2], false, false);
            auto swb = __builtin_amdgcn_permlane32_swap(xf[s].u[1], xf[s].u[3], false, false);
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int e0 = 4 * (2 * (s & 1) + gg);
                oacc[s >> 1][e0 + 0] += __uint_as_float(swa[gg] << 16); oacc[s >> 1][e0 + 1] += __uint_as_float(swa[gg] & 0xffff0000u);
                oacc[s >> 1][e0 + 2] += __uint_as_float(swb[gg] << 16); oacc[s >> 1][e0 + 3] += __uint_as_float(swb[gg] & 0xffff0000u);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                   // every wave is done reading the last weight slot
    {
        float* stage = (float*)(lds + wave * 32768);               // [32][C] fp32, row pitch C*4 bytes (<= 1 KB)
        constexpr int CHUNKS = C / 4;                               // 16-B chunks per row
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = nt * 8 + 2 * g + h;                   // chunk of channels nt*32 + 8g + 4h .. +3
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = oacc[nt][4 * g + e];
                *(f32x4*)(stage + r * C + ((c ^ (r & 15)) << 2)) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // wave-private tile: LDS ops of a wave complete in order
        constexpr int RPI = 64 / CHUNKS;                            // rows per wave instruction (1 for C=256, 2 for C=128)
        const int q = lane % CHUNKS, rsub = lane / CHUNKS;
        const int m0w = blockIdx.x * 128 + wave * 32;
        const f32x4 bv = *(const f32x4*)(b2s + 4 * q);
        u32x2 rr[32 / RPI];
        const bool res_mem = p.res != nullptr && !res_in_regs;
        if (res_mem) {
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int mm = m0w + it * RPI + rsub;
                rr[it] = mm < p.M ? *(const u32x2*)(p.res + (size_t)mm * p.ldr + 4 * q) : u32x2{0u, 0u};
            }
        }
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int row = it * RPI + rsub, mm = m0w + row;
            f32x4 v = *(const f32x4*)(stage + row * C + ((q ^ (row & 15)) << 2));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bv[e];
            if (res_mem) {
                v[0] += __uint_as_float(rr[it][0] << 16); v[1] += __uint_as_float(rr[it][0] & 0xffff0000u);
                v[2] += __uint_as_float(rr[it][1] << 16); v[3] += __uint_as_float(rr[it][1] & 0xffff0000u);
            }
            u32x2 o; o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]);
            if (mm < p.M) *(u32x2*)(p.out + (size_t)mm * p.ldo + 4 * q) = o;
        }
    }
#ifdef MLP_CLOCK_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PROBE_ADD(5);
    if (tid == 0 && g_probe) {
        g_probe[blockIdx.x * 4] = probe_t0; g_probe[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
        g_probe[blockIdx.x * 4 + 2] = probe_r0; g_probe[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 6; ++i) g_probe[gridDim.x * 4 + blockIdx.x * 6 + i] = probe_sec[i];
    }
#endif
}

template <int C>
int launch(const MlpArgs& p, hipStream_t s) {
    const size_t lds = (size_t)RING * SLOT + (size_t)(p.HID + 3 * C) * sizeof(float) + 4 * 1024;      // ring | b1, b2, ln | 1-KB DMA dummy per wave
    if (lds > 160 * 1024) { master_set_error("mlp: hidden width %d needs %zu B of LDS", p.HID, lds); return MASTER_ERR_UNSUPPORTED; }
    static LdsLimit lds_limit;
    lds_limit.ensure((const void*)mlp_bf16_kernel<C>, (int)((int)lds));
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
