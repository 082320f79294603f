// EXPERIMENT (not in libmaster_hip.so): measured equal-or-slower than mlp_bf16.hip in round 1; kept with its clock probe for the record.
// Fused MLP block, bf16 throughput mode, producer/consumer form:   out = res + W2 . gelu(W1 . LN?(x) + b1) + b2
// (same contract as mlp_bf16.hip; reference style_transformer.py:366,:392,:839-841,:865-882,:991,:1125 and the Swin blocks).
//
// Why a second form.  In mlp_bf16.hip one wave per SIMD owns the whole register file and does everything; a wave issues
// in order, so its LDS-DMA issue, barrier waits, fragment-read latency and the GELU all run while that SIMD's matrix core
// idles (112 k clocks per workgroup for 32.8 k clocks of MFMA, profiles/probes).  Here a 128-token workgroup has EIGHT
// waves, two per SIMD, with different jobs and <= 256 registers each:
//   waves 0-3  "fc1"  token band w:   H^T tile = W1[32 hidden rows] . X^T + b1  (X fragments in registers, LayerNorm applied there)
//   waves 4-7  "fc2"  token band w-4: O^T += W2[:, 32 hidden cols] . gelu(H^T)  (B fragments through an LDS mailbox)
// so one wave's MFMAs run under the other's stalls.
//
// A tick = one 32-wide hidden tile.  The GELU of a tile (16 values per lane) is split between the roles and dealt out FOUR
// plain VALU instructions after every MFMA (which is what a wave can issue in the shadow of its own MFMA almost for free,
// profiles/probes/issue_probe.hip: 4 x (MFMA ; 4 v_fma) = 156 clocks against 128):
//   fc1, tick t:  MFMAs of tile t            || GELU of k-step 0 of tile t-1 (its own accumulators of the previous tick)
//                 end of tick: B fragment 0 of tile t-1 -> mailbox F0;  raw fp32 k-step-1 values of tile t -> mailbox RAW
//   fc2, tick t:  MFMAs of tile t-2 (fragment 0 from F0, fragment 1 from its own registers)
//                                            || GELU of k-step 1 of tile t-1 (RAW) -> fragment 1 for the next tick
// Ring slot of tick t = W1 rows [32t, 32t+32) (fc1 at tick t) followed by W2 columns of tile t-2 (fc2 at tick t); four slots,
// LDS-DMA three ticks ahead (fc1 waves move the W1 part, fc2 waves the W2 part, one piece between MFMAs), counted vmcnt +
// ONE workgroup barrier per tick, which also publishes the mailboxes (double-buffered by tick parity).  Both LDS images are
// source-swizzled so that every 16-lane group of a fragment ds_read_b128 touches 16 distinct 16-B bank slots:
//   W1 tile: rows of C*2 bytes, position p of row r holds source chunk p ^ (r & 15)          (low four bits)
//   W2 tile: rows of 64 bytes,  position p of row r holds source chunk p ^ ((r >> 2) & 3)
#include "../../mastermetastyletransfer_amd/csrc/common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct PcArgs {
    const bf16_t* x; const float* ln_g; const float* ln_b;
    const bf16_t* w1; const float* b1; const bf16_t* w2; const float* b2;
    const bf16_t* res; bf16_t* out;
    int M, HID, ldx, ldr, ldo; float eps;
};

union Frag { u32x4 u; bf16x8_t b; };

constexpr int RING = 4, AHEAD = 3;
#ifndef PRIO_FC1
#define PRIO_FC1 3
#endif

template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); __builtin_amdgcn_sched_barrier(0); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
#define LDS_READ128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))

#ifdef MLP_CLOCK_PROBE   // diagnostic build only (profiles/probes/clock_probe_pc.hip): per-section s_memtime sums of waves 0 (fc1) and 4 (fc2)
#define PC_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); probe_sec[i] += t_ - probe_last; probe_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PC_STAMP(i) do {} while (0)
#endif

template <int C>
__global__ __launch_bounds__(512) void mlp_pc_kernel(PcArgs p) {
#ifdef MLP_CLOCK_PROBE
    unsigned long long probe_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, probe_last = __builtin_amdgcn_s_memtime();
    const unsigned long long probe_t0 = probe_last;
#endif
    constexpr int KS1 = C / 16;                 // k-steps of fc1 (K = C)
    constexpr int NT2 = C / 32;                 // output-channel tiles of fc2
    constexpr int W1B = 32 * C * 2;             // bytes of a W1 tile  [32 hidden rows][C]
    constexpr int W2B = C * 64;                 // bytes of a W2 tile  [C out rows][32 hidden]
    constexpr int SLOTB = W1B + W2B;
    constexpr int PW = W1B / 1024 / 4;          // one-KB DMA pieces per wave per tick (4 for C = 256, 2 for C = 128); W2B == W1B
    constexpr int CPR = C / 8;                  // 16-B chunks per W1 row
    constexpr int MBOX = 4 * 6144;              // mailbox per band: F0 [2 halves][64 lanes] x 16 B, RAW [2 halves][2][64 lanes] x 16 B

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // ring | mailbox | b1[HID] | b2[C] | ln_g[C] | ln_b[C]
    unsigned char* mbox = lds + RING * SLOTB;
    float* b1s = (float*)(mbox + MBOX);
    float* b2s = b1s + p.HID;
    float* lng = b2s + C;
    float* lnb = lng + C;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool fc1 = wave < 4;
    const int band = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const int m0w = blockIdx.x * 128 + band * 32;
    const int m = m0w + r, mc = m < p.M ? m : p.M - 1;
    const int NT = p.HID / 32;                  // hidden tiles; ticks 0..NT+1 (fc1: MFMAs in 0..NT-1, fc2: MFMAs in 2..NT+1)

    // ---- LDS-DMA of tick t: this wave's PW pieces of its role's tile ---------------------------------------------------
    auto dma_piece = [&](int t, int i) __attribute__((always_inline)) {
#ifdef PC_X_NODMA
        return;
#endif
        unsigned char* slot = lds + (t % RING) * SLOTB;
        const int I = band * PW + i;
        if (fc1) {
            if (t >= NT) return;
            const int row = I * (512 / C) + lane / CPR, pos = lane % CPR;     // piece = 1024 / (2C) rows of the W1 tile
            const int q = pos ^ (row & 15);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.w1 + (size_t)t * 32 * C + row * C + q * 8),
                                             (__attribute__((address_space(3))) void*)(slot + I * 1024), 16, 0, 0);
        } else {
            if (t < 2 || t > NT + 1) return;
            const int row = I * 16 + (lane >> 2), pos = lane & 3;             // piece = 16 rows x 64 B of the W2 tile
            const int q = pos ^ ((row >> 2) & 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.w2 + (t - 2) * 32 + (size_t)row * p.HID + q * 8),
                                             (__attribute__((address_space(3))) void*)(slot + W1B + I * 1024), 16, 0, 0);
        }
    };
    auto issue = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PW; ++i) dma_piece(t, i);
    };
    int t_dma = -1;                                            // tick whose pieces the current tick issues between its MFMAs
    const int fv = fc1 ? 0 : 2, lv = fc1 ? NT - 1 : NT + 1;   // first / last tick this wave has a tile in
    int t_issue = 0;
    for (; t_issue < AHEAD && t_issue <= NT + 1; ++t_issue) issue(t_issue);

    for (int i = tid; i < p.HID; i += 512) b1s[i] = p.b1[i];
    if (tid < C) {
        b2s[tid] = p.b2[tid];
        if (p.ln_g) { lng[tid] = p.ln_g[tid]; lnb[tid] = p.ln_b[tid]; }
    }

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // parameter vectors visible (the first DMA pieces have landed too)

    auto lds_off = [&](const unsigned char* ptr) { return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)ptr; };
    unsigned char* mb_p = mbox + band * 6144 + lane * 16;                    // F0: + half*1024;  RAW: + 2048 + half*2048 + {0, 1024}
    const unsigned mb_w = lds_off(mb_p);

    // ---- GELU of eight values in sixteen slots of (at most) four plain VALU instructions -------------------------------------
    // stage = slot >> 1 works on values 4*(slot&1) .. +3:  y = x*x | fma | x*y | exp2 | 1+y | rcp | x*y | pack.  Sigmoid form of the
    // tanh approximation, x / (1 + 2^(x (a x^2 + b))), |error| vs the erf form <= 5e-4 (below bf16 resolution).  The empty volatile
    // asm pins each slot's results where they are written: the arithmetic is pure and LLVM would sink it past the MFMAs.
    auto gelu_slot = [&](const float (&x)[8], float (&y)[8], unsigned (&u)[4], int slot) __attribute__((always_inline)) {
#ifdef PC_X_NOGELU
        if (slot == 0) { u[0] = __float_as_uint(x[0]); u[1] = __float_as_uint(x[1]); u[2] = __float_as_uint(x[2]); u[3] = __float_as_uint(x[3]); }
        return;
#endif
        const int st = slot >> 1, o = 4 * (slot & 1);
        if (st == 7) {
            u[o / 2] = pack_bf16x2(y[o], y[o + 1]); u[o / 2 + 1] = pack_bf16x2(y[o + 2], y[o + 3]);
            asm volatile("" : "+v"(u[o / 2]), "+v"(u[o / 2 + 1]));
            return;
        }
#pragma unroll
        for (int j = o; j < o + 4; ++j) {
            if (st == 0) y[j] = x[j] * x[j];
            else if (st == 1) y[j] = fmaf(y[j], -0.0713548163f * 1.4426950408889634f, -1.5957691216f * 1.4426950408889634f);
            else if (st == 2) y[j] = x[j] * y[j];
            else if (st == 3) y[j] = __builtin_amdgcn_exp2f(y[j]);
            else if (st == 4) y[j] = 1.0f + y[j];
            else if (st == 5) y[j] = __builtin_amdgcn_rcpf(y[j]);
            else y[j] = x[j] * y[j];
        }
        asm volatile("" : "+v"(y[o]), "+v"(y[o + 1]), "+v"(y[o + 2]), "+v"(y[o + 3]));
    };
    // packed words of a k-step (accumulator k order 8(e>>2) + 4 half + (e&3)) -> B fragment in natural order 8 half + e:
    // half 0 keeps u0,u1 and takes the partner's u0,u1 as words 2,3; half 1 takes the partner's u2,u3 as words 0,1
    auto to_fragment = [&](const unsigned (&u)[4]) __attribute__((always_inline)) {
        auto sw0 = __builtin_amdgcn_permlane32_swap(u[0], u[2], false, false);
        auto sw1 = __builtin_amdgcn_permlane32_swap(u[1], u[3], false, false);
        u32x4 o; o[0] = sw0[0]; o[2] = sw0[1]; o[1] = sw1[0]; o[3] = sw1[1];
        return o;
    };
    constexpr int SPM1 = 16 / KS1;              // GELU slots per fc1 MFMA (1 for C = 256, 2 for C = 128)
    constexpr int SPM2 = 16 / (2 * NT2);        // GELU slots per fc2 MFMA

    // ---- fc1 tick t: MFMAs of tile t (t < NT) || GELU of k-step 0 of tile t-1 (t >= 1) ---------------------------------------
    // hp: b1 + W1.X of the previous tile (accumulator registers 0..7 = k-step 0, 8..15 = k-step 1)
    auto fc1_tick = [&](const Frag (&xf)[KS1], float (&hp)[8], const unsigned char* slot, int t) __attribute__((always_inline)) {
        const bool mm = t < NT, gg = t >= 1;
        float y[8]; unsigned u[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = 0.f;
        f32x16 a0, a1;
        if (mm) {
            const unsigned base = lds_off(slot) + r * (C * 2);
            const int sw = r & 15;
            u32x4 fr[4], bv[4];                                               // fragment reads four k-steps ahead of their MFMA
#pragma unroll
            for (int s = 0; s < 4; ++s) LDS_READ128(fr[s], base + (((2 * s + h) ^ sw) << 4), 0);
            {
                const unsigned bp = lds_off((const unsigned char*)(b1s + t * 32 + 4 * h));   // accumulator register e = hidden row (e&3) + 8(e>>2) + 4h
                LDS_READ128(bv[0], bp, 0); LDS_READ128(bv[1], bp, 32); LDS_READ128(bv[2], bp, 64); LDS_READ128(bv[3], bp, 96);
                wait_lgkm<0>();
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a0[4 * g + e] = __uint_as_float(bv[g][e]); a1[4 * g + e] = 0.f; }
            }
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
#ifdef PC_X_NOREAD
                if (s == 0) wait_lgkm<0>();
#else
                if (s + 4 <= KS1) wait_lgkm<3>(); else if (s + 3 == KS1) wait_lgkm<2>(); else if (s + 2 == KS1) wait_lgkm<1>(); else wait_lgkm<0>();
#endif
                Frag f; f.u = fr[s & 3];
                if (s & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b, xf[s].b, a1, 0, 0, 0);
                else a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b, xf[s].b, a0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#ifndef PC_X_NOREAD
                if (s + 4 < KS1) LDS_READ128(fr[s & 3], base + (((2 * (s + 4) + h) ^ sw) << 4), 0);
#endif
                if (gg) {
#pragma unroll
                    for (int i = 0; i < SPM1; ++i) gelu_slot(hp, y, u, s * SPM1 + i);
                }
                if (s % (KS1 / PW) == 1 && t_dma >= 0) dma_piece(t_dma, s / (KS1 / PW));
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) gelu_slot(hp, y, u, i);
            if (t_dma >= 0) issue(t_dma);
        }
        PC_STAMP(4);
        if (gg) *(u32x4*)(mb_p + ((t - 1) & 1) * 1024) = to_fragment(u);     // B fragment 0 of tile t-1
        if (mm) {
            float hq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { hp[j] = a0[j] + a1[j]; hq[j] = a0[8 + j] + a1[8 + j]; }
            unsigned char* rp = mb_p + 2048 + (t & 1) * 2048;                // raw k-step-1 values of tile t for the fc2 wave
            *(f32x4*)rp = f32x4{hq[0], hq[1], hq[2], hq[3]};
            *(f32x4*)(rp + 1024) = f32x4{hq[4], hq[5], hq[6], hq[7]};
        }
    };

    // ---- fc2 tick t: MFMAs of tile t-2 (2 <= t) || GELU of k-step 1 of tile t-1 (1 <= t <= NT) -------------------------------
    auto fc2_tick = [&](f32x16 (&oacc)[NT2], u32x4& f1, const unsigned char* slot, int t) __attribute__((always_inline)) {
        const bool mm = t >= 2, gg = t >= 1 && t <= NT;
        float x[8], y[8]; unsigned u[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[j] = 0.f; y[j] = 0.f; }
        // raw k-step-1 values of tile t-1: issued first, so every later counted wait covers them (a garbage read when !gg is unused)
        u32x4 rw0, rw1;
        LDS_READ128(rw0, mb_w + 2048 + ((t - 1) & 1) * 2048, 0);
        LDS_READ128(rw1, mb_w + 2048 + ((t - 1) & 1) * 2048, 1024);
        if (!mm) wait_lgkm<0>();
        if (mm) {
            const unsigned base = lds_off(slot) + W1B + r * 64;
            const int sw = (r >> 2) & 3;
            u32x4 hb0, fr[2][NT2];
            LDS_READ128(hb0, mb_w + (t & 1) * 1024, 0);                       // fragment 0 of tile t-2 (half (t-2)&1)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) {
                    const unsigned ad = base + (((2 * ks + h) ^ sw) << 4);
#ifdef PC_X_NOREAD
                    if (ks + nt > 0) { fr[ks][nt] = u32x4{1u, 2u, 3u, 4u}; continue; }
#endif
                    if (nt == 0) LDS_READ128(fr[ks][0], ad, 0);
                    else if (nt == 1) LDS_READ128(fr[ks][1], ad, 2048);
                    else if (nt == 2) LDS_READ128(fr[ks][2], ad, 4096);
                    else if (nt == 3) LDS_READ128(fr[ks][3], ad, 6144);
                    else if (nt == 4) LDS_READ128(fr[ks][NT2 > 4 ? 4 : 0], ad, 8192);
                    else if (nt == 5) LDS_READ128(fr[ks][NT2 > 4 ? 5 : 0], ad, 10240);
                    else if (nt == 6) LDS_READ128(fr[ks][NT2 > 4 ? 6 : 0], ad, 12288);
                    else LDS_READ128(fr[ks][NT2 > 4 ? 7 : 0], ad, 14336);
                }
            // 1 + 2*NT2 asm reads in flight, returned in order; MFMA (ks, nt) needs hb0 and fr[ks][nt]
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) {
#ifdef PC_X_NOREAD
                    constexpr int TOTAL = 0;
                    const int left = 0;
#else
                    constexpr int TOTAL = 1 + 2 * NT2;
                    const int left = TOTAL - (1 + ks * NT2 + nt + 1);
#endif
                    if (left >= 15) wait_lgkm<15>(); else if (left == 14) wait_lgkm<14>(); else if (left == 13) wait_lgkm<13>();
                    else if (left == 12) wait_lgkm<12>(); else if (left == 11) wait_lgkm<11>(); else if (left == 10) wait_lgkm<10>();
                    else if (left == 9) wait_lgkm<9>(); else if (left == 8) wait_lgkm<8>(); else if (left == 7) wait_lgkm<7>();
                    else if (left == 6) wait_lgkm<6>(); else if (left == 5) wait_lgkm<5>(); else if (left == 4) wait_lgkm<4>();
                    else if (left == 3) wait_lgkm<3>(); else if (left == 2) wait_lgkm<2>(); else if (left == 1) wait_lgkm<1>();
                    else wait_lgkm<0>();
                    Frag f, b; f.u = fr[ks][nt]; b.u = ks == 0 ? hb0 : f1;
                    oacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b, b.b, oacc[nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    const int mi = ks * NT2 + nt;
                    if (mi == 0) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) { x[j] = __uint_as_float(rw0[j]); x[4 + j] = __uint_as_float(rw1[j]); }
                    }
                    if (gg) {
#pragma unroll
                        for (int i = 0; i < SPM2; ++i) gelu_slot(x, y, u, mi * SPM2 + i);
                    }
                    if (mi % (2 * NT2 / PW) == 1 && t_dma >= 0) dma_piece(t_dma, mi / (2 * NT2 / PW));
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { x[j] = __uint_as_float(rw0[j]); x[4 + j] = __uint_as_float(rw1[j]); }
            if (gg) {
#pragma unroll
                for (int i = 0; i < 16; ++i) gelu_slot(x, y, u, i);
            }
            if (t_dma >= 0) issue(t_dma);
        }
        if (gg) f1 = to_fragment(u);                                          // fragment 1 of tile t-1, used by the next tick's MFMAs
    };

    // ---- tick synchronisation, common to both roles (every wave executes it NT+1 times) -----------------------------------------
    auto tick_sync = [&](int tau) __attribute__((always_inline)) {
        // my DMA pieces of tick tau have landed: allow only the pieces of my later issued ticks to be outstanding
        const int hi = (t_issue - 1 < lv ? t_issue - 1 : lv), lo = (tau + 1 > fv ? tau + 1 : fv);
        const int later = hi - lo + 1;
        PC_STAMP(0);                                               // 0: work of the previous tick
        if (later >= 2) wait_vm<2 * PW>(); else if (later == 1) wait_vm<PW>(); else wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // my mailbox writes / fragment reads of the previous tick are done
        PC_STAMP(1);                                               // 1: DMA (+ LDS) wait
        __builtin_amdgcn_s_barrier();
        PC_STAMP(2);                                               // 2: barrier
        t_dma = t_issue <= NT + 1 ? t_issue++ : -1;                    // its pieces go out between this tick's MFMAs, into the slot of tick
        PC_STAMP(3);                                               // tau-1, which every wave has left
        return (const unsigned char*)(lds + (tau % RING) * SLOTB);
    };

    // The roles run separate loops (same number of barriers) so that fc1's X fragments and fc2's accumulators share registers.
    // fc1 is the longer job per tick (MFMAs + GELU): it gets issue priority, so its MFMAs go first and fc2's run under its GELU.
    if (fc1) {
        Frag xf[KS1];
        {
            const bf16_t* xp = p.x + (size_t)mc * p.ldx + 8 * h;
    #pragma unroll
            for (int s = 0; s < KS1; ++s) xf[s].u = *(const u32x4*)(xp + 16 * s);
        }
        if (p.ln_g) {
            float sum = 0.f;
    #pragma unroll
            for (int s = 0; s < KS1; ++s)
    #pragma unroll
                for (int e = 0; e < 4; ++e) sum += __uint_as_float(xf[s].u[e] << 16) + __uint_as_float(xf[s].u[e] & 0xffff0000u);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum / (float)C;
            float sq = 0.f;
    #pragma unroll
            for (int s = 0; s < KS1; ++s)
    #pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d0 = __uint_as_float(xf[s].u[e] << 16) - mean, d1 = __uint_as_float(xf[s].u[e] & 0xffff0000u) - mean;
                    sq += d0 * d0 + d1 * d1;
                }
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = rsqrtf(sq / (float)C + p.eps);
    #pragma unroll
            for (int s = 0; s < KS1; ++s) {
                const float* gp = lng + 16 * s + 8 * h; const float* bp = lnb + 16 * s + 8 * h;
                const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 4), b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
                float o[8];
    #pragma unroll
                for (int e = 0; e < 2; ++e) {
                    o[2 * e] = (__uint_as_float(xf[s].u[e] << 16) - mean) * rstd * g0[2 * e] + b0[2 * e];
                    o[2 * e + 1] = (__uint_as_float(xf[s].u[e] & 0xffff0000u) - mean) * rstd * g0[2 * e + 1] + b0[2 * e + 1];
                    o[4 + 2 * e] = (__uint_as_float(xf[s].u[2 + e] << 16) - mean) * rstd * g1[2 * e] + b1[2 * e];
                    o[5 + 2 * e] = (__uint_as_float(xf[s].u[2 + e] & 0xffff0000u) - mean) * rstd * g1[2 * e + 1] + b1[2 * e + 1];
                }
    #pragma unroll
                for (int e = 0; e < 4; ++e) xf[s].u[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]);
            }
        }
        float hp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) hp[j] = 0.f;
        for (int tau = 0; tau <= NT + 1; ++tau) {
            const unsigned char* slot = tick_sync(tau);
            if (tau <= NT) fc1_tick(xf, hp, slot, tau);
            else if (t_dma >= 0) issue(t_dma);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                               // every wave is done with the ring
#ifdef MLP_CLOCK_PROBE
        if (tid == 0 && g_probe) { for (int i = 0; i < 8; ++i) g_probe[(blockIdx.x * 2) * 9 + i] = probe_sec[i]; g_probe[(blockIdx.x * 2) * 9 + 8] = __builtin_amdgcn_s_memtime() - probe_t0; }
#endif
        return;
    }

    f32x16 oacc[NT2];
#pragma unroll
    for (int t = 0; t < NT2; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[t][e] = 0.f;
    u32x4 f1 = {0u, 0u, 0u, 0u};
    for (int tau = 0; tau <= NT + 1; ++tau) {
        const unsigned char* slot = tick_sync(tau);
        if (tau >= 1) fc2_tick(oacc, f1, slot, tau);
        else if (t_dma >= 0) issue(t_dma);
    }
    // ---- epilogue (fc2 waves): O^T tile -> [32 tokens][C] fp32 through the idle ring, then whole rows: + b2, + residual, bf16 ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                   // every wave is done with the ring
    {
        float* stage = (float*)(lds + band * (32 * C * 4));        // 4 x 32 KB (C = 256) / 4 x 16 KB (C = 128) = the ring
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
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // wave-private tile
        constexpr int RPI = 64 / CHUNKS;                            // rows per wave instruction (1 for C=256, 2 for C=128)
        const int q = lane % CHUNKS, rsub = lane / CHUNKS;
        const f32x4 bv = *(const f32x4*)(b2s + 4 * q);
        u32x2 rr[32 / RPI];
        if (p.res) {
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
            if (p.res) {
                v[0] += __uint_as_float(rr[it][0] << 16); v[1] += __uint_as_float(rr[it][0] & 0xffff0000u);
                v[2] += __uint_as_float(rr[it][1] << 16); v[3] += __uint_as_float(rr[it][1] & 0xffff0000u);
            }
            u32x2 o; o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]);
            if (mm < p.M) *(u32x2*)(p.out + (size_t)mm * p.ldo + 4 * q) = o;
        }
    }
#ifdef MLP_CLOCK_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PC_STAMP(5);                                                   // 5: fc2 epilogue
    if (tid == 256 && g_probe) { for (int i = 0; i < 8; ++i) g_probe[(blockIdx.x * 2 + 1) * 9 + i] = probe_sec[i]; g_probe[(blockIdx.x * 2 + 1) * 9 + 8] = __builtin_amdgcn_s_memtime() - probe_t0; }
#endif
}

template <int C>
int launch(const PcArgs& p, hipStream_t s) {
    const size_t lds = (size_t)RING * (32 * C * 2 + C * 64) + 4 * 6144 + (size_t)(p.HID + 3 * C) * sizeof(float);
    if (lds > 160 * 1024) return MASTER_ERR_UNSUPPORTED;
    static LdsLimit lds_limit;
    lds_limit.ensure((const void*)mlp_pc_kernel<C>, (int)((int)lds));
    hipLaunchKernelGGL((mlp_pc_kernel<C>), dim3((p.M + 127) / 128), dim3(512), lds, s, p);
    return master_check_launch("mlp_pc_bf16");
}

}  // namespace

// MASTER_ERR_UNSUPPORTED when the shape is outside what this form covers (the caller then uses mlp_bf16)
int mlp_pc_bf16(const bf16_t* x, const float* ln_g, const float* ln_b, const bf16_t* w1, const float* b1, const bf16_t* w2,
                const float* b2, const bf16_t* res, bf16_t* out, int M, int C, int HID, int ldx, int ldr, int ldo, float eps,
                hipStream_t s) {
    if ((C != 128 && C != 256) || HID % 32 != 0 || HID < 64) return MASTER_ERR_UNSUPPORTED;
    PcArgs p{x, ln_g, ln_b, w1, b1, w2, b2, res, out, M, HID, ldx, ldr, ldo, eps};
    return C == 256 ? launch<256>(p, s) : launch<128>(p, s);
}
