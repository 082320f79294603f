// Window attention + output projection + residual in ONE launch, bf16, C = 256 / 8 heads (the five single-value-stream attention
// calls of a step: Swin stage 2 x2, StyleEncoder Key and Scale/Shift (layer 1: Scale is Shift), StyleDecoder self attention):
//
//     out[token] = res[token] + Wproj . concat_heads( softmax(q k^T * scale + bias + mask) v ) + bproj
//
// (reference style_transformer.py:120-168: heads split :120-122, scale :127, QK^T + relative-position bias :130-132, shift mask
//  :134-150, softmax :152, AV :155, proj :156, window reverse / roll back / un-pad :160-168; the residual is the caller's
//  `x + StochasticDepth(attn(...))`, :391, identity in eval mode.)
//
// r02 ran this as two launches - window_attn_pw_kernel (half-window workgroups) writing the attention output to HBM and a
// token-stationary Linear reading it back (17 us each at bs = 32: one 128-token tile per CU, pure latency).  Here a workgroup is a
// WHOLE window and stays resident on its CU.  What decides the shape is the projection weight: 128 KB of bf16 that every window
// needs.  A first version (8 waves = 8 heads, 256 registers each) had no room to keep its rows and re-fetched them from L2 per
// window - 128 KB per window through a ~70 GB/s-per-CU path cost as much as the separate launch saved (measured: 2.07 ms per step
// fused against 2.06 split).  The register file is the only place 128 KB fit next to four 32-KB tiles in LDS, so the workgroup is
// FOUR waves, one per SIMD with the whole 512-register file: a wave owns two heads and, for the whole kernel, the 64 rows of Wproj
// it projects (128 registers), the relative-position bias of its heads as MFMA fragments (64) and the proj bias (32).  The
// attention core of window_attn_pw_bf16.hip is issue-bound (one wave per SIMD runs a head-window as fast as two waves run two),
// so two heads in sequence cost what two co-resident waves did.  Per window:
//   1. every wave pulls the Q, K and V fragments of its two heads out of LDS (16-B chunks source-swizzled by row & 15; V through
//      ds_read_b64_tr_b16), barrier, and the NEXT window's tiles are gathered by LDS-DMA through the pad / roll / partition table
//      (padded slots fetch the projection-bias row) while this window computes;
//   2. per head: S^T = K Q^T (+ rel-pos bias through an identity MFMA, shift mask), softmax with the query on the lane,
//      O^T = V^T P^T from the packed accumulator tile, normalised O rows into the shared O tile (bf16 [slot][256], swizzled);
//   3. barrier; out^T[64 channels of this wave][64 slots] = Wproj rows . O^T: 64 MFMAs, O fragments streamed from LDS;
//   4. barrier; the projected tile goes through LDS as fp32 [49 slots][256 channels] (rows padded to 1040 B: conflict-free 16-B writes
//      from the accumulator layout), barrier; every wave then takes whole token rows: 16-B coalesced residual loads and stores.
//      (Storing from the accumulator layout - lane = slot, 8-B pieces 512 B apart - costs ~250 clocks per wave instruction:
//      32 scattered loads + stores per window were 8 k of the first version's 21 k clocks per window.)
// Five barriers per window; the loop's only ordinary global loads are the residual rows (issued before the attention math).
//
// MEASURED OUTCOME (r03, MI355X; profiles/r03_attn_proj_probe_mi355x.txt, profiles/tools/attn_proj_bench.py): correct, and NOT faster
// than the two launches - 47 us against 24 + 15 at 800 windows, 63-69 against 62-64 at 1 600.  In-kernel clocks per window (~20 k):
// 2 x 3.6 k head math, 3.9 k issuing the next window's 24 LDS-DMA pieces and 7 residual loads (~115-240 clocks per wave
// instruction: the CU's memory pipeline carries 160 KB per window and one in-order wave per SIMD sits through every queue stall),
// 2.3 k proj (at its MFMA floor), 2.9 k row pass, 1.0 k stage, and 22 k of set-up per workgroup.  The attention core wants two
// waves per SIMD to hide memory-instruction issue, the projection wants the register file of one: the split form gives each what
// it needs.  The kernel therefore ships OFF by default (MASTER_ATTN_PROJ=1 selects it; tests cover it either way).
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int HEADS = 8, C = 256;
constexpr int ROWB = C * 2;                      // bytes of a token row: 512
constexpr int TILE = 64 * ROWB;                  // one 64-slot tile: 32 KB
constexpr int NWAVES = 4, HPW = HEADS / NWAVES;   // waves per workgroup, heads per wave
constexpr int NTHREADS = NWAVES * 64;
constexpr int SROW = C * 4 + 16;                 // bytes of a staged fp32 output row (padded: 8 consecutive slots hit 8 distinct 16-B bank groups)
constexpr int NSLOT = 49;                        // rows staged: the real slots of a 7x7 window (slots >= N are tile padding)
constexpr int STAGE = ((NSLOT * SROW + 1023) / 1024) * 1024;      // 51 200 B; its first 32 KB double as the bf16 O tile
constexpr float LOG2E = 1.4426950408889634f;

struct ApArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v;
    int ldq, ldk, ldv;
    const bf16_t* k_pad; const bf16_t* v_pad; int k_pad_bs, v_pad_bs;
    const float* bias; const int32_t* src; const uint8_t* region;
    const bf16_t* wproj; const float* bproj; const bf16_t* res; int ldr;
    bf16_t* out; int ldo;
    int B, T, nW, N; float scale_log2e;
};

union Frag { u32x4 u; bf16x8_t b; };

#define LDS_READ128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define LGKM_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ __forceinline__ unsigned lds_addr(const unsigned char* ptr) { return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)ptr; }

#ifdef ATTN_PROJ_PROBE   // diagnostic build only (profiles/probes/attn_proj_probe.hip): s_memtime sections of wave 0
#define PROBE_MARK() do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); probe_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define PROBE_ADD(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); probe_sec[i] += t_ - probe_last; probe_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PROBE_MARK() do {} while (0)
#define PROBE_ADD(i) do {} while (0)
#endif

__global__ __launch_bounds__(NTHREADS, 1) void window_attn_proj_kernel(ApArgs p) {
#ifdef ATTN_PROJ_PROBE
    const unsigned long long probe_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long probe_sec[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, probe_last = 0;
    unsigned long long probe_loop0 = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // Q | K | V tiles | O tile / fp32 output stage | src[nW][64] i32, region[nW][64] u8
    unsigned char* Qs = lds;
    unsigned char* Ks = lds + TILE;
    unsigned char* Vs = lds + 2 * TILE;
    unsigned char* Os = lds + 3 * TILE;
    int32_t* srcs = (int32_t*)(lds + 3 * TILE + STAGE);
    uint8_t* regs = (uint8_t*)(srcs + p.nW * 64);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // heads 2 wave, 2 wave + 1; output channels [64 wave, 64 wave + 64)
    const int r = lane & 31, h = lane >> 5;
    const int nwin = p.B * p.nW;

    for (int i = tid; i < p.nW * 64; i += NTHREADS) { srcs[i] = p.src[i]; regs[i] = p.region ? p.region[i] : 0; }
    // relative-position bias of this wave's heads as MFMA fragments, added through an identity fragment (window_attn_pw_bf16.hip)
    Frag bfr[HPW][2][2][2];                                               // [head][key tile][query tile][k-step]
    {
        const float inv_scale = 1.0f / p.scale_log2e * LOG2E;
#pragma unroll
        for (int hh = 0; hh < HPW; ++hh) {
            const float* bias = p.bias + (size_t)(wave * HPW + hh) * 64 * 64;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        float v[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] = bias[(q2 * 32 + 16 * ks + 8 * h + j) * 64 + k2 * 32 + r] * inv_scale;
#pragma unroll
                        for (int j = 0; j < 4; ++j) bfr[hh][k2][q2][ks].u[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
                    }
        }
    }
    Frag idf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q0 = 16 * ks + 8 * h + 2 * j;
            idf[ks].u[j] = (q0 == r ? 0x00003f80u : 0u) | (q0 + 1 == r ? 0x3f800000u : 0u);
        }
    // this wave's 64 rows of Wproj as A fragments (row = output channel, 8 consecutive input channels per lane and k-step), and the
    // proj bias of accumulator register e: channel 64 wave + 32 rt + (e&3) + 8(e>>2) + 4h.  Loaded once, 160 registers for the kernel.
    Frag wp[2][16];
    float bp[2][16];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const bf16_t* wrow = p.wproj + (size_t)(wave * 64 + rt * 32 + r) * C + 8 * h;
#pragma unroll
        for (int s = 0; s < 16; ++s) wp[rt][s].u = *(const u32x4*)(wrow + 16 * s);
#pragma unroll
        for (int e = 0; e < 16; ++e) bp[rt][e] = p.bproj[wave * 64 + rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- LDS-DMA gather of one window: every tile = 64 rows x 512 B = 32 pieces of 1 KB (2 rows each); wave w moves pieces 8w..8w+7
    //      of each tile.  Position c of a row (32 chunks of 16 B) holds source chunk c ^ (row & 15). ---------------------------------
    const bool same_ld = p.ldq == p.ldk && p.ldq == p.ldv;
    auto gather = [&](int win, int i0, int i1) __attribute__((always_inline)) {
        const int b = win / p.nW, w = win - b * p.nW;
        const int32_t* src = srcs + w * 64;
        const unsigned tok0 = (unsigned)b * (unsigned)p.T;
        const unsigned kpad0 = (unsigned)b * (unsigned)p.k_pad_bs * 2u, vpad0 = (unsigned)b * (unsigned)p.v_pad_bs * 2u;     // bytes
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int I = wave * 8 + i;
            const int row = 2 * I + (lane >> 5);
            const unsigned qcb = (unsigned)((lane & 31) ^ (row & 15)) * 16u;           // byte offset of the lane's source chunk in a token row
            const int idx = row < p.N ? src[row] : -1;
            const bool real = idx >= 0;
            const unsigned tk = tok0 + (real ? (unsigned)idx : 0u);
            const unsigned oq = __umul24(tk, (unsigned)p.ldq * 2u) + qcb;               // padded query rows are never stored: any row will do
            const unsigned ok = same_ld ? oq : __umul24(tk, (unsigned)p.ldk * 2u) + qcb;
            const unsigned ov = same_ld ? oq : __umul24(tk, (unsigned)p.ldv * 2u) + qcb;
            const char* sq = (const char*)p.q + oq;
            const char* sk = real ? (const char*)p.k + ok : (const char*)p.k_pad + (kpad0 + qcb);
            const char* sv = real ? (const char*)p.v + ov : (const char*)p.v_pad + (vpad0 + qcb);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sq, (__attribute__((address_space(3))) void*)(Qs + I * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sk, (__attribute__((address_space(3))) void*)(Ks + I * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sv, (__attribute__((address_space(3))) void*)(Vs + I * 1024), 16, 0, 0);
        }
    };

    int win = blockIdx.x;
    if (win < nwin) gather(win, 0, 8);
#ifdef ATTN_PROJ_PROBE
    probe_loop0 = __builtin_amdgcn_s_memtime();
#endif

    const int g16 = (lane >> 4) & 1, i16 = lane & 15;
    const float mask_raw = -100.0f * LOG2E / p.scale_log2e;              // the -100 of the shift mask in raw q.k units
    Frag ones; ones.u = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};      // bf16 1.0 x 8
    union VF { s16x4 v[2]; bf16x8_t b; };
    for (; win < nwin; win += gridDim.x) {
        const int b = win / p.nW, w = win - b * p.nW;
        const int32_t* src = srcs + w * 64;
        const size_t tok0 = (size_t)b * p.T;

        PROBE_MARK();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // this window's tiles have landed (every wave waited for its own pieces)
        PROBE_ADD(0);

        // ---- 1. fragments of this wave's two heads out of LDS --------------------------------------------------------------
        Frag qf[HPW][2][2], kf[HPW][2][2];
        VF vf[HPW][2][2];
#pragma unroll
        for (int hh = 0; hh < HPW; ++hh) {
            const int head = wave * HPW + hh;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int row = t * 32 + r;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int pos = ((head * 4 + 2 * s + h) ^ (row & 15)) << 4;   // chunk head*4 + 2s + h of the row's 32 chunks
                    qf[hh][t][s].u = *(const u32x4*)(Qs + row * ROWB + pos);
                    kf[hh][t][s].u = *(const u32x4*)(Ks + row * ROWB + pos);
                }
            }
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int row = 32 * k2 + 16 * s + 8 * jj + 4 * h + (i16 >> 2);
                        const int chunk = head * 4 + 2 * g16 + ((i16 & 3) >> 1);
                        const int off = row * ROWB + ((chunk ^ (row & 15)) << 4) + 8 * (i16 & 1);
                        vf[hh][k2][s].v[jj] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vs + off));
                    }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // every wave has its fragments: the tiles are free
        PROBE_ADD(1);
        // next window's rows fly under this window's math.  (Dealing the 24 pieces out over the head math instead of issuing them as a
        // burst moves their cost without reducing it - profiles/r03_attn_proj_probe_mi355x.txt: ~115 clocks per piece either way.)
        if (win + (int)gridDim.x < nwin) gather(win + gridDim.x, 0, 8);

        // residual rows this wave will store (step 4: slots 8 it + 2 wave + (lane >> 5), 8 channels per lane): in flight under the attention math
        u32x4 rsd[7];
        int ridx[7];
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int slot = it * 8 + wave * 2 + (lane >> 5);
            ridx[it] = slot < p.N ? src[slot] : -1;
            rsd[it] = u32x4{0u, 0u, 0u, 0u};
            if (p.res && ridx[it] >= 0) rsd[it] = *(const u32x4*)(p.res + (tok0 + ridx[it]) * p.ldr + (lane & 31) * 8);
        }
        bool mixed = false;
        if (p.region) {
            const uint8_t* reg = regs + w * 64;
            const unsigned r0 = reg[0];
            mixed = __builtin_amdgcn_ballot_w64(reg[lane] != r0) != 0ull;
        }
        PROBE_ADD(2);

#pragma unroll
        for (int hh = 0; hh < HPW; ++hh) {
            const int head = wave * HPW + hh;
            // ---- 2. S^T[key][query] = K Q^T, + bias (identity MFMA) + shift mask ---------------------------------------------------
            f32x16 st[2][2];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) st[k2][q2][e] = 0.f;
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        st[k2][q2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[hh][k2][s].b, qf[hh][q2][s].b, st[k2][q2], 0, 0, 0);
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        st[k2][q2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[hh][k2][q2][ks].b, idf[ks].b, st[k2][q2], 0, 0, 0);
                }
            if (mixed) {
                const uint8_t* reg = regs + w * 64;
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const unsigned rq = reg[q2 * 32 + r];
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const unsigned rk4 = *(const unsigned*)(reg + k2 * 32 + 8 * g + 4 * h);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (((rk4 >> (8 * e)) & 0xffu) != rq) st[k2][q2][4 * g + e] += mask_raw;
                        }
                }
            }
            // ---- softmax over keys (query on the lane), probabilities packed as the B operand of P.V -------------------------------
            float inv[2];
            Frag pf[2][2][2];
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                float mx = st[0][q2][0];
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int e = 0; e < 16; ++e) mx = fmaxf(mx, st[k2][q2][e]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float nmx = -mx * p.scale_log2e;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            pf[q2][k2][s].u[j] = pack_bf16x2(__builtin_amdgcn_exp2f(fmaf(st[k2][q2][8 * s + 2 * j], p.scale_log2e, nmx)),
                                                             __builtin_amdgcn_exp2f(fmaf(st[k2][q2][8 * s + 2 * j + 1], p.scale_log2e, nmx)));
                f32x16 rs;
#pragma unroll
                for (int e = 0; e < 16; ++e) rs[e] = 0.f;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int s = 0; s < 2; ++s) rs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones.b, pf[q2][k2][s].b, rs, 0, 0, 0);
                inv[q2] = __builtin_amdgcn_rcpf(rs[0]);
            }
            // ---- O^T[d][query] = V^T P^T, normalised, into the shared O tile: O[slot][head*32 + d], 8-B pieces d = 8g + 4h + (0..3) ------
            f32x16 ot[2];
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                for (int e = 0; e < 16; ++e) ot[q2][e] = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2)
                        ot[q2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[hh][k2][s].b, pf[q2][k2][s].b, ot[q2], 0, 0, 0);
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int slot = q2 * 32 + r;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 o;
                    o[0] = pack_bf16x2(ot[q2][4 * g] * inv[q2], ot[q2][4 * g + 1] * inv[q2]);
                    o[1] = pack_bf16x2(ot[q2][4 * g + 2] * inv[q2], ot[q2][4 * g + 3] * inv[q2]);
                    *(u32x2*)(Os + slot * ROWB + (((head * 4 + g) ^ (slot & 15)) << 4) + 8 * h) = o;
                }
            }
            PROBE_ADD(3 + hh);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // all heads' O rows are in LDS (the previous window's proj reads ended two barriers ago)

        PROBE_ADD(5);
        // ---- 3. proj: out^T[n][slot] for this wave's 64 output channels, O fragments streamed from LDS ----------------------------
        f32x16 pt[2][2];                                                   // [row tile][slot tile]
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) pt[rt][t][e] = bp[rt][e];
        {
            const unsigned ob = lds_addr(Os + r * ROWB);
            const int xsw = (h ^ (r & 15)) << 4;                           // chunk 2s + h of row r (and r + 32: same low bits) at position (2s + h) ^ (r & 15)
            u32x4 oa[2][2];
            LDS_READ128(oa[0][0], ob + (0 ^ xsw), 0); LDS_READ128(oa[0][1], ob + (0 ^ xsw), 16384);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + 1 < 16) {
                    const unsigned ad = ob + ((32 * (s + 1)) ^ xsw);
                    LDS_READ128(oa[(s + 1) & 1][0], ad, 0); LDS_READ128(oa[(s + 1) & 1][1], ad, 16384);
                    LGKM_WAIT(2);
                } else { LGKM_WAIT(0); }
                Frag o0, o1; o0.u = oa[s & 1][0]; o1.u = oa[s & 1][1];
                pt[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[0][s].b, o0.b, pt[0][0], 0, 0, 0);
                pt[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[0][s].b, o1.b, pt[0][1], 0, 0, 0);
                pt[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[1][s].b, o0.b, pt[1][0], 0, 0, 0);
                pt[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[1][s].b, o1.b, pt[1][1], 0, 0, 0);
            }
        }
        // ---- 4. projected tile -> LDS (fp32 [slot][256], rows of SROW bytes) -> whole rows: + residual, 16-B stores ------------------
        PROBE_ADD(6);
        __builtin_amdgcn_s_barrier();                                      // every wave has read its O fragments: the stage may overwrite the O tile
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int slot = t * 32 + r;
            if (slot < NSLOT) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = pt[rt][t][4 * g + e];
                        *(f32x4*)(Os + slot * SROW + (wave * 64 + rt * 32 + 8 * g + 4 * h) * 4) = v;
                    }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        PROBE_ADD(7);
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            if (ridx[it] < 0) continue;
            const int slot = it * 8 + wave * 2 + (lane >> 5);
            const unsigned char* sp = Os + slot * SROW + (lane & 31) * 32;
            const f32x4 v0 = *(const f32x4*)sp, v1 = *(const f32x4*)(sp + 16);
            const u32x4 a = rsd[it];
            u32x4 o;
            o[0] = pack_bf16x2(v0[0] + __uint_as_float(a[0] << 16), v0[1] + __uint_as_float(a[0] & 0xffff0000u));
            o[1] = pack_bf16x2(v0[2] + __uint_as_float(a[1] << 16), v0[3] + __uint_as_float(a[1] & 0xffff0000u));
            o[2] = pack_bf16x2(v1[0] + __uint_as_float(a[2] << 16), v1[1] + __uint_as_float(a[2] & 0xffff0000u));
            o[3] = pack_bf16x2(v1[2] + __uint_as_float(a[3] << 16), v1[3] + __uint_as_float(a[3] & 0xffff0000u));
            *(u32x4*)(p.out + (tok0 + ridx[it]) * p.ldo + (lane & 31) * 8) = o;
        }
        PROBE_ADD(8);
    }
#ifdef ATTN_PROJ_PROBE
    if (tid == 0 && g_probe) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        g_probe[blockIdx.x * 13] = t1 - probe_t0;
        g_probe[blockIdx.x * 13 + 1] = (unsigned long long)((nwin - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);
        g_probe[blockIdx.x * 13 + 2] = probe_loop0 - probe_t0;
        for (int i = 0; i < 10; ++i) g_probe[blockIdx.x * 13 + 3 + i] = probe_sec[i];
    }
#endif
}

}  // namespace

// returns MASTER_ERR_UNSUPPORTED when the geometry is outside this kernel (the caller then runs attention and projection separately)
int window_attn_proj_bf16(const bf16_t* q, const bf16_t* k, const bf16_t* v, int ldq, int ldk, int ldv,
                          const bf16_t* k_pad, const bf16_t* v_pad, int k_pad_bs, int v_pad_bs,
                          const float* bias, const int32_t* src, const uint8_t* region,
                          const bf16_t* wproj, const float* bproj, const bf16_t* res, int ldr, bf16_t* out, int ldo,
                          int B, int T, int heads, int nW, int N, float scale, hipStream_t s) {
    if (heads != HEADS) return MASTER_ERR_UNSUPPORTED;
    // the gather's 24-bit x 24-bit row offsets: token count, row pitches in bytes and every row offset below 2^24 / 2^32
    if ((long long)B * T >= (1ll << 24) || ldq * 2 >= (1 << 24) || ldk * 2 >= (1 << 24) || ldv * 2 >= (1 << 24) ||
        (long long)B * T * 2 * (ldq > ldk ? (ldq > ldv ? ldq : ldv) : (ldk > ldv ? ldk : ldv)) >= (1ll << 32) ||
        (long long)B * 2 * (k_pad_bs > v_pad_bs ? k_pad_bs : v_pad_bs) >= (1ll << 32)) return MASTER_ERR_UNSUPPORTED;
    if (N > NSLOT || ((uintptr_t)out & 15) || (ldo & 7) || (res && (((uintptr_t)res & 15) || (ldr & 7)))) return MASTER_ERR_UNSUPPORTED;
    const int lds = 3 * TILE + STAGE + nW * 64 * 5;
    if (lds > 160 * 1024) return MASTER_ERR_UNSUPPORTED;
    ApArgs p{q, k, v, ldq, ldk, ldv, k_pad, v_pad, k_pad_bs, v_pad_bs, bias, src, region, wproj, bproj, res, ldr, out, ldo,
             B, T, nW, N, scale * LOG2E};
    int grid = B * nW;
    if (grid > 256) grid = 256;                      // persistent: one workgroup (window) per CU
    static LdsLimit lim;
    lim.ensure((const void*)window_attn_proj_kernel, lds);
    hipLaunchKernelGGL(window_attn_proj_kernel, dim3(grid), dim3(NTHREADS), lds, s, p);
    return master_check_launch("window_attn_proj_bf16");
}
