// Fused MLP block, bf16 throughput mode, C = 256, EIGHT waves = two per SIMD (r05; DESIGN section 6, "what comes next" item 1):
//   out = res + W2 . gelu(W1 . LN?(x) + b1) + b2     (reference style_transformer.py:366,:392,:839-841,:865-882,:991,:1125)
//
// mlp_bf16.hip runs one wave per SIMD: everything between the MFMAs of its tile (weight-stream issue, fragment-read latency, GELU, the
// step barrier) is issue latency of an in-order wave that nothing overlaps, and a second wave does not fit its register budget.  Here both
// waves of a SIMD (w and w + 4) own the SAME 32 tokens and split the work of a hidden block of 64 units between them:
//   phase 1   wave half nh computes H^T rows [32 nh, +32) of the block   (A = W1 rows from LDS, B = the X fragments of the whole row, which
//             BOTH waves keep in registers: 64 registers)                               16 MFMAs per wave and block
//   GELU      of its own 32 x 32 tile, one hidden value behind every phase-1 MFMA of the NEXT block (second accumulator tile: 16 registers),
//             packed to bf16, half-wave exchange (v_permlane32_swap) into the natural k order - as in mlp_bf16.hip
//   exchange  the packed tile (2 fragments) goes to a 16-KB LDS region (lane-linear 16-B slots: conflict-free), the partner reads it behind
//             the next step barrier: every wave then has all four k-steps of H^T
//   phase 2   wave half nh computes O^T rows [128 nh, +128) (4 accumulator tiles: 64 registers), A = W2 rows from LDS   16 MFMAs per wave and block
// Same MFMA count, LDS fragment traffic and weight stream per workgroup as the one-wave form; ~245 registers per wave instead of ~480.
//
// Weight stream: 32-KB steps through a 4-slot LDS ring by LDS-DMA, three steps ahead, one barrier per step (16 MFMAs per wave = 32 per
// SIMD per step, as before): W1 step = [64 j][256 k] (512-B rows, position p of a row holds source chunk p ^ (row & 15): every 16-lane
// group of a fragment ds_read_b128 hits 16 distinct 16-B slots), W2 step = [256 n][64 j] (128-B rows, chunk p ^ ((row >> 1) & 7)).
// Stream order W1(0) | W1(1) W2(0) | W1(2) W2(1) | ... | W2(last).  Four 1-KB pieces per wave and step, dealt between the MFMAs.
//
// MEASURED OUTCOME (MI355X, profiles/r05_mlp_8w_mi355x.txt): correct on the first run (the fused-MLP tests, form "8w"), and SLOWER than the
// one-wave form: 53.1 us against 49.4 at M = 32 768 without a LayerNorm (62 against 53 with one: both waves of a band load and normalise
// the row), 56.4 before the fragment reads were pipelined by hand.  Probe builds: no GELU 48.0, no weight stream 46.7, neither 42.6, no
// step barrier 50.9 - the partner wave does NOT hide the weight stream's issue or the GELU (they cost what they cost the one-wave form),
// and what is left without them is still 3x the launch's 13.6 us of MFMA.  LDS bandwidth is not the reason (a 32x32x16 MFMA consumes one
// 1-KB A fragment: 128 B per clock for four SIMDs at full rate, half of the 256 B per clock that ds_read_b128 gets on gfx950; the weight
// stream adds a quarter): what the second wave cannot hide is per-instruction issue on the SIMD both waves share, and it pays twice for the
// row loads / LayerNorm and once more for the exchange.  Opt-in (MASTER_MLP_FORM=8w), not the default.
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

constexpr int C = 256;
constexpr int NW = 8;
constexpr int SLOT = 32768;      // bytes per ring slot = one step of weights
constexpr int RING = 4;
constexpr int AHEAD = 3;         // steps of DMA in flight ahead of the one being consumed
constexpr int PPW = 4;           // 1-KB LDS-DMA pieces per wave and step
constexpr int HB = 64;           // hidden units per block
constexpr int XCH = 16384;       // exchange region: [band 4][half 2][k-step 2][64 lanes x 16 B]
constexpr int KS1 = C / 16;      // k-steps of phase 1

#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define SB() __builtin_amdgcn_sched_barrier(0)

__global__ __launch_bounds__(NW * 64, 1) void mlp_8w_bf16_kernel(MlpArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [RING][SLOT] | exchange | b1 [HID] | b2 [C] | ln gamma, beta [C] | dummy 1 KB per wave
    unsigned char* xch = lds + RING * SLOT;
    float* b1s = (float*)(xch + XCH);
    float* b2s = b1s + p.HID;
    float* lng = b2s + C;
    float* lnb = lng + C;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int band = wave & 3, nh = wave >> 2;          // token band; which half of every hidden block / of the output channels
    const int r = lane & 31, h = lane >> 5;
    const int m = blockIdx.x * 128 + band * 32 + r;
    const int mc = m < p.M ? m : p.M - 1;
    const int nhb = p.HID / HB;
    const int T = 2 * nhb;                              // stream: W1(0) | W1(1) W2(0) | ... | W1(nhb-1) W2(nhb-2) | W2(nhb-1)

    // ---- LDS-DMA of stream step t into ring slot t % RING (the scheme of mlp_bf16.hip: a wave-uniform running source pointer and LDS
    //      address per step, one precomputed per-lane byte offset per piece; steps past the end re-read 1 KB into a dummy region) --------
    const int row1 = wave * 8 + (lane >> 5), p31 = lane & 31;          // W1 step: piece i = rows row1 + 2 i (2 rows x 512 B)
    const int row2 = wave * 32 + (lane >> 3), p7 = lane & 7;           // W2 step: piece i = rows row2 + 8 i (8 rows x 128 B)
#define SW512(i) ((p31 ^ ((row1 + 2 * (i)) & 15)) << 4)
#define SW128(i) ((p7 ^ (((row2 + 8 * (i)) >> 1) & 7)) << 4)
    const unsigned vo1_0 = row1 * (C * 2) + SW512(0), vo1_1 = row1 * (C * 2) + SW512(1), vo1_2 = row1 * (C * 2) + SW512(2),
                   vo1_3 = row1 * (C * 2) + SW512(3);
    const unsigned vo2_0 = row2 * (p.HID * 2) + SW128(0), vo2_1 = row2 * (p.HID * 2) + SW128(1);      // (row >> 1) & 7 of piece i depends on i & 1 only
    const int pstride_w1 = 2 * C * 2, pstride_w2 = 8 * p.HID * 2;
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds;
    const unsigned dummy_lds = lds0 + RING * SLOT + XCH + (p.HID + 3 * C) * 4 + wave * 1024;
    const char* d_ptr = nullptr; unsigned d_pstride = 0, d_lds = 0, d_lstride = 0, d_w1m = 0;
    auto set_dma_step = [&](int t) __attribute__((always_inline)) {
        const bool valid = t < T;
        bool is_w1; int hb;
        if (t == 0) { is_w1 = true; hb = 0; }
        else {
            const int u = t - 1, b = u >> 1, o = u & 1;
            if (b < nhb - 1 && o == 0) { is_w1 = true; hb = b + 1; }
            else { is_w1 = false; hb = b; }
        }
        if (!valid) { is_w1 = true; hb = 0; }
        const int mk = -(int)is_w1, vm = -(int)valid;      // mask arithmetic, not a ternary on two variables (hipcc turns that into a scratch table)
        d_w1m = (unsigned)mk;
        d_pstride = (unsigned)(((pstride_w1 & mk) | (pstride_w2 & ~mk)) & vm);
        d_lstride = 1024u & (unsigned)vm;
        d_lds = ((lds0 + (t % RING) * SLOT + wave * PPW * 1024) & (unsigned)vm) | (dummy_lds & ~(unsigned)vm);
        const char* b1p = (const char*)(p.w1 + (size_t)hb * HB * C);
        const char* b2p = (const char*)(p.w2 + hb * HB);
        d_ptr = (const char*)(((size_t)b1p & (size_t)(long long)mk) | ((size_t)b2p & ~(size_t)(long long)mk));
    };
    auto dma_piece = [&](int i) __attribute__((always_inline)) {                                      // pieces of a step go out in order i = 0..3 (i is a constant)
        const unsigned a1 = i == 0 ? vo1_0 : i == 1 ? vo1_1 : i == 2 ? vo1_2 : vo1_3;
        const unsigned a2 = (i & 1) == 0 ? vo2_0 : vo2_1;
        const unsigned voff = (a1 & d_w1m) | (a2 & ~d_w1m);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(d_ptr + voff),
                                         (__attribute__((address_space(3))) void*)(size_t)d_lds, 16, 0, 0);
        d_ptr += (size_t)d_pstride; d_lds += d_lstride;
    };
    auto issue = [&](int t) __attribute__((always_inline)) {
        set_dma_step(t);
#pragma unroll
        for (int i = 0; i < PPW; ++i) dma_piece(i);
    };

    // ---- X fragments (B operand of phase 1): lane (token m, half h) holds x[m][16 s + 8 h .. +8]; both waves of a band hold the row ----
    Frag xf[KS1];
    {
        const bf16_t* xp = p.x + (size_t)mc * p.ldx + 8 * h;
#pragma unroll
        for (int s = 0; s < KS1; ++s) xf[s].u = *(const u32x4*)(xp + 16 * s);
    }
    int t_issue = 0;
    for (; t_issue < AHEAD; ++t_issue) issue(t_issue);

    for (int i = tid; i < p.HID; i += NW * 64) b1s[i] = p.b1[i];
    if (tid < C) {
        b2s[tid] = p.b2[tid];
        if (p.ln_g) { lng[tid] = p.ln_g[tid]; lnb[tid] = p.ln_b[tid]; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                         // parameter vectors visible to every wave
    if (p.ln_g) layernorm_frags_pk<KS1>(xf, lng, lnb, h, p.eps);      // packed math (common.h); both waves of a band normalise the row

    // ---- per-step synchronisation: my DMAs of step t have landed -> barrier -> everyone's have, everyone is done reading step t - 1 (whose
    //      slot step t + AHEAD goes into), and every exchange write issued before it is visible ---------------------------------------
    int t_cons = 0;
    auto begin_step = [&]() __attribute__((always_inline)) {
#ifdef M8W_NODMA
        WAIT_VMCNT(0);
#else
        WAIT_VMCNT(8);                                      // two later steps (real or dummy: 4 pieces each) may still be in flight
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef M8W_NOBARRIER        // probe builds: no step barrier (timing only)
        __builtin_amdgcn_s_barrier();
#endif
        set_dma_step(t_issue);                              // its pieces go out between this step's MFMAs
        ++t_issue;
        const unsigned char* slot = lds + (t_cons % RING) * SLOT;
        ++t_cons;
        return slot;
    };

    f32x16 hacc, hprev, oacc[4];
    Frag hown[2];                                           // this wave's packed GELU'd tile: k-steps 2 nh, 2 nh + 1 of the block
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[t][e] = 0.f;

    // GELU + pack of the previous block's tile, one hidden value behind every phase-1 MFMA of the next block (mlp_bf16.hip's fine-grained
    // form): value v = 0..15 -> k-step s = v >> 3, pair q = (v >> 1) & 3, element z = v & 1; after the eighth value of a k-step the half-wave
    // exchange that turns the accumulator's k order (8 (e >> 2) + 4 half + (e & 3)) into the natural one (8 half + e), then the fragment
    // goes to the exchange region for the partner wave.
    unsigned gu[4];
#ifdef M8W_GELU_FINE1
    float gfine[2];
#endif
    unsigned char* xw = xch + ((band * 2 + nh) * 2) * 1024 + lane * 16;          // [k-step s] stride 1 KB
    const unsigned char* xr = xch + ((band * 2 + (nh ^ 1)) * 2) * 1024 + lane * 16;
    auto gelu_fine = [&](int v) __attribute__((always_inline)) {
        const int s = v >> 3, q = (v >> 1) & 3, z = v & 1;
        const int e0 = 4 * (2 * s + (q >> 1)) + 2 * (q & 1);
#ifdef M8W_GELU_FINE1        // one value per call in un-packed fp32 (the one-wave kernel's form)
        gfine[z] = gelu_poly1(hprev[e0 + z]);
        if (z == 1) gu[q] = pack_bf16x2(gfine[0], gfine[1]);
        asm volatile("" :: "v"(gfine[z]));                  // pin: LLVM otherwise sinks the arithmetic below the MFMAs
#else                        // a pair per odd call in packed fp32: the vector unit is shared by the SIMD's two waves, so instruction count matters
        if (z == 1) {
#ifdef M8W_NOGELU            // probe builds: identity instead of the polynomial
            const f32x2 g = f32x2{hprev[e0], hprev[e0 + 1]};
#else
            const f32x2 g = gelu_poly2(f32x2{hprev[e0], hprev[e0 + 1]});
#endif
            gu[q] = pack_bf16x2(g.x, g.y);
            asm volatile("" :: "v"(gu[q]));                 // pin: LLVM otherwise sinks the arithmetic below the MFMAs
        }
#endif
        if ((v & 7) == 7) {
            auto sw0 = __builtin_amdgcn_permlane32_swap(gu[0], gu[2], false, false);
            auto sw1 = __builtin_amdgcn_permlane32_swap(gu[1], gu[3], false, false);
            hown[s].u[0] = sw0[0]; hown[s].u[2] = sw0[1];
            hown[s].u[1] = sw1[0]; hown[s].u[3] = sw1[1];
            *(u32x4*)(xw + s * 1024) = hown[s].u;
        }
    };

    // fragment addressing: W1 step rows j = 32 nh + r (512-B rows), W2 step rows n = 128 nh + 32 nt + r (128-B rows)
    const int j1 = 32 * nh + r;
    const unsigned off1 = j1 * 512, sw1x = (h ^ (j1 & 15)) << 4;                  // chunk (2 s + h) ^ (j & 15)
    const unsigned off2 = (128 * nh + r) * 128, sw2x = (h ^ ((r >> 1) & 7)) << 4; // chunk (2 g + h) ^ ((n >> 1) & 7); (n >> 1) & 7 = (r >> 1) & 7

    // Fragment reads are software-pipelined by hand, as in mlp_bf16.hip (first version: compiler-scheduled reads between the sched_barriers
    // around the dealt GELU / DMA pieces put the whole LDS latency in front of every MFMA - 56 us against the one-wave form's 49): reads are
    // inline asm, requested three k-steps ahead in phase 1 (one fragment per MFMA) and one k-step ahead in phase 2 (four per group of four
    // MFMAs); a counted lgkmcnt retires exactly the fragments of the k-step about to be used (LDS returns in order; the exchange write and
    // other LDS operations of the wave only make the count conservative), a sched_barrier keeps the MFMAs below it.
#define LDS_READ128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define LGKM_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    auto lds_off = [&](const unsigned char* ptr) { return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)ptr; };

    auto phase1_step = [&](const unsigned char* slot, bool pipe) __attribute__((always_inline)) {
        const unsigned base = lds_off(slot) + off1;
        u32x4 a[4];
        auto rd = [&](int buf, int s) { const unsigned ad = base + ((32 * s) ^ sw1x); LDS_READ128(a[buf], ad, 0); };
        rd(0, 0); rd(1, 1); rd(2, 2);
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            if (s + 3 < KS1) { rd((s + 3) & 3, s + 3); LGKM_WAIT(3); }
            else if (s + 2 < KS1) { LGKM_WAIT(2); }
            else if (s + 1 < KS1) { LGKM_WAIT(1); }
            else { LGKM_WAIT(0); }
            Frag f; f.u = a[s & 3];
            hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b, xf[s].b, hacc, 0, 0, 0);
#ifndef M8W_NODMA            // probe builds: no weight stream behind the first three steps (timing only, stale operands)
            if ((s & 3) == 1) { SB(); dma_piece(s >> 2); SB(); }
#endif
            if (pipe) { SB(); gelu_fine(s); SB(); }
        }
    };
    auto phase2_step = [&](const unsigned char* slot) __attribute__((always_inline)) {
        const unsigned base = lds_off(slot) + off2;
        const unsigned xra = lds_off(xr);
        Frag hp[2];                                         // the partner's two k-steps
        u32x4 a[2][4];
        auto rd = [&](int buf, int g) {
            const unsigned ad = base + ((32 * g) ^ sw2x);
            LDS_READ128(a[buf][0], ad, 0); LDS_READ128(a[buf][1], ad, 4096); LDS_READ128(a[buf][2], ad, 8192); LDS_READ128(a[buf][3], ad, 12288);
        };
        LDS_READ128(hp[0].u, xra, 0); LDS_READ128(hp[1].u, xra, 1024);
        rd(0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g + 1 < 4) { rd((g + 1) & 1, g + 1); LGKM_WAIT(4); } else { LGKM_WAIT(0); }
            // k-step g covers hidden units 16 g .. +16 of the block: tile g >> 1 (half 0's for g < 2), k-step g & 1 inside it
            Frag hb; hb.u = (g >> 1) == nh ? hown[g & 1].u : hp[g & 1].u;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                Frag f; f.u = a[g & 1][nt];
                oacc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b, hb.b, oacc[nt], 0, 0, 0);
#ifndef M8W_NODMA
                if (nt == 1) { SB(); dma_piece(g); SB(); }
#endif
            }
        }
    };
    auto init_hacc = [&](int hb) __attribute__((always_inline)) {      // accumulators start from the bias b1 of their rows
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bv = *(const f32x4*)(b1s + hb * HB + nh * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) hacc[4 * g + e] = bv[e];
        }
    };

    // prologue: phase 1 of block 0
    init_hacc(0);
    phase1_step(begin_step(), false);
    for (int hb = 0; hb < nhb; ++hb) {
        hprev = hacc;
        if (hb + 1 < nhb) {
            init_hacc(hb + 1);
            phase1_step(begin_step(), true);                // + GELU(hb) in the gaps, tile -> exchange region
        } else {
#pragma unroll
            for (int v = 0; v < 16; ++v) gelu_fine(v);
        }
        phase2_step(begin_step());                          // the barrier inside makes every wave's exchange write visible
    }

    // ---- epilogue: each wave transposes its [32 tokens][128 channels] fp32 tile through the idle ring (own 16-KB region, 16-B chunks
    //      swizzled by token & 15) and walks it row by row: residual loads and bf16 stores are contiguous 256-B row halves ---------------
    const bool res_in_regs = p.res == p.x && p.ldr == p.ldx && p.ln_g == nullptr;
    if (res_in_regs) {
        // residual = x without a LayerNorm: the raw X fragments are still in registers; fragment s holds channels 16 s + 8 h .. +7, the
        // accumulators 8 g + 4 h .. +3: the GELU path's half-wave exchange lines them up.  This wave's channels are k-steps 8 nh .. 8 nh + 7.
        auto add_x = [&](int s, int t) __attribute__((always_inline)) {       // s: compile-time k-step of x; t: accumulator tile
            auto swa = __builtin_amdgcn_permlane32_swap(xf[s].u[0], xf[s].u[2], false, false);
            auto swb = __builtin_amdgcn_permlane32_swap(xf[s].u[1], xf[s].u[3], false, false);
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int e0 = 4 * (2 * (s & 1) + gg);
                oacc[t][e0 + 0] += __uint_as_float(swa[gg] << 16); oacc[t][e0 + 1] += __uint_as_float(swa[gg] & 0xffff0000u);
                oacc[t][e0 + 2] += __uint_as_float(swb[gg] << 16); oacc[t][e0 + 3] += __uint_as_float(swb[gg] & 0xffff0000u);
            }
        };
        if (nh == 0) {
#pragma unroll
            for (int s = 0; s < 8; ++s) add_x(s, s >> 1);
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) add_x(8 + s, s >> 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // the dummy look-ahead pieces too: the ring is reused below
    __builtin_amdgcn_s_barrier();                                   // every wave is done reading the last weight slot
    {
        float* stage = (float*)(lds + wave * 16384);                // [32][128] fp32, row pitch 512 B
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = nt * 8 + 2 * g + h;                   // 16-B chunk of channels nt*32 + 8g + 4h .. +3
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = oacc[nt][4 * g + e];
                *(f32x4*)(stage + r * 128 + ((c ^ (r & 15)) << 2)) = v;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // wave-private tile: LDS operations of a wave complete in order
        const int q = lane & 31, rsub = lane >> 5;                  // two rows per wave instruction
        const int m0w = blockIdx.x * 128 + band * 32;
        const int ch0 = nh * 128 + 4 * q;
        const f32x4 bv = *(const f32x4*)(b2s + ch0);
        u32x2 rr[16];
        const bool res_mem = p.res != nullptr && !res_in_regs;
        if (res_mem) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int mm = m0w + it * 2 + rsub;
                rr[it] = mm < p.M ? *(const u32x2*)(p.res + (size_t)mm * p.ldr + ch0) : u32x2{0u, 0u};
            }
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 2 + rsub, mm = m0w + row;
            f32x4 v = *(const f32x4*)(stage + row * 128 + ((q ^ (row & 15)) << 2));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bv[e];
            if (res_mem) {
                v[0] += __uint_as_float(rr[it][0] << 16); v[1] += __uint_as_float(rr[it][0] & 0xffff0000u);
                v[2] += __uint_as_float(rr[it][1] << 16); v[3] += __uint_as_float(rr[it][1] & 0xffff0000u);
            }
            u32x2 o; o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]);
            if (mm < p.M) *(u32x2*)(p.out + (size_t)mm * p.ldo + ch0) = o;
        }
    }
}

}  // namespace

int mlp_8w_bf16(const bf16_t* x, const float* ln_g, const float* ln_b, const bf16_t* w1, const float* b1, const bf16_t* w2,
                const float* b2, const bf16_t* res, bf16_t* out, int M, int Cin, int HID, int ldx, int ldr, int ldo, float eps,
                hipStream_t s) {
    if (Cin != C || HID % HB != 0 || HID < 2 * HB) return MASTER_ERR_UNSUPPORTED;      // the caller falls back to the one-wave form
    MlpArgs p{x, ln_g, ln_b, w1, b1, w2, b2, res, out, M, HID, ldx, ldr, ldo, eps};
    const size_t lds = (size_t)RING * SLOT + XCH + (size_t)(HID + 3 * C) * sizeof(float) + NW * 1024;
    if (lds > 160 * 1024) return MASTER_ERR_UNSUPPORTED;
    static LdsLimit lds_limit;
    lds_limit.ensure((const void*)mlp_8w_bf16_kernel, (int)lds);
    hipLaunchKernelGGL(mlp_8w_bf16_kernel, dim3((M + 127) / 128), dim3(NW * 64), lds, s, p);
    return master_check_launch("mlp_8w_bf16");
}
