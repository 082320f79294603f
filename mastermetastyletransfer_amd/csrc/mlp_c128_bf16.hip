// Fused MLP block for C = 128 (Swin-B stage 1: 128 -> 512 -> 128), bf16 throughput mode:  out = res + W2 . gelu(W1 . LN?(x) + b1) + b2
// (same contract as mlp_bf16.hip; torchvision MLP + norm2 + residual of the stage-1 Swin blocks, built at reference utils.py:89-93).
//
// Why a specialised form.  With C = 128 the generic kernel (128 tokens per workgroup, 32 per wave) spends 39 k clocks per
// workgroup on 8 k clocks of MFMA: prologue (X + LayerNorm) 5 k, epilogue 6 k, ~2.4 k of fixed cost per 32-KB weight step and the
// GELU bursts (profiles/probes/clock_probe.hip 128).  The small channel count leaves registers for TWO token tiles per wave:
//   workgroup = 256 tokens, 4 waves (one per SIMD), wave = 64 tokens = two 32-token MFMA tiles,
// so every weight fragment read from LDS feeds two MFMAs, the weight stream per token halves (2048 -> 1024 workgroups per
// 262 144-token launch) and every fixed cost is amortised over twice the tokens.  Register budget per lane: X fragments 64,
// hidden accumulators 128, output accumulators 128, GELU'd hidden fragments 16 (produced per hidden tile inside phase 2), weight fragments 32.
//
// Per hidden chunk of 128 (HID/128 chunks):  phase 1  H^T = W1c . X^T + b1c  (64 MFMAs per wave),  GELU + pack + half-wave exchange
// (packed-fp32 polynomial, common.h),  phase 2  O^T += W2c . H^T  (64 MFMAs).  Weights: 32-KB steps W1c = [128 hidden][128 k] and
// W2c = [128 out][128 hidden of the chunk], both with 256-byte rows, through a 4-slot LDS ring by LDS-DMA three steps ahead
// (source-swizzled: position p of row r holds chunk p ^ (r & 15)), counted vmcnt + one barrier per step, fragment reads pipelined
// with inline-asm ds_read_b128 - the machinery of mlp_bf16.hip.
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct M128Args {
    const bf16_t* x; const float* ln_g; const float* ln_b;
    const bf16_t* w1; const float* b1; const bf16_t* w2; const float* b2;
    const bf16_t* res; bf16_t* out;
    int M, HID, ldx, ldr, ldo; float eps;
};

union Frag { u32x4 u; bf16x8_t b; };

constexpr int C = 128, KS = C / 16, SLOT = 32768, RING = 4, AHEAD = 3, PIECES = 8;   // 8 one-KB DMA pieces per wave per step

#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define LDS_READ128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define LGKM_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

__global__ __launch_bounds__(256, 1) void mlp_c128_kernel(M128Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [RING][SLOT] weights | b1[HID] | b2[C] | ln_g[C] | ln_b[C]
    float* b1s = (float*)(lds + RING * SLOT);
    float* b2s = b1s + p.HID;
    float* lng = b2s + C;
    float* lnb = lng + C;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int m0w = blockIdx.x * 256 + wave * 64;                         // this wave's 64 tokens: tile tt covers m0w + 32 tt + r
    const int nhc = p.HID / 128, T = 2 * nhc;                             // stream: W1(0) W2(0) W1(1) W2(1) ...

    // ---- LDS-DMA of stream step t into ring slot t % RING: piece I = rows 4I..4I+3 of 256 B, source chunk (lane&15) ^ (row&15).
    //      A piece is four instructions (running scalar source pointer and LDS address, the lane's byte offset precomputed per i & 3 (one v_add per piece),
    //      no branch: steps past the end of the stream re-read the first KB of W1 into a 1-KB dummy region per wave), as in
    //      mlp_bf16.hip: a wave issues in order and each LDS-DMA instruction already costs it 60+ clocks next to MFMAs -------------------
    const int row0 = wave * (PIECES * 4) + (lane >> 4), l15 = lane & 15;
#define SWB(i) ((l15 ^ ((row0 + 4 * (i)) & 15)) << 4)      /* bytes */
    const unsigned vo1_0 = row0 * (C * 2) + SWB(0), vo1_1 = row0 * (C * 2) + SWB(1), vo1_2 = row0 * (C * 2) + SWB(2), vo1_3 = row0 * (C * 2) + SWB(3);
    const unsigned vo_w2 = row0 * 2 * (p.HID - C);           // a W2 step's offsets = the W1 ones + this (row pitch HID instead of C)
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds;
    const unsigned dummy_lds = lds0 + RING * SLOT + (p.HID + 3 * C) * 4 + wave * 1024;
    const char* d_ptr = nullptr; unsigned d_pstride = 0, d_lds = 0, d_lstride = 0, d_vadd = 0;
    auto set_dma_step = [&](int t) __attribute__((always_inline)) {
        const bool valid = t < T;
        const int hc = valid ? t >> 1 : 0;
        const int mk = -(int)(!valid || !(t & 1)), vm = -(int)valid;                      // mk: a W1 step (or the dummy)
        d_vadd = vo_w2 & ~(unsigned)mk;                          // per lane, once per step
        d_pstride = (unsigned)(((4 * C * 2) & mk) | ((4 * p.HID * 2) & ~mk)) & (unsigned)vm;
        d_lstride = 1024u & (unsigned)vm;
        d_lds = ((lds0 + (t % RING) * SLOT + wave * PIECES * 1024) & (unsigned)vm) | (dummy_lds & ~(unsigned)vm);
        const char* b1p = (const char*)(p.w1 + (size_t)hc * 128 * C);
        const char* b2p = (const char*)(p.w2 + hc * 128);
        d_ptr = (const char*)(((size_t)b1p & (size_t)(long long)mk) | ((size_t)b2p & ~(size_t)(long long)mk));
    };
    auto dma_piece = [&](int i) __attribute__((always_inline)) {                              // pieces of a step go out in order i = 0..7
        const unsigned a1 = (i & 3) == 0 ? vo1_0 : (i & 3) == 1 ? vo1_1 : (i & 3) == 2 ? vo1_2 : vo1_3;     // i is a constant here
        const unsigned voff = a1 + d_vadd;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(d_ptr + voff),
                                         (__attribute__((address_space(3))) void*)(size_t)d_lds, 16, 0, 0);
        d_ptr += (size_t)d_pstride; d_lds += d_lstride;
    };
    int t_issue = 0;
    for (; t_issue < AHEAD; ++t_issue) {
        set_dma_step(t_issue);
#pragma unroll
        for (int i = 0; i < PIECES; ++i) dma_piece(i);
    }

    for (int i = tid; i < p.HID; i += 256) b1s[i] = p.b1[i];
    if (tid < C) {
        b2s[tid] = p.b2[tid];
        if (p.ln_g) { lng[tid] = p.ln_g[tid]; lnb[tid] = p.ln_b[tid]; }
    }

    // ---- X fragments (B operand of phase 1): lane (token, half h) holds x[token][16s + 8h .. +8]; two token tiles ----------------------
    Frag xf[2][KS];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        int m = m0w + 32 * tt + r; m = m < p.M ? m : p.M - 1;
        const bf16_t* xp = p.x + (size_t)m * p.ldx + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[tt][s].u = *(const u32x4*)(xp + 16 * s);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // parameter vectors visible
    if (p.ln_g) {
#ifdef MLP_LN_SCALAR         // the r01-r04 form (un-packed arithmetic), kept for A/B runs
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) sum += __uint_as_float(xf[tt][s].u[e] << 16) + __uint_as_float(xf[tt][s].u[e] & 0xffff0000u);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * (1.0f / C);
            float sq = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d0 = __uint_as_float(xf[tt][s].u[e] << 16) - mean, d1 = __uint_as_float(xf[tt][s].u[e] & 0xffff0000u) - mean;
                    sq += d0 * d0 + d1 * d1;
                }
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = rsqrtf(sq * (1.0f / C) + p.eps);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const float* gp = lng + 16 * s + 8 * h; const float* bp = lnb + 16 * s + 8 * h;
                const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 4), b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
                float o[8];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    o[2 * e] = (__uint_as_float(xf[tt][s].u[e] << 16) - mean) * rstd * g0[2 * e] + b0[2 * e];
                    o[2 * e + 1] = (__uint_as_float(xf[tt][s].u[e] & 0xffff0000u) - mean) * rstd * g0[2 * e + 1] + b0[2 * e + 1];
                    o[4 + 2 * e] = (__uint_as_float(xf[tt][s].u[2 + e] << 16) - mean) * rstd * g1[2 * e] + b1[2 * e];
                    o[5 + 2 * e] = (__uint_as_float(xf[tt][s].u[2 + e] & 0xffff0000u) - mean) * rstd * g1[2 * e + 1] + b1[2 * e + 1];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) xf[tt][s].u[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]);
            }
        }
#else                        // r05: packed math (common.h)
        layernorm_frags_pk<KS>(xf[0], lng, lnb, h, p.eps);
        layernorm_frags_pk<KS>(xf[1], lng, lnb, h, p.eps);
#endif
    }

    // ---- per-step synchronisation: my DMAs of step t landed -> barrier -> everyone's have, and everyone left step t-1's slot ------
    int t_cons = 0;
    auto begin_step = [&]() __attribute__((always_inline)) {
        WAIT_VMCNT(16);                                                 // two later steps (real or dummy: 8 pieces each) may still be in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        set_dma_step(t_issue++);                                        // its pieces are issued between this step's MFMAs
        const unsigned char* slot = lds + (t_cons % RING) * SLOT;
        ++t_cons;
        return slot;
    };
    auto lds_off = [&](const unsigned char* ptr) { return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)ptr; };
    const int x256 = (h ^ (r & 15)) << 4;                                // swizzle term: chunk (2s+h) ^ (row & 15)

    f32x16 oacc[2][4];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[tt][t][e] = 0.f;

    for (int hc = 0; hc < nhc; ++hc) {
        // ---- phase 1: H^T[128 hidden][64 tokens] = W1c . X^T + b1c --------------------------------------------------------------
        f32x16 hacc[2][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *(const f32x4*)(b1s + hc * 128 + t * 32 + 8 * g + 4 * h);       // register e = hidden row (e&3) + 8(e>>2) + 4h
#pragma unroll
                for (int e = 0; e < 4; ++e) { hacc[0][t][4 * g + e] = bv[e]; hacc[1][t][4 * g + e] = bv[e]; }
            }
        {
            const unsigned base = lds_off(begin_step()) + r * 256;
            u32x4 a[2][4];
            auto rd = [&](int buf, int s) {
                const unsigned ad = base + ((32 * s) ^ x256);
                LDS_READ128(a[buf][0], ad, 0); LDS_READ128(a[buf][1], ad, 8192);
                LDS_READ128(a[buf][2], ad, 16384); LDS_READ128(a[buf][3], ad, 24576);
            };
            rd(0, 0);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) { rd((s + 1) & 1, s + 1); LGKM_WAIT(4); } else { LGKM_WAIT(0); }
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    Frag f; f.u = a[s & 1][jt];
                    hacc[0][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b, xf[0][s].b, hacc[0][jt], 0, 0, 0);
                    hacc[1][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b, xf[1][s].b, hacc[1][jt], 0, 0, 0);
                    if (jt == 1) { __builtin_amdgcn_sched_barrier(0); dma_piece(s); __builtin_amdgcn_sched_barrier(0); }
                }
            }
        }
        // ---- phase 2: O^T[128 out][64 tokens] += W2c . H^T ------------------------------------------------------------------------
        {
            const unsigned base = lds_off(begin_step()) + r * 256;
            u32x4 a[2][4];
            auto rd = [&](int buf, int g) {
                const unsigned ad = base + ((32 * g) ^ x256);
                LDS_READ128(a[buf][0], ad, 0); LDS_READ128(a[buf][1], ad, 8192);
                LDS_READ128(a[buf][2], ad, 16384); LDS_READ128(a[buf][3], ad, 24576);
            };
            rd(0, 0);
            Frag hf[2][2];                                                // B fragments of the hidden tile being consumed: [token tile][k-step]
#pragma unroll
            for (int g = 0; g < 8; ++g) {                                 // k-step g = hidden units 16g .. 16g+15 of the chunk
                if ((g & 1) == 0) {
                    // GELU + pack of hidden tile jt = g/2, produced right before the two k-steps that consume it (16 live registers
                    // instead of 64): accumulator tile -> bf16 pairs -> natural k order by the half-wave exchange
                    const int jt = g >> 1;
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            unsigned u[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int e0 = 8 * s2 + 2 * q;
                                const f32x2 v = gelu_poly2(f32x2{hacc[tt][jt][e0], hacc[tt][jt][e0 + 1]});
                                u[q] = pack_bf16x2(v[0], v[1]);
                            }
                            // half 0 keeps u0,u1 and takes the partner's u0,u1 as words 2,3; half 1 takes the partner's u2,u3 as words 0,1
                            auto sw0 = __builtin_amdgcn_permlane32_swap(u[0], u[2], false, false);
                            auto sw1 = __builtin_amdgcn_permlane32_swap(u[1], u[3], false, false);
                            hf[tt][s2].u[0] = sw0[0]; hf[tt][s2].u[2] = sw0[1];
                            hf[tt][s2].u[1] = sw1[0]; hf[tt][s2].u[3] = sw1[1];
                        }
                }
                if (g + 1 < 8) { rd((g + 1) & 1, g + 1); LGKM_WAIT(4); } else { LGKM_WAIT(0); }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    Frag f; f.u = a[g & 1][nt];
                    oacc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b, hf[0][g & 1].b, oacc[0][nt], 0, 0, 0);
                    oacc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b, hf[1][g & 1].b, oacc[1][nt], 0, 0, 0);
                    if (nt == 1) { __builtin_amdgcn_sched_barrier(0); dma_piece(g); __builtin_amdgcn_sched_barrier(0); }
                }
            }
        }
    }

    // ---- epilogue: each wave transposes its [64 tokens][128 ch] fp32 tile through the idle ring (own 32-KB region, 16-B chunks
    //      swizzled by token & 15) and walks it two rows per instruction: + b2, + residual, bf16, whole 256-B rows ----------------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                         // every wave is done reading the last weight slot
    {
        float* stage = (float*)(lds + wave * 32768);                     // [64][128] fp32
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = 32 * tt + r, c = nt * 8 + 2 * g + h;  // chunk of channels nt*32 + 8g + 4h .. +3
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = oacc[tt][nt][4 * g + e];
                    *(f32x4*)(stage + row * C + ((c ^ (row & 15)) << 2)) = v;
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // wave-private tile
        const int q = lane & 31, rsub = lane >> 5;                        // 32 chunks of 16 B per row, two rows per instruction
        const f32x4 bv = *(const f32x4*)(b2s + 4 * q);
#pragma unroll
        for (int half = 0; half < 2; ++half) {                            // residual loads of 16 rows' worth at a time (registers)
            u32x2 rr[16];
            if (p.res) {
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int mm = m0w + half * 32 + it * 2 + rsub;
                    rr[it] = mm < p.M ? *(const u32x2*)(p.res + (size_t)mm * p.ldr + 4 * q) : u32x2{0u, 0u};
                }
            }
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int row = half * 32 + it * 2 + rsub, mm = m0w + row;
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
    }
}

}  // namespace

// MASTER_ERR_UNSUPPORTED unless C = 128 and HID is a multiple of 128 (the caller then uses mlp_bf16)
int mlp_c128_bf16(const bf16_t* x, const float* ln_g, const float* ln_b, const bf16_t* w1, const float* b1, const bf16_t* w2,
                  const float* b2, const bf16_t* res, bf16_t* out, int M, int Cc, int HID, int ldx, int ldr, int ldo, float eps,
                  hipStream_t s) {
    if (Cc != C || HID % 128 != 0 || HID < 128) return MASTER_ERR_UNSUPPORTED;
    const size_t lds = (size_t)RING * SLOT + (size_t)(HID + 3 * C) * sizeof(float) + 4 * 1024;        // ring | b1, b2, ln | 1-KB DMA dummy per wave
    if (lds > 160 * 1024) return MASTER_ERR_UNSUPPORTED;
    M128Args p{x, ln_g, ln_b, w1, b1, w2, b2, res, out, M, HID, ldx, ldr, ldo, eps};
    static LdsLimit lds_limit;
    lds_limit.ensure((const void*)mlp_c128_kernel, (int)((int)lds));
    hipLaunchKernelGGL(mlp_c128_kernel, dim3((M + 255) / 256), dim3(256), lds, s, p);
    return master_check_launch("mlp_c128_bf16");
}
