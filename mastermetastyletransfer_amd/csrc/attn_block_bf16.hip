// Fused window-attention block, bf16 throughput mode, C = 128 channels / 4 heads (Swin-B stage 1):
//
//     out = x + proj( window_attention( qkv( LayerNorm?(x) ) ) )          one launch, one pass over x
//
// (torchvision SwinTransformerBlock first half = reference style_transformer.py:390 with q = k = v:
//  norm1, pad/roll/partition :83-111, fused qkv :114-116, attention :120-155, proj :156, reverse :160-168, residual.)
//
// Persistent workgroups (one per CU) walk the windows; wave w IS head w.  Everything a wave needs besides the
// window's tokens is lane-private and loaded once per workgroup: its head's rows of Wq/Wk/Wv and its 32 rows of
// Wproj as MFMA fragments (128 VGPRs), the matching bias values, and its head's relative-position tile in LDS.
// Per window only the 49(+15) token rows move:
//   1. the NEXT window's rows are gathered by LDS-DMA (per-lane source address = the pad/roll/partition table)
//      while the current window computes;
//   2. LayerNorm in LDS (4 lanes per row); zero-padded slots stay exactly zero, so their q/k/v come out as the
//      projection bias, as in the reference (it pads after norm1 and before the Linear);
//   3. Q^T, K^T (channel x slot) and V (slot x channel) accumulate from the same X fragments;
//   4. S^T = K Q^T, softmax, O^T = V^T P^T: every operand is an accumulator tile of the previous product, packed
//      to bf16 in registers (k order 16s + 8(e>>2) + 4*half + (e&3) on both operands), no LDS traffic;
//   5. the heads' O tiles meet in LDS, each wave projects its 32 output channels and adds the residual.
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int C = 128, HEADS = 4, KS = C / 16;
constexpr int ROWB = C * 2;                     // bytes per token row (256)
constexpr int XRAW = 0, XS = 2 * 64 * ROWB, BIAS = 3 * 64 * ROWB;      // LDS offsets: 2 raw tiles (ping-pong), 1 working tile
constexpr int BSTR = 68;                                               // floats per rel-pos bias row in LDS (272 B: conflict-free column reads)
constexpr int TABLES = BIAS + HEADS * 64 * BSTR * 4;                   // then rel-pos bias (68 KB), then src[nW][64] i32, region[nW][64] u8
constexpr int LDS_MAX = 160 * 1024;
constexpr float LOG2E = 1.4426950408889634f;

struct BlockArgs {
    const bf16_t* x; const float* ln_g; const float* ln_b;
    const bf16_t* wqkv; const float* bqkv; const bf16_t* wproj; const float* bproj;
    const float* bias64; const int32_t* src; const uint8_t* region;
    bf16_t* out;
    int B, T, nW, N; float eps, scale;
    int wchunk, R;               // windows-per-image slices: workgroup g serves slice g % R = windows [slice*wchunk, +wchunk) of every image
};

union Frag { u32x4 u; bf16x8_t b; };

// hand-pipelined LDS fragment reads (see mlp_bf16.hip): the compiler neither waits for nor moves these
#define LDS_READ128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define LGKM_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ __forceinline__ unsigned lds_addr(const unsigned char* ptr) { return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)ptr; }

// registers 8s..8s+7 of an accumulator tile, scaled, packed to one bf16 fragment
__device__ __forceinline__ Frag pack_tile(const f32x16& t, int s, float mul) {
    Frag f;
#pragma unroll
    for (int i = 0; i < 4; ++i) f.u[i] = pack_bf16x2(t[8 * s + 2 * i] * mul, t[8 * s + 2 * i + 1] * mul);
    return f;
}

#ifdef ATTN_CLOCK_PROBE   // diagnostic build only (profiles/probes/attn_block_probe.hip): s_memtime sections of wave 0
#define PROBE_MARK() do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); probe_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define PROBE_ADD(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); probe_sec[i] += t_ - probe_last; probe_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PROBE_MARK() do {} while (0)
#define PROBE_ADD(i) do {} while (0)
#endif

__global__ __launch_bounds__(256, 1) void attn_block_c128_kernel(BlockArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef ATTN_CLOCK_PROBE
    const unsigned long long probe_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long probe_sec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, probe_last = 0;
    // 0 wait for the gathered tile + barrier | 1 LayerNorm + barrier + next gather issue | 2 QKV MFMAs | 3 S^T + bias + mask |
    // 4 softmax (exp2, pack, denominators) | 5 P.V | 6 heads meet in LDS (2 barriers) | 7 proj + residual + store
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // = head
    const int r = lane & 31, h = lane >> 5;
    // Geometries with many windows (512x512: 361) do not fit their tables in LDS next to the tiles and the rel-pos bias, so
    // the windows of an image are cut into R slices; a workgroup loads the tables of ONE slice and walks (image, window in slice).
    const int slice = blockIdx.x % p.R, w0 = slice * p.wchunk;
    const int cnt = p.nW - w0 < p.wchunk ? p.nW - w0 : p.wchunk;            // windows in this slice
    const int gi = blockIdx.x / p.R, ng = (gridDim.x - slice + p.R - 1) / p.R;   // this workgroup's index among those serving the slice
    const int nwin = p.B * cnt;

    // ---- once per workgroup: lane-private weight fragments and bias values -------------------------------
    Frag wq[KS], wk[KS], wv[KS], wp[KS];
    {
        const bf16_t* base = p.wqkv + (size_t)(wave * 32 + r) * C + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            wq[s].u = *(const u32x4*)(base + 16 * s);
            wk[s].u = *(const u32x4*)(base + (size_t)C * C + 16 * s);
            wv[s].u = *(const u32x4*)(base + (size_t)2 * C * C + 16 * s);
            wp[s].u = *(const u32x4*)(p.wproj + (size_t)(wave * 32 + r) * C + 8 * h + 16 * s);
        }
    }
    // the softmax scale (head_dim^-0.5, x log2 e for exp2) is folded into Wq and bq once per workgroup instead of 32 multiplies per
    // head and window when Q is packed
    const float qmul = p.scale * LOG2E;
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            wq[s].u[i] = pack_bf16x2(__uint_as_float(wq[s].u[i] << 16) * qmul, __uint_as_float(wq[s].u[i] & 0xffff0000u) * qmul);
    float bq[16], bk[16], bp[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int ch = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;        // channel of accumulator register e
        bq[e] = p.bqkv[ch] * qmul; bk[e] = p.bqkv[C + ch]; bp[e] = p.bproj[ch];
    }
    const float bvl = p.bqkv[2 * C + wave * 32 + r];                      // V tiles have the channel on the lane
    // rel-pos bias in the exp2 domain (x log2 e) once, instead of 64 multiplies per head and window
    for (int i = tid; i < HEADS * 64 * 64; i += 256) ((float*)(lds + BIAS))[(i >> 6) * BSTR + (i & 63)] = p.bias64[i] * LOG2E;
    // the pad/roll/partition tables of every window of the geometry: the loop below then has no ordinary global load
    int32_t* srcs = (int32_t*)(lds + TABLES);
    uint8_t* regs = (uint8_t*)(srcs + cnt * 64);
    for (int i = tid; i < cnt * 64; i += 256) { srcs[i] = p.src[w0 * 64 + i]; regs[i] = p.region ? p.region[w0 * 64 + i] : 0; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // LayerNorm parameters of this thread's 32 channels (thread = (row t>>2, quarter t&3) in the LN phase)
    const int lrow = tid >> 2, lq = tid & 3;
    float lg[32], lb[32];
    if (p.ln_g) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { lg[i] = p.ln_g[lq * 32 + i]; lb[i] = p.ln_b[lq * 32 + i]; }
    }

    // ---- gather of a window's token rows by LDS-DMA: piece I of 1 KB = slots 4I..4I+3 ----------------------------
    auto gather = [&](int win, int buf) {
        const int b = win / cnt, w = win - b * cnt;                        // w: window within the slice (table row)
        const int32_t* src = srcs + w * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int I = wave * 4 + i;
            const int slot = 4 * I + (lane >> 4);
            int idx = src[slot];
            idx = idx < 0 ? 0 : idx;                                       // padded slots are zeroed in the LN phase
            // r03: source-swizzled like the working tile (LDS position c of a row holds source chunk c ^ (slot & 15)): the residual reads
            // of the epilogue (32 lanes = 32 rows, one 8-byte piece each) were a 32-way bank conflict on the linear tile (rows are
            // 256 B = all 64 banks apart), 12.8 % of the kernel's wave cycles in SQ_LDS_BANK_CONFLICT; now 2-way
            const bf16_t* sp = p.x + ((size_t)b * p.T + idx) * C + ((lane & 15) ^ (slot & 15)) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                             (__attribute__((address_space(3))) void*)(lds + XRAW + buf * 64 * ROWB + I * 1024), 16, 0, 0);
        }
    };

    Frag ones; ones.u = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};      // bf16 1.0 x 8
    int win = gi, buf = 0;
    if (win < nwin) gather(win, 0);

    for (; win < nwin; win += ng, buf ^= 1) {
        const int b = win / cnt, w = win - b * cnt;
        const int32_t* src = srcs + w * 64;
        const size_t tok0 = (size_t)b * p.T;
        const unsigned char* raw = lds + XRAW + buf * 64 * ROWB;           // this window's un-normalised rows (also the residual)

        // ---- 1. LayerNorm: raw rows (LDS, linear) -> normalised bf16 rows (LDS, 16-B chunks swizzled by row&15) ------
        PROBE_MARK();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // raw tile landed; previous window's proj reads of XS done
        PROBE_ADD(0);
        {
            const int sidx = lrow < p.N ? src[lrow] : -1;
            // r05: two channels per instruction (v_pk_add / v_pk_mul / v_pk_fma: the phase runs no MFMA, so packed math costs nothing here)
            // and two instead of three operations per value in the normalisation: out = v * (rstd * gamma) + (beta - mean * rstd * gamma).
            // The arithmetic of this phase was 10 % of the kernel (probe build without it: 141.0 -> 126.5 us per launch).
            f32x2 v[16];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const u32x4 u = *(const u32x4*)(raw + lrow * ROWB + (((lq * 4 + c4) ^ (lrow & 15)) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[c4 * 4 + e] = f32x2{__uint_as_float(u[e] << 16), __uint_as_float(u[e] & 0xffff0000u)};
            }
            if (p.ln_g) {
                f32x2 s2 = v[0];
#pragma unroll
                for (int i = 1; i < 16; ++i) s2 += v[i];
                float sum = s2[0] + s2[1];
                sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64);
                const float mean = sum * (1.0f / C);
                const f32x2 m2 = {mean, mean};
                f32x2 q2 = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 16; ++i) { const f32x2 d = v[i] - m2; q2 = d * d + q2; }
                float sq = q2[0] + q2[1];
                sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64);
                const float rstd = rsqrtf(sq * (1.0f / C) + p.eps);
                const f32x2 r2 = {rstd, rstd}, nm2 = {-mean, -mean};
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x2 a = r2 * f32x2{lg[2 * i], lg[2 * i + 1]};
                    const f32x2 c = nm2 * a + f32x2{lb[2 * i], lb[2 * i + 1]};
                    v[i] = v[i] * a + c;
                }
            }
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                u32x4 o = {0u, 0u, 0u, 0u};
                if (sidx >= 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(v[c4 * 4 + e][0], v[c4 * 4 + e][1]);
                }
                *(u32x4*)(lds + XS + lrow * ROWB + (((lq * 4 + c4) ^ (lrow & 15)) << 4)) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // XS complete
        if (win + ng < nwin) gather(win + ng, buf ^ 1);      // next window's rows fly under this window's math
        PROBE_ADD(1);

        // ---- 2. Q^T, K^T [channel][slot] and V [slot][channel] of this wave's head --------------------------------------
        f32x16 qt[2], kt[2], vt[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) { qt[t][e] = bq[e]; kt[t][e] = bk[e]; vt[t][e] = bvl; }
        {
            const unsigned xb = lds_addr(lds + XS + r * ROWB);
            const int xsw = (h ^ (r & 15)) << 4;
            u32x4 xa[2][2];
            LDS_READ128(xa[0][0], xb + (0 ^ xsw), 0); LDS_READ128(xa[0][1], xb + (0 ^ xsw), 8192);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) {
                    const unsigned ad = xb + ((32 * (s + 1)) ^ xsw);
                    LDS_READ128(xa[(s + 1) & 1][0], ad, 0); LDS_READ128(xa[(s + 1) & 1][1], ad, 8192);
                    LGKM_WAIT(2);
                } else { LGKM_WAIT(0); }
                Frag x0, x1; x0.u = xa[s & 1][0]; x1.u = xa[s & 1][1];
                qt[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[s].b, x0.b, qt[0], 0, 0, 0);
                qt[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[s].b, x1.b, qt[1], 0, 0, 0);
                kt[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk[s].b, x0.b, kt[0], 0, 0, 0);
                kt[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk[s].b, x1.b, kt[1], 0, 0, 0);
                vt[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0.b, wv[s].b, vt[0], 0, 0, 0);
                vt[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1.b, wv[s].b, vt[1], 0, 0, 0);
            }
        }

        PROBE_ADD(2);
        // ---- 3. S^T[key][query] = K Q^T from packed accumulator tiles; bias, mask, softmax -------------------------------------
        f32x16 st[2][2];
        {
            Frag qf[2][2], kf[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s) { qf[t][s] = pack_tile(qt[t], s, 1.0f); kf[t][s] = pack_tile(kt[t], s, 1.0f); }
            const float* bias = (const float*)(lds + BIAS) + wave * 64 * BSTR;
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const float* brow = bias + (q2 * 32 + r) * BSTR;
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 bv = *(const f32x4*)(brow + k2 * 32 + 8 * g + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) st[k2][q2][4 * g + e] = bv[e];
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        st[k2][q2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[k2][s].b, qf[q2][s].b, st[k2][q2], 0, 0, 0);
                }
            }
        }
        // the shift mask only bites in windows that straddle a region boundary: when all 64 slots carry the same region id
        // (interior windows) nothing is masked and the per-element test is skipped (wave-uniform branch)
        bool mixed = false;
        if (p.region) {
            const uint8_t* reg = regs + w * 64;
            const unsigned r0 = reg[0];
            mixed = __builtin_amdgcn_ballot_w64(reg[lane] != r0) != 0ull;
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
                            if (((rk4 >> (8 * e)) & 0xffu) != rq) st[k2][q2][4 * g + e] += -100.0f * LOG2E;
                    }
            }
        }
        PROBE_ADD(3);
        float inv[2];
        Frag pf[2][2][2];                                                  // [query tile][key tile][k-step]
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
            float mx = st[0][q2][0];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, st[k2][q2][e]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        pf[q2][k2][s].u[j] = pack_bf16x2(__builtin_amdgcn_exp2f(st[k2][q2][8 * s + 2 * j] - mx),
                                                         __builtin_amdgcn_exp2f(st[k2][q2][8 * s + 2 * j + 1] - mx));
            // denominator on the matrix core (4 % busy in this kernel) instead of ~128 VALU ops: with an all-ones A fragment every
            // row of the product holds sum_k P[k][query] of the ROUNDED probabilities (the values P.V uses), so register 0 is it
            f32x16 rs;
#pragma unroll
            for (int e = 0; e < 16; ++e) rs[e] = 0.f;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int s = 0; s < 2; ++s) rs = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones.b, pf[q2][k2][s].b, rs, 0, 0, 0);
            inv[q2] = __builtin_amdgcn_rcpf(rs[0]);
        }

        PROBE_ADD(4);
        // ---- 4. O^T[d][query] = V^T P^T (V tiles as the A operand: X^T B with X = V[key][d]) ------------------------------------
        f32x16 ot[2];
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
            for (int e = 0; e < 16; ++e) ot[q2][e] = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const Frag vf = pack_tile(vt[k2], s, 1.0f);
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2)
                    ot[q2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.b, pf[q2][k2][s].b, ot[q2], 0, 0, 0);
            }

        PROBE_ADD(5);
        __builtin_amdgcn_s_barrier();                                      // every head is done reading XS (its X fragments)
        // O[slot][head*32 + d] -> LDS (the XS tile is reused), 8-B pieces: d = 8g + 4h + e
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
            const int slot = q2 * 32 + r;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 o;
                o[0] = pack_bf16x2(ot[q2][4 * g] * inv[q2], ot[q2][4 * g + 1] * inv[q2]);
                o[1] = pack_bf16x2(ot[q2][4 * g + 2] * inv[q2], ot[q2][4 * g + 3] * inv[q2]);
                *(u32x2*)(lds + XS + slot * ROWB + (((wave * 4 + g) ^ (slot & 15)) << 4) + 8 * h) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // all heads' O in LDS
        PROBE_ADD(6);

        // ---- 5. proj: out^T[n][slot] for this wave's 32 output channels, + bias + residual -------------------------------------------
        f32x16 pt[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) pt[t][e] = bp[e];
        {
            const unsigned ob = lds_addr(lds + XS + r * ROWB);
            const int xsw = (h ^ (r & 15)) << 4;
            u32x4 oa[2][2];
            LDS_READ128(oa[0][0], ob + (0 ^ xsw), 0); LDS_READ128(oa[0][1], ob + (0 ^ xsw), 8192);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) {
                    const unsigned ad = ob + ((32 * (s + 1)) ^ xsw);
                    LDS_READ128(oa[(s + 1) & 1][0], ad, 0); LDS_READ128(oa[(s + 1) & 1][1], ad, 8192);
                    LGKM_WAIT(2);
                } else { LGKM_WAIT(0); }
                Frag o0, o1; o0.u = oa[s & 1][0]; o1.u = oa[s & 1][1];
                pt[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[s].b, o0.b, pt[0], 0, 0, 0);
                pt[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[s].b, o1.b, pt[1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int slot = t * 32 + r;
            const int idx = slot < p.N ? src[slot] : -1;
            if (idx < 0) continue;
            const size_t off = (tok0 + idx) * C + wave * 32 + 4 * h;
            u32x2 res[4];                                                  // residual = the raw row still sitting in LDS
#pragma unroll
            for (int g = 0; g < 4; ++g) res[g] = *(const u32x2*)(raw + slot * ROWB + (((wave * 4 + g) ^ (slot & 15)) << 4) + 8 * h);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 o;
                o[0] = pack_bf16x2(pt[t][4 * g] + __uint_as_float(res[g][0] << 16), pt[t][4 * g + 1] + __uint_as_float(res[g][0] & 0xffff0000u));
                o[1] = pack_bf16x2(pt[t][4 * g + 2] + __uint_as_float(res[g][1] << 16), pt[t][4 * g + 3] + __uint_as_float(res[g][1] & 0xffff0000u));
                *(u32x2*)(p.out + off + 8 * g) = o;
            }
        }
        PROBE_ADD(7);
    }
#ifdef ATTN_CLOCK_PROBE
    if (tid == 0 && g_probe) {
        g_probe[blockIdx.x * 10] = __builtin_amdgcn_s_memtime() - probe_t0;
        g_probe[blockIdx.x * 10 + 1] = (unsigned long long)((nwin - gi + ng - 1) / ng);
        for (int i = 0; i < 8; ++i) g_probe[blockIdx.x * 10 + 2 + i] = probe_sec[i];
    }
#endif
}

}  // namespace

int attn_block_bf16(const bf16_t* x, const float* ln_g, const float* ln_b, const bf16_t* wqkv, const float* bqkv,
                    const bf16_t* wproj, const float* bproj, const float* bias64, const int32_t* src, const uint8_t* region,
                    bf16_t* out, int B, int T, int Cc, int heads, int nW, int N, float eps, float scale, hipStream_t s) {
    if (Cc != C || heads != HEADS) return MASTER_ERR_UNSUPPORTED;
    constexpr int WMAX = (LDS_MAX - TABLES) / (64 * 5);               // table rows that fit next to the tiles and the bias (140)
    const int R = (nW + WMAX - 1) / WMAX, wchunk = (nW + R - 1) / R;
    const int LDS_BYTES = TABLES + wchunk * 64 * 5;
    BlockArgs p{x, ln_g, ln_b, wqkv, bqkv, wproj, bproj, bias64, src, region, out, B, T, nW, N, eps, scale, wchunk, R};
    static LdsLimit lds_limit;
    lds_limit.ensure((const void*)attn_block_c128_kernel, (int)(LDS_BYTES));
    int grid = B * nW;
    if (grid > 256) grid = 256;                      // persistent: one workgroup per CU walks the windows
    if (grid < R) grid = R;                          // every slice needs a workgroup
    hipLaunchKernelGGL(attn_block_c128_kernel, dim3(grid), dim3(256), LDS_BYTES, s, p);
    return master_check_launch("attn_block_bf16");
}
