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
// WHOLE window (8 waves = 8 heads, two per SIMD) and stays resident: while window i computes, the 64-slot Q / K / V tiles of window
// i + 1 (32 KB each) are gathered by LDS-DMA through the pad / roll / partition table (padded slots fetch the projection-bias row).
// Per window a wave
//   1. pulls its head's Q, K and V fragments out of LDS (16-B chunks source-swizzled by row & 15; V through ds_read_b64_tr_b16),
//   2. frees the tiles (barrier) and fires the next window's DMA,
//   3. S^T = K Q^T (+ rel-pos bias through an identity MFMA, shift mask), softmax with the query on the lane, O^T = V^T P^T from the
//      packed accumulator tile - all as in window_attn_pw_bf16.hip,
//   4. fetches its 32 rows of Wproj (16 KB, L2-resident; the registers of the score tiles are free by then),
//   5. writes its head's normalised O rows into the shared O tile (bf16 [slot][256], swizzled), barrier,
//   6. projects: out^T[32 channels of this wave][64 slots] = Wproj rows . O^T, 32 MFMAs with the O fragments streamed from LDS,
//   7. adds bias + residual and stores its 32 channels of the real query tokens.
// Three barriers per window; the loop's only ordinary global loads are the Wproj rows and the residual (both issued early).
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int HEADS = 8, C = 256;
constexpr int ROWB = C * 2;                      // bytes of a token row: 512
constexpr int TILE = 64 * ROWB;                  // one 64-slot tile: 32 KB
constexpr int NTHREADS = HEADS * 64;
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

__global__ __launch_bounds__(NTHREADS) void window_attn_proj_kernel(ApArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // Q | K | V | O tiles, then src[nW][64] i32, region[nW][64] u8
    unsigned char* Qs = lds;
    unsigned char* Ks = lds + TILE;
    unsigned char* Vs = lds + 2 * TILE;
    unsigned char* Os = lds + 3 * TILE;
    int32_t* srcs = (int32_t*)(lds + 4 * TILE);
    uint8_t* regs = (uint8_t*)(srcs + p.nW * 64);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // = head; also the 32 output channels this wave projects
    const int r = lane & 31, h = lane >> 5;
    const int nwin = p.B * p.nW;

    for (int i = tid; i < p.nW * 64; i += NTHREADS) { srcs[i] = p.src[i]; regs[i] = p.region ? p.region[i] : 0; }
    // relative-position bias of this head as MFMA fragments, added through an identity fragment (window_attn_pw_bf16.hip)
    Frag bfr[2][2][2];                                                    // [key tile][query tile][k-step]
    {
        const float* bias = p.bias + (size_t)wave * 64 * 64;
        const float inv_scale = 1.0f / p.scale_log2e * LOG2E;
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
                    for (int j = 0; j < 4; ++j) bfr[k2][q2][ks].u[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
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
    float bp[16];                                                         // proj bias of accumulator register e: channel wave*32 + (e&3) + 8(e>>2) + 4h
#pragma unroll
    for (int e = 0; e < 16; ++e) bp[e] = p.bproj[wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- LDS-DMA gather of one window: every tile = 64 rows x 512 B = 32 pieces of 1 KB (2 rows each); wave w moves pieces 4w..4w+3
    //      of each tile.  Position c of a row (32 chunks of 16 B) holds source chunk c ^ (row & 15). ---------------------------------
    const bool same_ld = p.ldq == p.ldk && p.ldq == p.ldv;
    auto gather = [&](int win) __attribute__((always_inline)) {
        const int b = win / p.nW, w = win - b * p.nW;
        const int32_t* src = srcs + w * 64;
        const unsigned tok0 = (unsigned)b * (unsigned)p.T;
        const unsigned kpad0 = (unsigned)b * (unsigned)p.k_pad_bs * 2u, vpad0 = (unsigned)b * (unsigned)p.v_pad_bs * 2u;     // bytes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int I = wave * 4 + i;
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
    if (win < nwin) gather(win);

    const int g16 = (lane >> 4) & 1, i16 = lane & 15;
    const float mask_raw = -100.0f * LOG2E / p.scale_log2e;              // the -100 of the shift mask in raw q.k units
    Frag ones; ones.u = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};      // bf16 1.0 x 8
    const bf16_t* wrow = p.wproj + (size_t)(wave * 32 + r) * C + 8 * h;   // this lane's row of Wproj (A fragments: row = output channel)
    for (; win < nwin; win += gridDim.x) {
        const int b = win / p.nW, w = win - b * p.nW;
        const int32_t* src = srcs + w * 64;
        const size_t tok0 = (size_t)b * p.T;

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // this window's tiles have landed (every wave waited for its own pieces)

        // ---- 1. fragments of this head out of LDS ---------------------------------------------------------------------
        Frag qf[2][2], kf[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = t * 32 + r;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int pos = ((wave * 4 + 2 * s + h) ^ (row & 15)) << 4;   // chunk wave*4 + 2s + h of the row's 32 chunks
                qf[t][s].u = *(const u32x4*)(Qs + row * ROWB + pos);
                kf[t][s].u = *(const u32x4*)(Ks + row * ROWB + pos);
            }
        }
        union VF { s16x4 v[2]; bf16x8_t b; };
        VF vf[2][2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int row = 32 * k2 + 16 * s + 8 * jj + 4 * h + (i16 >> 2);
                    const int chunk = wave * 4 + 2 * g16 + ((i16 & 3) >> 1);
                    const int off = row * ROWB + ((chunk ^ (row & 15)) << 4) + 8 * (i16 & 1);
                    vf[k2][s].v[jj] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(Vs + off));
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // every head has its fragments: the tiles are free
        if (win + (int)gridDim.x < nwin) gather(win + gridDim.x);          // next window's rows fly under this window's math

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
                    st[k2][q2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[k2][s].b, qf[q2][s].b, st[k2][q2], 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    st[k2][q2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[k2][q2][ks].b, idf[ks].b, st[k2][q2], 0, 0, 0);
            }
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
                            if (((rk4 >> (8 * e)) & 0xffu) != rq) st[k2][q2][4 * g + e] += mask_raw;
                    }
            }
        }
        // ---- 3. softmax over keys (query on the lane), probabilities packed as the B operand of P.V ------------------------------
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
        // ---- 4. O^T[d][query] = V^T P^T ----------------------------------------------------------------------------------------------
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
                    ot[q2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[k2][s].b, pf[q2][k2][s].b, ot[q2], 0, 0, 0);
        // ---- 5. this wave's rows of Wproj and its residual values: issued now (score / probability tiles are dead: with them live the
        //         64 + 16 registers spill), they fly under the O exchange and are used after its barrier -----------------------------------
        __builtin_amdgcn_sched_barrier(0);
        Frag wp[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) wp[s].u = *(const u32x4*)(wrow + 16 * s);
        int sidx[2];
        u32x2 rsd[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int slot = t * 32 + r;
            sidx[t] = slot < p.N ? src[slot] : -1;
#pragma unroll
            for (int g = 0; g < 4; ++g) rsd[t][g] = u32x2{0u, 0u};
            if (p.res && sidx[t] >= 0) {
                const bf16_t* rp = p.res + (tok0 + sidx[t]) * p.ldr + wave * 32 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) rsd[t][g] = *(const u32x2*)(rp + 8 * g);
            }
        }
        // ---- normalised O rows into the shared O tile: O[slot][head*32 + d], 8-B pieces d = 8g + 4h + (0..3) ---------------------------
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
            const int slot = q2 * 32 + r;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 o;
                o[0] = pack_bf16x2(ot[q2][4 * g] * inv[q2], ot[q2][4 * g + 1] * inv[q2]);
                o[1] = pack_bf16x2(ot[q2][4 * g + 2] * inv[q2], ot[q2][4 * g + 3] * inv[q2]);
                *(u32x2*)(Os + slot * ROWB + (((wave * 4 + g) ^ (slot & 15)) << 4) + 8 * h) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // all heads' O rows are in LDS (the previous window's proj reads ended two barriers ago)

        // ---- 6. proj: out^T[n][slot] for this wave's 32 output channels, O fragments streamed from LDS ----------------------------
        f32x16 pt[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) pt[t][e] = bp[e];
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
                pt[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[s].b, o0.b, pt[0], 0, 0, 0);
                pt[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[s].b, o1.b, pt[1], 0, 0, 0);
            }
        }
        // ---- 7. + residual, stores of the real query tokens (lane = slot, 8-B pieces of this wave's 32 channels) -----------------
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (sidx[t] < 0) continue;
            bf16_t* op = p.out + (tok0 + sidx[t]) * p.ldo + wave * 32 + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 o;
                o[0] = pack_bf16x2(pt[t][4 * g] + __uint_as_float(rsd[t][g][0] << 16), pt[t][4 * g + 1] + __uint_as_float(rsd[t][g][0] & 0xffff0000u));
                o[1] = pack_bf16x2(pt[t][4 * g + 2] + __uint_as_float(rsd[t][g][1] << 16), pt[t][4 * g + 3] + __uint_as_float(rsd[t][g][1] & 0xffff0000u));
                *(u32x2*)(op + 8 * g) = o;
            }
        }
    }
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
    const int lds = 4 * TILE + nW * 64 * 5;
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
