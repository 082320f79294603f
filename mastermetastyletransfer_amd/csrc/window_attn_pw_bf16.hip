// Persistent window-attention core, bf16, C = 256 / 8 heads (StyleTransformer and Swin stage 2; one or two value
// streams sharing the softmax: the sigma/mu cross attention, reference style_transformer.py:536-607, and the
// Scale/Shift calls of the StyleEncoder :867-882).
//
// The one-problem-per-wave kernel (window_attn_bf16.hip) is latency-bound: every wave walks a chain of dependent
// global loads (table -> rows -> bias) with two waves per SIMD to hide it (3 % MFMA busy).  Here a workgroup is HALF A
// WINDOW (4 waves = 4 of its 8 heads; two workgroups are resident per CU, so a CU still runs 8 heads at a time but is fed in
// half-window units: 800 windows on 256 CUs then cost 3.5 window-times instead of 4) and stays resident on its head group:
// while window i computes, the 64-slot Q / K / V(/V1) tiles of window i+1 (16 KB each, the group's 128 channels) are
// gathered by LDS-DMA with the pad/roll/partition table as per-lane source
// address (padded slots fetch the projection-bias row).  Per window a wave
//   1. pulls its head's Q, K and V fragments out of LDS (16-B chunks source-swizzled by row&31: conflict-free;
//      V through ds_read_b64_tr_b16, the hardware transpose),
//   2. frees the tiles (barrier) and fires the next window's DMA,
//   3. S^T = K Q^T, + rel-pos bias (head-constant: held packed in 32 VGPRs for the whole kernel) + shift mask,
//      softmax with the query on the lane, O^T = V^T P^T from the packed accumulator tile,
//   4. stores its 32 channels of the real query tokens.
// Gather/mask tables of the geometry sit in LDS; the loop has no ordinary global load.
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int HEADS = 8, HPW = 4;                // heads of the layer; heads (= waves) per workgroup
constexpr int ROWB = HPW * 32 * 2;               // bytes of a workgroup's slice of a token row: 256
constexpr int TILE = 64 * ROWB;                  // one 64-slot tile of the workgroup's heads: 16 KB
constexpr float LOG2E = 1.4426950408889634f;

struct PwArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v0; const bf16_t* v1;
    int ldq, ldk, ldv;
    const bf16_t* k_pad; const bf16_t* v0_pad; const bf16_t* v1_pad; int k_pad_bs, v_pad_bs;
    const float* bias; const int32_t* src; const uint8_t* region;
    bf16_t* out0; bf16_t* out1; int ldo;
    int B, T, nW, N; float scale_log2e;
};

union Frag { u32x4 u; bf16x8_t b; };

#ifdef PW_CLOCK_PROBE     // diagnostic build only (profiles/tools/pw_probe.py): s_memtime around the QK^T MFMAs of every (window, head), per wave
__device__ unsigned long long* g_pw_probe = nullptr;           // [grid * 4 waves][6]: QK^T clocks, QK^T + bias clocks, windows, wave clocks, wave time (100 MHz ticks), -
#define PW_STAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

template <bool DUAL>
__global__ __launch_bounds__(HPW * 64, 2) void window_attn_pw_kernel(PwArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // Q | K | V0 | (V1) tiles, then src[nW][64] i32, region[nW][64] u8
    constexpr int NT = DUAL ? 4 : 3;
    unsigned char* Qs = lds;
    unsigned char* Ks = lds + TILE;
    unsigned char* V0s = lds + 2 * TILE;
    unsigned char* V1s = lds + 3 * TILE;
    int32_t* srcs = (int32_t*)(lds + NT * TILE);
    uint8_t* regs = (uint8_t*)(srcs + p.nW * 64);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // head within the workgroup's group
    const int r = lane & 31, h = lane >> 5;
    const int hg = blockIdx.x % (HEADS / HPW);                            // this workgroup's head group: heads hg*HPW .. +HPW-1 (its bias stays in registers)
    const int head = hg * HPW + wave;
    const int nwin = p.B * p.nW;

    for (int i = tid; i < p.nW * 64; i += HPW * 64) { srcs[i] = p.src[i]; regs[i] = p.region ? p.region[i] : 0; }
    // Relative-position bias of this head, added on the MATRIX CORE (this kernel is VALU-bound, its matrix cores ~7 % busy):
    //   S^T[key][query] += sum_q' BT[key][q'] I[q'][query],   BT[key][q'] = bias[q'][key] / scale  (S stays in raw q.k units; the
    // softmax scale and log2 e are applied once, inside the exp2 argument).  For the (key tile k2, query tile q2) accumulator only
    // q' in tile q2 contributes: two k-steps, A = BT fragment (lane = key row, 8 consecutive q'; packed bf16, 32 VGPRs for the
    // whole kernel, the same budget the packed table had), B = identity fragment (lane = query, one 1.0 in one of the two k-steps).
    // The table is trained at std 0.02; bf16 keeps 3 significant digits of it, like every other operand in this mode.  Entries for
    // tile-padding keys are -1e30 and stay a large negative bf16.
    Frag bfr[2][2][2];                                                    // [key tile][query tile][k-step]
    {
        const float* bias = p.bias + (size_t)head * 64 * 64;
        const float inv_scale = 1.0f / p.scale_log2e * LOG2E;            // bias is in natural-log units: bias * log2e / (scale * log2e)
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
    Frag idf[2];                                                          // identity: element j of k-step ks is 1 iff 16 ks + 8 h + j == r
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q0 = 16 * ks + 8 * h + 2 * j;
            idf[ks].u[j] = (q0 == r ? 0x00003f80u : 0u) | (q0 + 1 == r ? 0x3f800000u : 0u);
        }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- LDS-DMA gather of one window, the channels of this workgroup's heads: every tile = 64 rows x 256 B = 16 pieces of 1 KB
    //      (4 rows each); wave w moves pieces 4w..4w+3 of each tile.  Position p of a row holds source chunk p ^ (row & 15). ---------
    // Address arithmetic per piece (r02): one 24-bit multiply for the token row (tokens < 2^24, row pitch < 2^24 bytes: checked by the
    // launcher) and 32-bit offsets from wave-uniform bases; a padded slot selects the pad row's (base, offset) pair instead.  The
    // 64-bit multiply-adds the compiler made of  p.k + tk * p.ldk + ...  were ~40 quarter-rate instructions per window in a kernel
    // that is issue-bound (profiles/r02_window_attn_bound_study.txt).
    const bool same_ld = p.ldq == p.ldk && p.ldq == p.ldv;
    auto gather = [&](int win) __attribute__((always_inline)) {
        const int b = win / p.nW, w = win - b * p.nW;
        const int32_t* src = srcs + w * 64;
        const unsigned tok0 = (unsigned)b * (unsigned)p.T;
        const unsigned kpad0 = (unsigned)b * (unsigned)p.k_pad_bs * 2u, vpad0 = (unsigned)b * (unsigned)p.v_pad_bs * 2u;     // bytes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int I = wave * 4 + i;
            const int row = 4 * I + (lane >> 4);
            const unsigned qcb = (unsigned)(hg * (HPW * 4) + ((lane & 15) ^ (row & 15))) * 16u;   // byte offset of the lane's source chunk (8 channels) in a token row
            const int idx = row < p.N ? src[row] : -1;
            const bool real = idx >= 0;
            const unsigned tk = tok0 + (real ? (unsigned)idx : 0u);
            const unsigned oq = __umul24(tk, (unsigned)p.ldq * 2u) + qcb;           // padded query rows are never stored: any row will do
            const unsigned ok = same_ld ? oq : __umul24(tk, (unsigned)p.ldk * 2u) + qcb;
            const unsigned ov = same_ld ? oq : __umul24(tk, (unsigned)p.ldv * 2u) + qcb;
            const char* sq = (const char*)p.q + oq;
            const char* sk = real ? (const char*)p.k + ok : (const char*)p.k_pad + (kpad0 + qcb);
            const char* s0 = real ? (const char*)p.v0 + ov : (const char*)p.v0_pad + (vpad0 + qcb);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sq, (__attribute__((address_space(3))) void*)(Qs + I * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sk, (__attribute__((address_space(3))) void*)(Ks + I * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s0, (__attribute__((address_space(3))) void*)(V0s + I * 1024), 16, 0, 0);
            if (DUAL) {
                const char* s1 = real ? (const char*)p.v1 + ov : (const char*)p.v1_pad + (vpad0 + qcb);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s1, (__attribute__((address_space(3))) void*)(V1s + I * 1024), 16, 0, 0);
            }
        }
    };

    const int wstride = gridDim.x / (HEADS / HPW);                       // workgroups serving this head group
    int win = blockIdx.x / (HEADS / HPW);
    if (win < nwin) gather(win);

    const int g16 = (lane >> 4) & 1, i16 = lane & 15;
#ifdef PW_CLOCK_PROBE
    unsigned long long pw_qk = 0, pw_qkb = 0, pw_n = 0, pw_t0, pw_t1, pw_t2, pw_k0;
    const unsigned long long pw_r0 = __builtin_amdgcn_s_memrealtime();
    PW_STAMP(pw_k0);
#endif
    const float mask_raw = -100.0f * LOG2E / p.scale_log2e;              // the -100 of the shift mask in raw q.k units
    Frag ones; ones.u = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};      // bf16 1.0 x 8
    for (; win < nwin; win += wstride) {
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
                const int pos = ((wave * 4 + 2 * s + h) ^ (row & 15)) << 4;   // chunk wave*4 + 2s + h of the workgroup's 16 chunks
                qf[t][s].u = *(const u32x4*)(Qs + row * ROWB + pos);
                kf[t][s].u = *(const u32x4*)(Ks + row * ROWB + pos);
            }
        }
        // V^T fragments by transposed reads: k-step (k2, s) covers keys 32k2 + 16s + 8jj + 4h + (0..3); lane 4q+p of each
        // 16-lane group addresses row key0+q, channels 16*g16 + 4p .. +3 of this head
        union VF { s16x4 v[2]; bf16x8_t b; };
        VF vf0[2][2], vf1[2][2];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int row = 32 * k2 + 16 * s + 8 * jj + 4 * h + (i16 >> 2);
                    const int chunk = wave * 4 + 2 * g16 + ((i16 & 3) >> 1);
                    const int off = row * ROWB + ((chunk ^ (row & 15)) << 4) + 8 * (i16 & 1);
                    vf0[k2][s].v[jj] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(V0s + off));
                    if (DUAL) vf1[k2][s].v[jj] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(V1s + off));
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // every head has its fragments: the tiles are free
        if (win + wstride < nwin) gather(win + wstride);               // next window's rows fly under this window's math

        // ---- 2. S^T[key][query] = K Q^T, scaled, + bias + shift mask -----------------------------------------------------
        f32x16 st[2][2];
#ifdef PW_CLOCK_PROBE
        // the named GEMM alone: the 8 MFMAs of S^T = K Q^T (4 tiles x 2 k-steps) back to back, a dependent read of the last accumulator
        // register (issue stalls until the last MFMA has written it), stamp; then the 8 bias-through-identity MFMAs the same way
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                for (int e = 0; e < 16; ++e) st[k2][q2][e] = 0.f;
        PW_STAMP(pw_t0);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    st[k2][q2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[k2][s].b, qf[q2][s].b, st[k2][q2], 0, 0, 0);
        { float sink; asm volatile("v_mov_b32 %0, %1" : "=v"(sink) : "v"(st[1][1][15])); asm volatile("" :: "v"(sink)); }
        PW_STAMP(pw_t1);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    st[k2][q2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[k2][q2][ks].b, idf[ks].b, st[k2][q2], 0, 0, 0);
        { float sink; asm volatile("v_mov_b32 %0, %1" : "=v"(sink) : "v"(st[1][1][15])); asm volatile("" :: "v"(sink)); }
        PW_STAMP(pw_t2);
        pw_qk += pw_t1 - pw_t0; pw_qkb += pw_t2 - pw_t0; ++pw_n;
#else
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
                for (int ks = 0; ks < 2; ++ks)                             // + bias / scale, through the identity
                    st[k2][q2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[k2][q2][ks].b, idf[ks].b, st[k2][q2], 0, 0, 0);
            }
#endif
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
            const float nmx = -mx * p.scale_log2e;                        // scale, log2 e and the max in one fma per element
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        pf[q2][k2][s].u[j] = pack_bf16x2(__builtin_amdgcn_exp2f(fmaf(st[k2][q2][8 * s + 2 * j], p.scale_log2e, nmx)),
                                                         __builtin_amdgcn_exp2f(fmaf(st[k2][q2][8 * s + 2 * j + 1], p.scale_log2e, nmx)));
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
        // ---- 4. O^T[d][query] = V^T P^T per value stream, stores for real query tokens -------------------------------------------
        auto pv_store = [&](VF (&vf)[2][2], bf16_t* out) __attribute__((always_inline)) {
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
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int slot = q2 * 32 + r;
                const int idx = slot < p.N ? src[slot] : -1;
                if (idx < 0) continue;
                bf16_t* op = out + (tok0 + idx) * p.ldo + head * 32 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 o;
                    o[0] = pack_bf16x2(ot[q2][4 * g] * inv[q2], ot[q2][4 * g + 1] * inv[q2]);
                    o[1] = pack_bf16x2(ot[q2][4 * g + 2] * inv[q2], ot[q2][4 * g + 3] * inv[q2]);
                    *(u32x2*)(op + 8 * g) = o;
                }
            }
        };
        pv_store(vf0, p.out0);
        if (DUAL) pv_store(vf1, p.out1);
    }
#ifdef PW_CLOCK_PROBE
    if (lane == 0 && g_pw_probe) {
        unsigned long long pw_k1; PW_STAMP(pw_k1);
        unsigned long long* dst = g_pw_probe + ((size_t)blockIdx.x * HPW + wave) * 6;
        dst[0] = pw_qk; dst[1] = pw_qkb; dst[2] = pw_n; dst[3] = pw_k1 - pw_k0; dst[4] = __builtin_amdgcn_s_memrealtime() - pw_r0;
    }
#endif
}

}  // namespace

#ifdef PW_CLOCK_PROBE
extern "C" int master_pw_probe_buffer(void* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_pw_probe), &buf, sizeof(buf)) == hipSuccess ? 0 : -1; }
#endif

// returns MASTER_ERR_UNSUPPORTED when the geometry is outside this kernel (caller falls back to window_attn_bf16)
int window_attn_pw_bf16(const bf16_t* q, const bf16_t* k, const bf16_t* v0, const bf16_t* v1, int ldq, int ldk, int ldv,
                        const bf16_t* k_pad, const bf16_t* v0_pad, const bf16_t* v1_pad, int k_pad_bs, int v_pad_bs,
                        const float* bias, const int32_t* src, const uint8_t* region, bf16_t* out0, bf16_t* out1, int ldo,
                        int B, int T, int heads, int nW, int N, float scale, hipStream_t s) {
    if (heads != HEADS) return MASTER_ERR_UNSUPPORTED;
    // the gather's 24-bit x 24-bit row offsets: token count, row pitches in bytes and every row offset below 2^24 / 2^32
    if ((long long)B * T >= (1ll << 24) || ldq * 2 >= (1 << 24) || ldk * 2 >= (1 << 24) || ldv * 2 >= (1 << 24) ||
        (long long)B * T * 2 * (ldq > ldk ? (ldq > ldv ? ldq : ldv) : (ldk > ldv ? ldk : ldv)) >= (1ll << 32) ||
        (long long)B * 2 * (k_pad_bs > v_pad_bs ? k_pad_bs : v_pad_bs) >= (1ll << 32)) return MASTER_ERR_UNSUPPORTED;
    const int lds = (v1 ? 4 : 3) * TILE + nW * 64 * 5;
    if (lds > 160 * 1024) return MASTER_ERR_UNSUPPORTED;
    PwArgs p{q, k, v0, v1, ldq, ldk, ldv, k_pad, v0_pad, v1_pad, k_pad_bs, v_pad_bs, bias, src, region, out0, out1, ldo,
             B, T, nW, N, scale * LOG2E};
    // two workgroups (head groups) per window, two workgroups resident per CU: 512 slots; even grid so that both groups are served
    int grid = B * nW * (HEADS / HPW);
    if (grid > 512) grid = 512;
    static LdsLimit lim1, lim2;
    if (v1) {
        lim2.ensure((const void*)window_attn_pw_kernel<true>, lds);
        hipLaunchKernelGGL(window_attn_pw_kernel<true>, dim3(grid), dim3(HPW * 64), lds, s, p);
    } else {
        lim1.ensure((const void*)window_attn_pw_kernel<false>, lds);
        hipLaunchKernelGGL(window_attn_pw_kernel<false>, dim3(grid), dim3(HPW * 64), lds, s, p);
    }
    return master_check_launch("window_attn_pw_bf16");
}
