// Backward of the windowed attention core (window_attn_f32.hip), fp32, one wave per (image, window, head), nothing but two 64-float rows
// in LDS.  Given dO_s (s = one or two value streams sharing one softmax) it recomputes the probabilities and produces
//     dV_s = P^T dO_s,   dP = sum_s dO_s V_s^T,   dS = P o (dP - rowsum(P o dP)),   dQ = scale dS K,   dK = scale dS^T Q,   dBias += dS
// (reference style_transformer.py:127-155 and :536-575 differentiated; the forward kernel documents the gather tables.)
//
// The v_mfma_f32_32x32x2_f32 result has its COLUMN index on the lane and its rows in the 16 accumulator registers, and an accumulator
// tile can be fed back as the B operand of the next product with the contraction running over its register index.  dQ needs dS with the
// contraction over keys, dK and dV need dS / P with the contraction over queries, so both orientations are computed, each from two
// MFMA chains on the same Q / K / V / dO fragments:
//     phase A  (query on the lane):  S^T = K Q^T -> softmax in-lane -> P^T, L = logsumexp;  dP^T = V dO^T;  D = rowsum;  dS^T;  dQ^T = K^T dS^T
//     phase B  (key on the lane):    S = Q K^T,  P = exp2(S - L) (L, D per query row through LDS);  dV^T = dO^T P;  dP = dO V^T;  dS;  dK^T = Q^T dS
// Zero-padded window slots are real keys whose K / V rows are the projection bias (the reference pads before the Linear): their dK / dV
// rows go to per-image pad-slot buffers [B][n_pad][C] (summed by the caller into the bias gradient / the InstanceNorm backward).
// Padded query slots produce no output in the forward, so their dO is zero here.  dS tiles go to a per-(image, window) workspace that the
// caller reduces over images and windows into the [heads][64][64] bias-tile gradient (fixed order: deterministic).
#include "common.h"

namespace {

struct AttnBwdArgs {
    const float* q; const float* k; const float* v0; const float* v1;
    int ldq, ldk, ldv;
    const float* k_pad; const float* v0_pad; const float* v1_pad;
    int k_pad_bs, v_pad_bs;
    const float* bias; const int32_t* src; const uint8_t* region; const int32_t* pad_ord;
    const float* do0; const float* do1; int ldo;
    float* dq; float* dk; float* dv0; float* dv1; int lddq, lddk, lddv;
    float* dk_pad; float* dv0_pad; float* dv1_pad;         // [B][n_pad][C]
    float* dbias_ws;                                       // [B*nW][heads][64][64]
    int B, T, heads, nW, N, n_pad, C;
    float scale;
};

constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ int rowof(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

template <bool DUAL>
__global__ __launch_bounds__(256) void window_attn_bwd_f32_kernel(AttnBwdArgs p) {
    __shared__ float lrow[4][64], drow[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // uniform: the problem's bases become scalar arithmetic (r04)
    const int r = lane & 31, h = lane >> 5;
    const int pid = blockIdx.x * 4 + wave;
    const int total = p.B * p.nW * p.heads;
    if (pid >= total) return;
    const int head = pid % p.heads;
    const int w = (pid / p.heads) % p.nW;
    const int b = pid / (p.heads * p.nW);
    // the window's gather-table row is read ONCE (slot = lane) and handed around with ds_bpermute: every row this kernel gathers used to
    // start with a dependent load of its table entry (the same cure as the forward kernel's, window_attn_f32.hip r03)
    const int my_src = lane < p.N ? p.src[w * 64 + lane] : -1;
    auto src_of = [&](int slot) __attribute__((always_inline)) { return __shfl(my_src, slot, 64); };
    const int32_t* pord = p.pad_ord + w * 64;
    const uint8_t* reg = p.region ? p.region + w * 64 : nullptr;
    const size_t tok0 = (size_t)b * p.T;
    const int coff = head * 32;
    const float sl2 = p.scale * LOG2E;
    const float* bias = p.bias + (size_t)head * 64 * 64;

    // ---- row fragments: lane (slot t*32 + r, half h) holds channels [16h, 16h+16) of its slot's row --------------------------------------
    f32x4 qv[2][4], kv[2][4], vv0[2][4], gv0[2][4], vv1[2][4], gv1[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int slot = t * 32 + r;
        const int idx = src_of(slot);
        const bool key_on = slot < p.N, real = idx >= 0;
        const size_t tok = tok0 + (real ? idx : 0);
        const int co = coff + 16 * h;
        const float* kp = real ? p.k + tok * p.ldk + co : p.k_pad + (size_t)b * p.k_pad_bs + co;
        const float* v0p = real ? p.v0 + tok * p.ldv + co : p.v0_pad + (size_t)b * p.v_pad_bs + co;
        const float* v1p = DUAL ? (real ? p.v1 + tok * p.ldv + co : p.v1_pad + (size_t)b * p.v_pad_bs + co) : nullptr;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            qv[t][c] = real ? *(const f32x4*)(p.q + tok * p.ldq + co + 4 * c) : z;
            gv0[t][c] = real ? *(const f32x4*)(p.do0 + tok * p.ldo + co + 4 * c) : z;
            kv[t][c] = key_on ? *(const f32x4*)(kp + 4 * c) : z;
            vv0[t][c] = key_on ? *(const f32x4*)(v0p + 4 * c) : z;
            if (DUAL) {
                gv1[t][c] = real ? *(const f32x4*)(p.do1 + tok * p.ldo + co + 4 * c) : z;
                vv1[t][c] = key_on ? *(const f32x4*)(v1p + 4 * c) : z;
            }
        }
    }

    // =====================================  phase A: query on the lane  =================================================================
    f32x16 st[2][2];                                   // [key tile][query tile], log2 domain
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float* brow = bias + (qt * 32 + r) * 64;
        const unsigned rq = reg ? reg[qt * 32 + r] : 0u;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *(const f32x4*)(brow + kt * 32 + 8 * g + 4 * h);
                const unsigned rk4 = reg ? *(const unsigned*)(reg + kt * 32 + 8 * g + 4 * h) : 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = bv[e] * LOG2E;
                    if (reg && ((rk4 >> (8 * e)) & 0xffu) != rq) v += -100.0f * LOG2E;
                    st[kt][qt][4 * g + e] = v;
                }
            }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    st[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[kt][c][e], qv[qt][c][e] * sl2, st[kt][qt], 0, 0, 0);
    // softmax over keys (in-lane + partner half), L = max + log2(sum)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float m = st[0][qt][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) m = fmaxf(m, st[kt][qt][e]);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float s = 0.0f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float pe = __builtin_amdgcn_exp2f(st[kt][qt][e] - m); st[kt][qt][e] = pe; s += pe; }
        s += __shfl_xor(s, 32, 64);
        const float inv = 1.0f / s;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) st[kt][qt][e] *= inv;
        if (h == 0) lrow[wave][qt * 32 + r] = m + log2f(s);
    }
    // dP^T = V dO^T (both streams add up), D[q] = sum_k P^T dP^T, dS^T = P^T (dP^T - D)
    {
        f32x16 dpt[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int e = 0; e < 16; ++e) dpt[kt][qt][e] = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) {
                        dpt[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv0[kt][c][e], gv0[qt][c][e], dpt[kt][qt], 0, 0, 0);
                        if (DUAL) dpt[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv1[kt][c][e], gv1[qt][c][e], dpt[kt][qt], 0, 0, 0);
                    }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float d = 0.0f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) d += st[kt][qt][e] * dpt[kt][qt][e];
            d += __shfl_xor(d, 32, 64);
            if (h == 0) drow[wave][qt * 32 + r] = d;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) st[kt][qt][e] *= dpt[kt][qt][e] - d;
        }
    }
    // dQ^T[d][q] = scale * sum_key K^T[d][key] dS^T[key][q]: A = K gathered with the channel on the lane, B = the dS^T accumulators
    {
        f32x16 o[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[qt][e] = 0.0f;
        const float* kpad = p.k_pad + (size_t)b * p.k_pad_bs + coff + r;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            float kk[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kt * 32 + rowof(e, h);
                const int idx = src_of(key);
                kk[e] = key < p.N ? (idx >= 0 ? p.k[(tok0 + idx) * p.ldk + coff + r] : *kpad) : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) o[qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[e], st[kt][qt][e], o[qt], 0, 0, 0);
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int slot = qt * 32 + r;
            const int idx = src_of(slot);
            if (idx < 0) continue;
            float* op = p.dq + (tok0 + idx) * p.lddq + coff + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = o[qt][4 * g + e] * p.scale;
                *(f32x4*)(op + 8 * g) = ov;
            }
        }
    }

    // =====================================  phase B: key on the lane  ===================================================================
    // P[q][key] = exp2(S[q][key] - L[q]): tiles [query tile][key tile], rows (registers) = queries
    f32x16 pq[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const unsigned rk = reg ? reg[kt * 32 + r] : 0u;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int qi = qt * 32 + rowof(e, h);
                float v = bias[qi * 64 + kt * 32 + r] * LOG2E;
                if (reg && reg[qi] != rk) v += -100.0f * LOG2E;
                pq[qt][kt][e] = v;
            }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    pq[qt][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(qv[qt][c][e] * sl2, kv[kt][c][e], pq[qt][kt], 0, 0, 0);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float L = lrow[wave][qt * 32 + rowof(e, h)];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) pq[qt][kt][e] = __builtin_amdgcn_exp2f(pq[qt][kt][e] - L);      // raw v_exp_f32: the argument is <= 0 (exp2f adds a 4-instruction range fix-up per value)
        }

    // rows of one key slot: the token row for a real token, the pad-slot buffer for a zero-padded slot
    auto store_key_rows = [&](const f32x16 (&o)[2], float* dtok, int ldd, float* dpad) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int slot = kt * 32 + r;
            const int idx = src_of(slot);                                // before the branch: a cross-lane read wants every lane active
            if (slot >= p.N) continue;
            float* op = idx >= 0 ? dtok + (tok0 + idx) * ldd + coff + 4 * h
                                 : dpad + ((size_t)b * p.n_pad + pord[slot]) * p.C + coff + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = o[kt][4 * g + e];
                *(f32x4*)(op + 8 * g) = ov;
            }
        }
    };
    // O^T[d][key] = sum_q X^T[d][q] M[q][key]: A = X gathered with the channel on the lane (rows of real query slots, else 0)
    auto contract_q = [&](const float* xsrc, int ldx, const f32x16 (&m)[2][2], float mul, f32x16 (&o)[2]) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[kt][e] = 0.0f;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            float xx[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int qi = qt * 32 + rowof(e, h);
                const int idx = src_of(qi);
                xx[e] = idx >= 0 ? xsrc[(tok0 + idx) * ldx + coff + r] : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) o[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(xx[e], m[qt][kt][e], o[kt], 0, 0, 0);
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[kt][e] *= mul;
    };

    {   // dV_s^T = dO_s^T P
        f32x16 o[2];
        contract_q(p.do0, p.ldo, pq, 1.0f, o);
        store_key_rows(o, p.dv0, p.lddv, p.dv0_pad);
        if (DUAL) {
            contract_q(p.do1, p.ldo, pq, 1.0f, o);
            store_key_rows(o, p.dv1, p.lddv, p.dv1_pad);
        }
    }
    {   // dP = dO V^T, dS = P (dP - D)
        f32x16 dp[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int e = 0; e < 16; ++e) dp[qt][kt][e] = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) {
                        dp[qt][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv0[qt][c][e], vv0[kt][c][e], dp[qt][kt], 0, 0, 0);
                        if (DUAL) dp[qt][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv1[qt][c][e], vv1[kt][c][e], dp[qt][kt], 0, 0, 0);
                    }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float d = drow[wave][qt * 32 + rowof(e, h)];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) pq[qt][kt][e] *= dp[qt][kt][e] - d;
            }
    }
    if (p.dbias_ws) {
        float* ws = p.dbias_ws + ((size_t)(b * p.nW + w) * p.heads + head) * 4096;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) ws[(qt * 32 + rowof(e, h)) * 64 + kt * 32 + r] = pq[qt][kt][e];
    }
    {   // dK^T = scale * Q^T dS
        f32x16 o[2];
        contract_q(p.q, p.ldq, pq, p.scale, o);
        store_key_rows(o, p.dk, p.lddk, p.dk_pad);
    }
}

}  // namespace

int window_attn_bwd_f32(const float* q, const float* k, const float* v0, const float* v1, int ldq, int ldk, int ldv, const float* k_pad,
                        const float* v0_pad, const float* v1_pad, int k_pad_bs, int v_pad_bs, const float* bias, const int32_t* src,
                        const uint8_t* region, const int32_t* pad_ord, const float* do0, const float* do1, int ldo, float* dq, float* dk,
                        float* dv0, float* dv1, int lddq, int lddk, int lddv, float* dk_pad, float* dv0_pad, float* dv1_pad, float* dbias_ws,
                        int B, int T, int heads, int nW, int N, int n_pad, float scale, hipStream_t s) {
    AttnBwdArgs p{q, k, v0, v1, ldq, ldk, ldv, k_pad, v0_pad, v1_pad, k_pad_bs, v_pad_bs, bias, src, region, pad_ord, do0, do1, ldo,
                  dq, dk, dv0, dv1, lddq, lddk, lddv, dk_pad, dv0_pad, dv1_pad, dbias_ws, B, T, heads, nW, N, n_pad, heads * 32, scale};
    const int total = B * nW * heads;
    const int grid = (total + 3) / 4;
    if (v1) hipLaunchKernelGGL(window_attn_bwd_f32_kernel<true>, dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(window_attn_bwd_f32_kernel<false>, dim3(grid), dim3(256), 0, s, p);
    return master_check_launch("window_attn_bwd_f32");
}
