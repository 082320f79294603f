// Windowed attention core, fp32 parity mode: one wave per (batch, window, head), nothing in LDS.
//
// Swapped product S^T = K Q^T on v_mfma_f32_32x32x2_f32: the C/D map puts the QUERY on the lane
// (col = lane&31) and the KEYS in the 16 accumulator registers (row = (reg&3) + 8*(reg>>2) + 4*half),
// so each lane owns 32 of one query's 64 (padded) keys and its half-wave partner the other 32:
// softmax max/sum = 31 in-lane ops + one cross-half exchange.  The probabilities never leave the
// accumulators: for O^T = V^T P^T the accumulator tile IS the B operand (k-step <-> register), and
// the A operand V^T[d][key] is read straight from global memory with the channel on the lane
// (128-B coalesced rows).  Pad -> roll -> window partition (reference style_transformer.py:83-111)
// is a per-window gather table; zero-padded slots take the projection bias as their K/V row because
// the reference pads before the Linear.  Outputs are written back un-windowed, for real tokens only
// (reverse partition/roll/un-pad, :160-168).
//
// X3 (the split-bf16 fast parity mode, MASTER_F32X3; r04): the same dataflow with both products on v_mfma_f32_32x32x16_bf16 - every
// operand split in registers into bf16 hi + lo, three MFMAs per product (a_lo*b_hi + a_hi*b_lo + a_hi*b_hi), fp32 accumulation:
// 24 + 24 MFMAs of 32 clocks per problem instead of 64 + 64 of 64 clocks (the exact-fp32 form spends ~40 % of its time in the matrix
// pipe).  The channel k order (lane half h holds channels 16h + 8s' + j for k-step s') is the same for K and Q, so it cancels; the
// packed probability tile IS the B fragment of the second product, whose k order (keys 16s + 4h + 0..3, 16s + 8 + 4h + 0..3) is the
// order the V^T registers already have.  Softmax, bias, mask and the normalisation stay fp32.
#include "common.h"

namespace {

struct AttnArgs {
    const float* q; const float* k; const float* v0; const float* v1;
    int ldq, ldk, ldv;
    const float* k_pad; const float* v0_pad; const float* v1_pad;
    int k_pad_bs, v_pad_bs;
    const float* bias;        // [heads][64][64], key >= N pre-set to -1e30
    const int32_t* src;       // [nW][64]
    const uint8_t* region;    // [nW][64] or null
    float* out0; float* out1; int ldo;
    int B, T, heads, nW, N;
    float scale_log2e;        // head_dim^-0.5 * log2(e): softmax evaluated with exp2
};

constexpr float LOG2E = 1.4426950408889634f;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct FragX { u32x4 u; };       // 8 sixteen-bit halves: bf16 or IEEE half, by FMT

// 8 fp32 -> the bf16 hi and lo fragments (round to nearest even both times)
template <int FMT>
__device__ __forceinline__ void split_frag(const float (&x)[8], FragX& hi, FragX& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned a, l;
        split_x2<FMT>(x[2 * j], x[2 * j + 1], a, l);
        hi.u[j] = a; lo.u[j] = l;
    }
}

template <bool DUAL, bool X3, int FMT = X3_BF16>
__global__ __launch_bounds__(256) void window_attn_f32_kernel(AttnArgs p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);    // uniform: the problem's bases and table pointers become scalar arithmetic
    const int r = lane & 31, h = lane >> 5;
    // problem id: heads of one window are adjacent waves (same gather table, same rows of K/V in L2)
    const int pid = blockIdx.x * 4 + wave;
    const int total = p.B * p.nW * p.heads;
    if (pid >= total) return;
    const int head = pid % p.heads;
    const int w = (pid / p.heads) % p.nW;
    const int b = pid / (p.heads * p.nW);
    const int32_t* src = p.src + w * 64;
    const size_t tok0 = (size_t)b * p.T;
    const int coff = head * 32;
    // r03: the window's gather table is read ONCE (one coalesced load, slot = lane) and handed around with shuffles / readlanes.  The
    // kernel is a chain of dependent loads - table entry -> token row - and r02 walked it 100 times per problem (2 557 instructions,
    // 79 vmcnt waits, ~70 k clocks for 8 k clocks of MFMA); with the table in a register every row address is known at once, and the
    // V rows are requested before the S^T MFMAs instead of after the softmax.
    const int my_src = lane < p.N ? src[lane] : -1;

    // ---- Q (B operand of S^T = K Q^T) and K (A operand): lane (row r, half h) holds channels
    //      [16h, 16h+16) of its token row; k-step s pairs channel s with channel 16+s.
    f32x4 qv[2][4], kv[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int slot = t * 32 + r;
        const int idx = __shfl(my_src, slot, 64);
        // r04: one 64-bit base per problem (uniform: scalar registers) + a 32-bit byte offset per lane - an image's rows span < 4 GB - so that
        // the loads take the saddr + voffset form instead of a 64-bit multiply-add per row (the kernel is VALU-issue bound: 2 210 vector
        // instructions per problem for 48 MFMAs)
        const float* qp = (const float*)((const char*)(p.q + tok0 * p.ldq) + (unsigned)(((idx >= 0 ? idx : 0) * p.ldq + coff + 16 * h) * 4));
        const float* kp = idx >= 0 ? (const float*)((const char*)(p.k + tok0 * p.ldk) + (unsigned)((idx * p.ldk + coff + 16 * h) * 4))
                                   : p.k_pad + (size_t)b * p.k_pad_bs + coff + 16 * h;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#ifdef ATTN_PROBE_NOLOAD
            qv[t][c] = f32x4{(float)lane, 1.f, (float)c, 0.5f} * 1e-3f; kv[t][c] = f32x4{(float)t, 1.f, (float)lane, 0.25f} * 1e-3f; (void)qp; (void)kp;
#else
            qv[t][c] = *(const f32x4*)(qp + 4 * c);     // padded query slots: any finite row, never stored
            kv[t][c] = *(const f32x4*)(kp + 4 * c);     // key slots >= N: masked by bias = -1e30
#endif
        }
    }

    // ---- V^T rows (A operand of O^T = V^T P^T: channel on the lane, one key per k-step = accumulator register (kt, e)): requested
    //      now, consumed after the softmax
    float vv0[2][16], vv1[2][16];
    // r04: the row of key (kt, e) is the same for all 32 lanes of a half-wave, so its byte offset comes from the lane that holds the slot
    // (ONE ds_bpermute; the LDS pipe is otherwise idle here) instead of two readlanes + a per-half select + a multiply, and whether the slot
    // is a real token is a SCALAR bit test on a ballot of the table (the padded slots take the pad row through one v_cndmask on a scalar
    // lane mask): 2 vector instructions per V element instead of 8 in a kernel that is bound by their count.
    const int my_voff = my_src >= 0 ? my_src * (p.ldv * 4) : 0;              // padded slots read row 0 (any valid address), then take the pad value
    const unsigned long long real_slots = __ballot(my_src >= 0);
    const int bp_sel = 16 * h;                                               // ds_bpermute address of lane 4 h
    auto load_v = [&](const float* v, const float* v_pad, float (&vv)[2][16]) __attribute__((always_inline)) {
        const float padv = v_pad[(size_t)b * p.v_pad_bs + coff + r];
        const char* vb = (const char*)(v + tok0 * p.ldv);
        const int lane_off = (coff + r) * 4;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key0 = kt * 32 + (e & 3) + 8 * (e >> 2);                   // half 0's key; half 1 takes key0 + 4
                const int off = __builtin_amdgcn_ds_bpermute(bp_sel + 4 * key0, my_voff);
#ifdef ATTN_PROBE_NOLOAD
                const float a = (float)(off & 1023) * 1e-3f + (float)lane_off; (void)vb;
#else
                const float a = *(const float*)(vb + (unsigned)(off + lane_off));
#endif
                // lanes 0-31: is slot key0 real, lanes 32-63: is slot key0 + 4 real (slots >= N count as padded: they read the finite pad
                // row and their probability is exactly 0 - bias -1e30)
                const unsigned long long m = (((real_slots >> key0) & 1ull) ? 0x00000000ffffffffull : 0ull) |
                                             (((real_slots >> (key0 + 4)) & 1ull) ? 0xffffffff00000000ull : 0ull);
                float sel;
                asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(sel) : "v"(padv), "v"(a), "s"(m));
                vv[kt][e] = sel;
            }
    };
    load_v(p.v0, p.v0_pad, vv0);

    // ---- S^T tiles [key tile kt][query tile qt], initialised with bias (+ shift mask), pre-scaled
    //      by log2(e) so that P = exp2(S' - max').
    f32x16 st[2][2];
    const float* bias = p.bias + (size_t)head * 64 * 64;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float* brow = bias + (qt * 32 + r) * 64;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *(const f32x4*)(brow + kt * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) st[kt][qt][4 * g + e] = bv[e] * LOG2E;
            }
    }
    if (p.region) {
        const uint8_t* reg = p.region + w * 64;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const unsigned rq = reg[qt * 32 + r];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned rk4 = *(const unsigned*)(reg + kt * 32 + 8 * g + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (((rk4 >> (8 * e)) & 0xffu) != rq) st[kt][qt][4 * g + e] += -100.0f * LOG2E;
                }
        }
    }
#ifdef ATTN_PROBE_NOCOMPUTE
    {   // loads + stores only: every loaded value folded into the stored rows with one add each
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc += qv[t][c][e] + kv[t][c][e];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc += vv0[kt][e];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int idx = __shfl(my_src, qt * 32 + r, 64);
            if (idx < 0) continue;
            float* op = (float*)((char*)(p.out0 + tok0 * p.ldo) + (unsigned)((idx * p.ldo + coff + 4 * h) * 4));
#pragma unroll
            for (int g = 0; g < 4; ++g) *(f32x4*)(op + 8 * g) = f32x4{acc, acc, acc, acc};
        }
        return;
    }
#endif
    if constexpr (X3) {
        FragX kh[2][2], kl[2][2], qh[2][2], ql[2][2];                     // [tile][k-step s']: channels 16h + 8s' .. +8
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                float kx[8], qx[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    kx[j] = kv[t][2 * sp + (j >> 2)][j & 3];
                    qx[j] = qv[t][2 * sp + (j >> 2)][j & 3] * p.scale_log2e;
                }
                split_frag<FMT>(kx, kh[t][sp], kl[t][sp]);
                split_frag<FMT>(qx, qh[t][sp], ql[t][sp]);
            }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    st[kt][qt] = mfma_x3_32<FMT>(kl[kt][sp].u, qh[qt][sp].u, st[kt][qt]);
                    st[kt][qt] = mfma_x3_32<FMT>(kh[kt][sp].u, ql[qt][sp].u, st[kt][qt]);
                    st[kt][qt] = mfma_x3_32<FMT>(kh[kt][sp].u, qh[qt][sp].u, st[kt][qt]);
                }
    } else {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    st[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[kt][c][e], qv[qt][c][e] * p.scale_log2e,
                                                                      st[kt][qt], 0, 0, 0);
    }

    if constexpr (DUAL) load_v(p.v1, p.v1_pad, vv1);      // second value stream: requested once the Q / K registers are free, lands under the softmax

    // ---- softmax over keys: 32 in-lane values + the partner half (lane ^ 32) ------------------
    float inv[2];
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
            for (int e = 0; e < 16; ++e) {
                const float pe = __builtin_amdgcn_exp2f(st[kt][qt][e] - m);        // raw v_exp_f32: the argument is <= 0, results under 2^-126 may flush (exp2f's range fix-up was 4 more instructions per value)
                st[kt][qt][e] = pe;
                s += pe;
            }
        s += __shfl_xor(s, 32, 64);
        inv[qt] = 1.0f / s;
    }

    // ---- O^T[d][q] = sum_key V^T[d][key] P^T[key][q]; k-step (kt, reg) covers keys
    //      kt*32 + (reg&3) + 8*(reg>>2) + 4*half, exactly the rows the accumulator register holds.
    // X3: the probabilities as bf16 hi / lo B fragments, built once (both value streams use them): fragment (kt, qt, s) = registers
    // 8s .. 8s+7 of st[kt][qt]
    FragX ph[2][2][2], pl[2][2][2];
    if constexpr (X3) {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    float px[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) px[j] = st[kt][qt][8 * sp + j];
                    split_frag<FMT>(px, ph[kt][qt][sp], pl[kt][qt][sp]);
                }
    }
    auto pv = [&](const float (&vv)[2][16], float* out) {
        f32x16 o[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[qt][e] = 0.0f;
        if constexpr (X3) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    float vx[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) vx[j] = vv[kt][8 * sp + j];
                    FragX vh, vl;
                    split_frag<FMT>(vx, vh, vl);
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) {
                        o[qt] = mfma_x3_32<FMT>(vl.u, ph[kt][qt][sp].u, o[qt]);
                        o[qt] = mfma_x3_32<FMT>(vh.u, pl[kt][qt][sp].u, o[qt]);
                        o[qt] = mfma_x3_32<FMT>(vh.u, ph[kt][qt][sp].u, o[qt]);
                    }
                }
        } else {
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                    o[qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[kt][e], st[kt][qt][e], o[qt], 0, 0, 0);
        }
        // o[qt][e]: query = qt*32 + r (lane), channel d = (e&3) + 8*(e>>2) + 4*h  -> 4 x 16-B stores
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int idx = __shfl(my_src, qt * 32 + r, 64);
            if (idx < 0) continue;
            float* op = (float*)((char*)(out + tok0 * p.ldo) + (unsigned)((idx * p.ldo + coff + 4 * h) * 4));
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = o[qt][4 * g + e] * inv[qt];
                *(f32x4*)(op + 8 * g) = ov;
            }
        }
    };
    pv(vv0, p.out0);
    if constexpr (DUAL) pv(vv1, p.out1);
}

}  // namespace

int window_attn_f32(const float* q, const float* k, const float* v0, const float* v1, int ldq, int ldk, int ldv,
                    const float* k_pad, const float* v0_pad, const float* v1_pad, int k_pad_bs, int v_pad_bs,
                    const float* bias, const int32_t* src, const uint8_t* region, float* out0, float* out1, int ldo,
                    int B, int T, int heads, int nW, int N, float scale, hipStream_t s, int x3) {
    AttnArgs p{q, k, v0, v1, ldq, ldk, ldv, k_pad, v0_pad, v1_pad, k_pad_bs, v_pad_bs, bias, src, region,
               out0, out1, ldo, B, T, heads, nW, N, scale * LOG2E};
    const int total = B * nW * heads;
    const int grid = (total + 3) / 4;
    if (x3 == 2) {                       // IEEE-half halves (MASTER_F16X3)
        if (v1) hipLaunchKernelGGL((window_attn_f32_kernel<true, true, X3_F16>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((window_attn_f32_kernel<false, true, X3_F16>), dim3(grid), dim3(256), 0, s, p);
    } else if (x3) {                     // bf16 halves (MASTER_F32X3)
        if (v1) hipLaunchKernelGGL((window_attn_f32_kernel<true, true, X3_BF16>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((window_attn_f32_kernel<false, true, X3_BF16>), dim3(grid), dim3(256), 0, s, p);
    } else {
        if (v1) hipLaunchKernelGGL((window_attn_f32_kernel<true, false>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((window_attn_f32_kernel<false, false>), dim3(grid), dim3(256), 0, s, p);
    }
    return master_check_launch("window_attn_f32");
}
