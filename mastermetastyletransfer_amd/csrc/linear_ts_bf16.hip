// Token-stationary Linear, bf16 throughput mode:   out[M,N] = r1 (.) (LN?(x) . W^T + bias) + r2
// (the q/k/v/proj projections and PatchMerging's reduction: reference style_transformer.py:114-116,156,
//  515-517,576,595 and the fused qkv of the Swin blocks; LayerNorm = norm1 of the block, folded in).
//
// Same machinery as the fused MLP (mlp_bf16.hip), minus its second phase.  One workgroup = 128 tokens, 8 waves = two per
// SIMD: waves w and w+4 both own tokens [32(w&3), +32) and split every 128-column chunk between them (tiles 0,1 / 2,3).
// A wave issues in order, so with one wave per SIMD every stall (LDS-DMA issue on the 64 B/clk path, barrier, fragment
// wait, epilogue) idled that SIMD's matrix core; with two, one wave's MFMAs run under the other's stalls, at the same LDS
// and L2 traffic.  The product is computed transposed, O^T[n][m] = sum_k W[n][k] X[m][k]:
// the token sits on the MFMA lane, the X fragments of the whole row (K <= 512) stay in registers for the entire
// kernel (LayerNorm applied there), and the weight matrix is the only thing that streams - 32-KB steps
// ([128 n][128 k]) through a 3-slot LDS ring by LDS-DMA, two steps ahead, counted vmcnt + one raw barrier per
// step, source-swizzled so the fragment ds_read_b128 are conflict-free.  Output columns are produced in chunks of
// 128; a chunk's epilogue runs while the next chunks' weights are already in flight: each wave transposes its
// O^T accumulators tile by tile through a private fp32 LDS tile and emits contiguous row segments (bias, r1, r2, bf16 stores).
// Residual operands (N <= 256) are fetched once up front in that row layout, so the loop has no ordinary global load.
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct LinArgs {
    const bf16_t* x; const float* ln_g; const float* ln_b; const bf16_t* w; const float* bias;
    const bf16_t* r1; const bf16_t* r2; bf16_t* out;
    int M, N, ldx, ldr, ldo; float eps;
    int mH, mW, mC;              // PatchMerging gather: x is NHWC [B, mH, mW, mC] and row m = the 2x2 neighbourhood (K = 4 mC); 0 = off
};

// up to three independent problems of one shape class (same M, K; N per problem) in ONE launch: blockIdx.y picks the problem
// (r03: the K | V_scale | V_shift projections that feed the sigma/mu attention, and q|k + v of the StyleEncoder's Scale / Shift
// call - each a 13-us one-tile-per-CU launch on its own)
constexpr int MAXG = 3;
struct LinGroup { LinArgs g[MAXG]; };

union Frag { u32x4 u; bf16x8_t b; };

constexpr int SLOT = 32768, RING = 3, AHEAD = 2;
constexpr int STAGE = 32768;     // epilogue staging: 8 waves x [32 tokens][32 channels] fp32

#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#ifdef LIN_CLOCK_PROBE   // diagnostic build only (profiles/probes/linear_probe.hip): s_memtime sections of wave 0
#define LPROBE_MARK() do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); probe_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define LPROBE_ADD(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); probe_sec[i] += t_ - probe_last; probe_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define LPROBE_MARK() do {} while (0)
#define LPROBE_ADD(i) do {} while (0)
#endif
#define LDS_READ128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define LGKM_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// NW = 8 waves (K <= 256), or 4 (K = 512: the X row alone is 128 registers, which leaves no room for two waves per SIMD)
template <bool GROUP> struct KArgs { typedef LinArgs type; };
template <> struct KArgs<true> { typedef LinGroup type; };
__device__ __forceinline__ const LinArgs& pick(const LinArgs& a) { return a; }
__device__ __forceinline__ const LinArgs& pick(const LinGroup& a) { return a.g[blockIdx.y]; }

// GROUP: the kernel argument is a LinGroup and blockIdx.y picks the problem (kept out of the single-problem instantiations: indexing
// the argument block costs the residual variants enough registers to spill)
template <int K, bool RES, int NW, bool GROUP>
__global__ __launch_bounds__(NW * 64) void linear_ts_bf16_kernel(typename KArgs<GROUP>::type karg) {
    const LinArgs& p = pick(karg);
#ifdef LIN_CLOCK_PROBE
    unsigned long long probe_sec[6] = {0, 0, 0, 0, 0, 0}, probe_last = 0;
    const unsigned long long probe_t0 = __builtin_amdgcn_s_memtime();
    LPROBE_MARK();
#endif
    constexpr int NTHREADS = NW * 64;
    constexpr int DMA_PER_STEP = 32 / NW;       // one-KB LDS-DMA pieces per wave per 32-KB step
    constexpr int TPW = 16 / NW;                // 32-column tiles per wave per 128-column chunk (4 or 2)
    constexpr int KS = K / 16;                  // k-steps over the whole row
    constexpr int KB = K / 128;                 // 32-KB weight steps per 128-column chunk

    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // [RING][SLOT] weights | stage | bias[N] | ln_g[K] | ln_b[K]
    float* bs = (float*)(lds + RING * SLOT + STAGE);
    float* lng = bs + p.N;
    float* lnb = lng + K;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int band = wave & 3, nh = wave >> 2;      // token band, and which half of every 128-column chunk this wave computes
    const int r = lane & 31, h = lane >> 5;
    const int m = blockIdx.x * 128 + band * 32 + r;
    const int mc = m < p.M ? m : p.M - 1;
    const int nch = p.N / 128;
    const int T = nch * KB;

    const int row0 = wave * (DMA_PER_STEP * 4) + (lane >> 4), l15 = lane & 15;
    auto issue = [&](int t) __attribute__((always_inline)) {
        const int nc = t / KB, kb = t - nc * KB;
        unsigned char* slot = lds + (t % RING) * SLOT + wave * DMA_PER_STEP * 1024;
        const bf16_t* base = p.w + (size_t)nc * 128 * K + kb * 128;               // wave-uniform
#pragma unroll
        for (int i = 0; i < DMA_PER_STEP; ++i) {
            const int row = row0 + 4 * i;                                          // piece = 4 rows x 256 B, source-swizzled
            const int off = row * K + ((l15 ^ (row & 15)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                             (__attribute__((address_space(3))) void*)(slot + i * 1024), 16, 0, 0);
        }
    };
    // K <= 256 (r03): the X tile itself comes in by LDS-DMA - whole rows per piece, each row once per workgroup - into ring slots 1 (and 2),
    // which the weight stream does not need yet, and the fragments are read back from LDS.  The r02 form loaded the fragments straight
    // from global memory: 16 instructions per wave, each touching 32 rows (lane = token), and both waves of a band loaded the same rows:
    // the prologue was 43 % of the kernel (profiles/r03_linear_probe_mi355x.txt).
    const bool XDMA = K <= 256 && p.mC == 0;
    int t_issue = 0;
    for (; t_issue < (XDMA ? 1 : AHEAD) && t_issue < T; ++t_issue) issue(t_issue);
    if (XDMA) {
        constexpr int RPP = 1024 / (K * 2), XPW = (128 / RPP) / NW;     // rows per 1-KB piece, pieces per wave
        unsigned char* xs = lds + SLOT;
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
            const int I = wave * XPW + i;
            const int row = I * RPP + lane / (64 / RPP), pos = lane % (64 / RPP);
            int mm = blockIdx.x * 128 + row; mm = mm < p.M ? mm : p.M - 1;
            const bf16_t* src = p.x + (size_t)mm * p.ldx + ((pos ^ (row & 15)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(xs + I * 1024), 16, 0, 0);
        }
    }

    for (int i = tid; i < p.N; i += NTHREADS) bs[i] = p.bias ? p.bias[i] : 0.f;
    if (p.ln_g) for (int i = tid; i < K; i += NTHREADS) { lng[i] = p.ln_g[i]; lnb[i] = p.ln_b[i]; }

    // residual operands, fetched once in the layout of the epilogue's row pass: iteration (nc, jt, it) handles rows
    // it*8 + (lane>>3) of this wave's 32 tokens, channels nc*128 + nh*64 + jt*32 + (lane&7)*4 .. +3  (64-B row segments)
    const int m0w = blockIdx.x * 128 + band * 32;
    constexpr int NRQ = RES ? 8 * TPW : 1;
    u32x2 rq1[NRQ], rq2[NRQ];
    if (RES) {
#pragma unroll
        for (int j = 0; j < 8 * TPW; ++j) {
            const int nc = j / (4 * TPW), jt = (j >> 2) % TPW, it = j & 3;
            const int mm = m0w + it * 8 + (lane >> 3);
            const int n = nc * 128 + nh * (TPW * 32) + jt * 32 + (lane & 7) * 4;
            rq1[j] = u32x2{0u, 0u}; rq2[j] = u32x2{0u, 0u};
            if (n < p.N && mm < p.M) {
                if (p.r1) rq1[j] = *(const u32x2*)(p.r1 + (size_t)mm * p.ldr + n);
                if (p.r2) rq2[j] = *(const u32x2*)(p.r2 + (size_t)mm * p.ldr + n);
            }
        }
    }

    Frag xf[KS];
    {
        if (p.mC) {
            // torchvision PatchMerging: cat(x[0::2,0::2], x[1::2,0::2], x[0::2,1::2], x[1::2,1::2]) -> segment q is pixel (dy = q&1, dx = q>>1)
            const int W2 = p.mW >> 1, H2 = p.mH >> 1;
            const int b = mc / (H2 * W2), rem = mc - b * (H2 * W2), y2 = rem / W2, x2 = rem - y2 * W2;
            const bf16_t* px = p.x + (((size_t)b * p.mH + 2 * y2) * p.mW + 2 * x2) * p.mC;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int k0 = 16 * s + 8 * h, q = k0 / p.mC, c = k0 - q * p.mC;
                xf[s].u = *(const u32x4*)(px + ((size_t)(q & 1) * p.mW + (q >> 1)) * p.mC + c);
            }
        } else {
            if (XDMA) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's X pieces (and weight step 0, residual values) have landed
                __builtin_amdgcn_s_barrier();
                const int row = band * 32 + r;
                const unsigned char* xr = lds + SLOT + row * (K * 2);
#pragma unroll
                for (int s = 0; s < KS; ++s) xf[s].u = *(const u32x4*)(xr + (((2 * s + h) ^ (row & 15)) << 4));
            } else {
                const bf16_t* xp = p.x + (size_t)mc * p.ldx + 8 * h;
#pragma unroll
                for (int s = 0; s < KS; ++s) xf[s].u = *(const u32x4*)(xp + 16 * s);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                        // parameter vectors visible; with XDMA: every wave has its X fragments, slots 1.. are free
        if (XDMA) for (; t_issue < AHEAD && t_issue < T; ++t_issue) issue(t_issue);
        if (p.ln_g) {
            float sum = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    sum += __uint_as_float(xf[s].u[e] << 16) + __uint_as_float(xf[s].u[e] & 0xffff0000u);
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum / (float)K;
            float sq = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d0 = __uint_as_float(xf[s].u[e] << 16) - mean, d1 = __uint_as_float(xf[s].u[e] & 0xffff0000u) - mean;
                    sq += d0 * d0 + d1 * d1;
                }
            sq += __shfl_xor(sq, 32, 64);
            const float rstd = rsqrtf(sq / (float)K + p.eps);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
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
    }

    LPROBE_ADD(0);                                                       // prologue: first DMA issue, parameter vectors, residual + X loads, LayerNorm
    int t_cons = 0;
    auto begin_step = [&]() __attribute__((always_inline)) {
        // conservative count: stores of the previous chunk's epilogue are younger than every DMA, so "at most 4
        // outstanding" still implies the DMAs of step t_cons have landed
        const int ahead = t_issue - 1 - t_cons;
        if (ahead >= 1) { if (DMA_PER_STEP == 8) WAIT_VMCNT(8); else WAIT_VMCNT(4); } else WAIT_VMCNT(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        LPROBE_ADD(1);                                                   // wait for the step's weights + barrier
        if (t_issue < T) { issue(t_issue); ++t_issue; }
        LPROBE_ADD(2);                                                   // DMA issue of a later step
        const unsigned char* slot = lds + (t_cons % RING) * SLOT;
        ++t_cons;
        return slot;
    };
    auto lds_off = [&](const unsigned char* ptr) { return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)ptr; };
    const int x256 = (h ^ (r & 15)) << 4;

    f32x16 acc[TPW];
    auto zero_acc = [&]() {
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    };
    auto step = [&](const unsigned char* slot, int kb) __attribute__((always_inline)) {
        const unsigned base = lds_off(slot) + nh * 16384 + r * 256;             // this wave's 32-row tiles of the 128-row step
        u32x4 a[2][TPW];
        auto rd = [&](int buf, int s) {
            const unsigned ad = base + ((32 * s) ^ x256);
            LDS_READ128(a[buf][0], ad, 0); LDS_READ128(a[buf][1], ad, 8192);
            if (TPW == 4) { LDS_READ128(a[buf][TPW - 2], ad, 16384); LDS_READ128(a[buf][TPW - 1], ad, 24576); }
        };
        rd(0, 0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s + 1 < 8) { rd((s + 1) & 1, s + 1); if (TPW == 4) LGKM_WAIT(4); else LGKM_WAIT(2); } else { LGKM_WAIT(0); }
#pragma unroll
            for (int jt = 0; jt < TPW; ++jt) {
                Frag f; f.u = a[s & 1][jt];
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b, xf[kb * 8 + s].b, acc[jt], 0, 0, 0);
            }
        }
    };
    // epilogue of this wave's output columns [nc*128 + nh*32*TPW, +32*TPW).  The accumulators hold O^T (lane = token, registers =
    // channels); storing that directly means 8-byte stores strided by the row pitch.  Each wave instead transposes one tile
    // (32 channels) at a time through its private fp32 staging tile and emits 64-B row segments.
    auto epilogue = [&](int nc) __attribute__((always_inline)) {
        float* stage = (float*)(lds + RING * SLOT + wave * 4096);              // [32 tokens][32 ch] fp32, 16-B chunks swizzled by token&7
#pragma unroll
        for (int jt = 0; jt < TPW; ++jt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 2 * g + h;                                        // 16-B chunk of channels 8g + 4h .. +3 within the tile
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[jt][4 * g + e];
                *(f32x4*)(stage + r * 32 + ((c ^ (r & 7)) << 2)) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int n0 = nc * 128 + nh * (TPW * 32) + jt * 32 + (lane & 7) * 4;
            const f32x4 bv = *(const f32x4*)(bs + n0);
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3), mm = m0w + row;
                f32x4 v = *(const f32x4*)(stage + row * 32 + (((lane & 7) ^ (row & 7)) << 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bv[e];
                if (RES) {
                    const u32x2 a1 = nc == 0 ? rq1[(jt * 4 + it) % NRQ] : rq1[(4 * TPW + jt * 4 + it) % NRQ];
                    const u32x2 a2 = nc == 0 ? rq2[(jt * 4 + it) % NRQ] : rq2[(4 * TPW + jt * 4 + it) % NRQ];
                    if (p.r1) {
                        v[0] *= __uint_as_float(a1[0] << 16); v[1] *= __uint_as_float(a1[0] & 0xffff0000u);
                        v[2] *= __uint_as_float(a1[1] << 16); v[3] *= __uint_as_float(a1[1] & 0xffff0000u);
                    }
                    if (p.r2) {
                        v[0] += __uint_as_float(a2[0] << 16); v[1] += __uint_as_float(a2[0] & 0xffff0000u);
                        v[2] += __uint_as_float(a2[1] << 16); v[3] += __uint_as_float(a2[1] & 0xffff0000u);
                    }
                }
                u32x2 o; o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]);
                if (mm < p.M) *(u32x2*)(p.out + (size_t)mm * p.ldo + n0) = o;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // tile reads done before the next tile overwrites it
        }
    };

    for (int nc = 0; nc < nch; ++nc) {
        zero_acc();
        LPROBE_MARK();
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) { step(begin_step(), kb); LPROBE_ADD(3); }
        epilogue(nc);
        LPROBE_ADD(4);
    }
#ifdef LIN_CLOCK_PROBE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && g_probe) {
        g_probe[blockIdx.x * 8] = __builtin_amdgcn_s_memtime() - probe_t0;
        for (int i = 0; i < 5; ++i) g_probe[blockIdx.x * 8 + 1 + i] = probe_sec[i];
    }
#endif
}

template <int K, bool RES>
int launch(const LinArgs& p, hipStream_t s) {
    constexpr int NW = K <= 256 ? 8 : 4;
    const size_t lds = (size_t)RING * SLOT + STAGE + (size_t)(p.N + 2 * K) * sizeof(float);
    static LdsLimit lds_limit;
    lds_limit.ensure((const void*)linear_ts_bf16_kernel<K, RES, NW, false>, (int)((int)lds));
    hipLaunchKernelGGL((linear_ts_bf16_kernel<K, RES, NW, false>), dim3((p.M + 127) / 128), dim3(NW * 64), lds, s, p);
    return master_check_launch("linear_ts_bf16");
}

template <int K>
int launch_group(const LinGroup& grp, int n, hipStream_t s) {
    constexpr int NW = 8;
    int nmax = 0;
    for (int i = 0; i < n; ++i) nmax = grp.g[i].N > nmax ? grp.g[i].N : nmax;
    const size_t lds = (size_t)RING * SLOT + STAGE + (size_t)(nmax + 2 * K) * sizeof(float);
    static LdsLimit lds_limit;
    lds_limit.ensure((const void*)linear_ts_bf16_kernel<K, false, NW, true>, (int)((int)lds));
    hipLaunchKernelGGL((linear_ts_bf16_kernel<K, false, NW, true>), dim3((grp.g[0].M + 127) / 128, n), dim3(NW * 64), lds, s, grp);
    return master_check_launch("linear_ts_bf16 (group)");
}

}  // namespace

// returns MASTER_ERR_UNSUPPORTED when the shape is outside what this kernel covers (the caller then uses the tiled GEMM)
int linear_ts_bf16(const bf16_t* x, const float* ln_g, const float* ln_b, const bf16_t* w, const float* bias,
                   const bf16_t* r1, const bf16_t* r2, bf16_t* out, int M, int N, int K, int ldx, int ldr, int ldo, float eps,
                   hipStream_t s, int mH = 0, int mW = 0, int mC = 0) {
    const bool res = r1 || r2;
    if (N % 128 != 0 || N > 1024 || (res && N > 256) || (K != 128 && K != 256 && K != 512)) return MASTER_ERR_UNSUPPORTED;
    if (mC && (4 * mC != K || mC % 8 != 0)) return MASTER_ERR_UNSUPPORTED;
    LinArgs p{x, ln_g, ln_b, w, bias, r1, r2, out, M, N, ldx, ldr, ldo, eps, mH, mW, mC};
    if (K == 128) return res ? launch<128, true>(p, s) : launch<128, false>(p, s);
    if (K == 256) return res ? launch<256, true>(p, s) : launch<256, false>(p, s);
    return res ? launch<512, true>(p, s) : launch<512, false>(p, s);
}

// n (<= 3) problems out_i[M, N_i] = LN?(x_i) . W_i^T + bias_i in one launch; same M and K, no residual operands.
// MASTER_ERR_UNSUPPORTED outside the token-stationary kernel's shapes (the caller then launches them one by one).
int linear_ts_group_bf16(int n, const bf16_t* const* x, const bf16_t* const* w, const float* const* bias, bf16_t* const* out, const int* N,
                         const int* ldx, const int* ldo, const float* ln_g, const float* ln_b, int M, int K, float eps, hipStream_t s) {
    if (n < 1 || n > MAXG || (K != 128 && K != 256)) return MASTER_ERR_UNSUPPORTED;
    LinGroup grp{};
    for (int i = 0; i < n; ++i) {
        if (N[i] % 128 != 0 || N[i] > 1024) return MASTER_ERR_UNSUPPORTED;
        grp.g[i] = LinArgs{x[i], ln_g, ln_b, w[i], bias[i], nullptr, nullptr, out[i], M, N[i], ldx[i], 0, ldo[i], eps, 0, 0, 0};
    }
    return K == 128 ? launch_group<128>(grp, n, s) : launch_group<256>(grp, n, s);
}
