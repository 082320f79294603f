// Shared device helpers for libmaster_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/master_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));   // 8 bf16 bit patterns = one MFMA A/B fragment
typedef unsigned short bf16_t;                               // raw bf16 bits

typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16x2 (round-to-nearest-even, NaN preserved): one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    bf16x2_hw v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}
// two fp32 values -> packed bf16 hi pair + packed bf16 lo pair (hi = bf16(x), lo = bf16(x - hi), round to nearest even both times): the split
// of the F32X3 mode.  The packed hi word is made opaque to the optimiser: left visible, hipcc computes float(hi0) from a SECOND
// v_cvt_pk_bf16_f32 of x0 alone instead of shifting the packed word - 3 conversions per pair instead of 2 in kernels that are bound by
// their vector-instruction count (r04: 240 -> 160 conversions per problem in the attention kernel).
__device__ __forceinline__ void split_bf16x2(float x0, float x1, unsigned& hi, unsigned& lo) {
    unsigned a = pack_bf16x2(x0, x1);
    asm("" : "+v"(a));
    hi = a;
    lo = pack_bf16x2(x0 - __uint_as_float(a << 16), x1 - __uint_as_float(a & 0xffff0000u));
}
// The same split with IEEE-half halves (MASTER_F16X3, r05): hi = half(x), lo = half(x - hi), round to nearest even (v_cvt_pk_f16_f32).  11
// significant bits per half instead of 8: hi + lo reproduces x to 2^-22 |x| (bf16 halves: 2^-17) down to the subnormal floor 2^-25 - the
// matrix core takes subnormal half operands at full precision (profiles/r05_f16_split_probe_mi355x.txt).  What it gives up is bf16's
// exponent range: |x| > 65 504 becomes inf (as under fp16 autocast), so the frozen loss network and its gradients stay on bf16 halves.
typedef _Float16 f16x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {
    f16x2_hw v; v[0] = (_Float16)lo; v[1] = (_Float16)hi;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split_f16x2(float x0, float x1, unsigned& hi, unsigned& lo) {
    unsigned a = pack_f16x2(x0, x1);
    asm("" : "+v"(a));
    hi = a;
#if defined(MASTER_SPLIT_NO_FMA_MIX)   // A/B builds: the remainder through v_cvt_f32_f16 + v_sub_f32 + v_cvt_pk (6 vector instructions per pair)
    const f16x2_hw v = __builtin_bit_cast(f16x2_hw, a);
    lo = pack_f16x2(x0 - (float)v[0], x1 - (float)v[1]);
#elif defined(MASTER_SPLIT_FMA_MIX_F32)  // A/B builds: x - float(hi) by v_fma_mix_f32, then v_cvt_pk (4 per pair; measured equal to the shipped form)
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(a), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(a), "v"(x1));
    lo = pack_f16x2(l0, l1);
#else
    // lo = half(x - float(hi)) in ONE instruction per value, THREE per pair: v_fma_mixlo_f16 / v_fma_mixhi_f16 read the half straight out of
    // the packed word (op_sel picks the half, op_sel_hi marks source 0 as f16), compute hi * -1.0 + x in fp32 (exact: x - hi is representable)
    // and round the result to half (nearest even) into the low / high half of the destination - bit-identical to the long form
    // (tests/test_gpu_f16x3.py::test_split_weights_half_layout_bit_exact, subnormal remainders included)
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(a), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(a), "v"(x1));
    lo = l;
#endif
}
// operand format of the split kernels: X3_BF16 (MASTER_F32X3) or X3_F16 (MASTER_F16X3); fragments travel as four 32-bit words either way
constexpr int X3_BF16 = 0, X3_F16 = 1;
typedef unsigned int x3_frag __attribute__((ext_vector_type(4)));
template <int FMT> __device__ __forceinline__ void split_x2(float x0, float x1, unsigned& hi, unsigned& lo) {
    if constexpr (FMT == X3_F16) split_f16x2(x0, x1, hi, lo); else split_bf16x2(x0, x1, hi, lo);
}
template <int FMT> __device__ __forceinline__ f32x16 mfma_x3_32(const x3_frag& a, const x3_frag& b, const f32x16& c) {
    typedef __bf16 b8 __attribute__((ext_vector_type(8)));
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    if constexpr (FMT == X3_F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
}
template <int FMT> __device__ __forceinline__ f32x4 mfma_x3_16(const x3_frag& a, const x3_frag& b, const f32x4& c) {
    typedef __bf16 b8 __attribute__((ext_vector_type(8)));
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    if constexpr (FMT == X3_F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a), __builtin_bit_cast(b8, b), c, 0, 0, 0);
}
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.0f) & 0xffffu); }

// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7): one v_rcp, one v_exp, six FMAs.  libm's erff costs
// ~40 VALU instructions per value, which made the GELU epilogue of the MLP's first GEMM VALU-bound.
__device__ __forceinline__ float fast_erf(float z) {
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-az * az * 1.4426950408889634f);
    return copysignf(fmaf(-p * t, e, 1.0f), z);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f)); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
// The same GELU in packed math (r05 experiment, used by mlp_x3.hip under -DMLPX3_GELU_7128 only), two values per instruction: erf by Abramowitz & Stegun 7.1.28,
// erf(z) = 1 - 1 / (1 + a1 z + ... + a6 z^6)^16 (|error| <= 3e-7), z = |x| / sqrt(2) folded into the coefficients: six v_pk_fma, four
// v_pk_mul (the 16th power), ONE transcendental per value (v_rcp; 7.1.26 above needs v_rcp and v_exp, and transcendentals issue at quarter
// rate) - 12.5 issue slots per value instead of 21.  |GELU error| <= 8.8e-7 over |x| <= 12 in fp32 arithmetic (7.1.26: 4.6e-7), i.e. 2 ulp at
// |x| = 4; large |x|: the power overflows to inf, 1 / inf = 0, erf = 1 exactly.  In two halves so that the MLP can deal them out between MFMA groups.
__device__ __forceinline__ f32x2 gelu_erf2_pow(f32x2 x) {                 // first half: (1 + a1 z + ... + a6 z^6)^16
    f32x2 a;
    a.x = fabsf(x.x); a.y = fabsf(x.y);
    f32x2 P = a * 5.382975e-06f + 4.8890636e-05f;                             // a6 / 8, a5 / (4 sqrt 2)  (z = |x| / sqrt 2)
    P = P * a + 3.8003575e-05f;                                           // a4 / 4
    P = P * a + 3.2776263e-03f;                                           // a3 / (2 sqrt 2)
    P = P * a + 2.11410062e-02f;                                          // a2 / 2
    P = P * a + 4.98673469e-02f;                                          // a1 / sqrt 2
    P = P * a + 1.0f;
    P = P * P; P = P * P; P = P * P;
    return P * P;
}
__device__ __forceinline__ f32x2 gelu_erf2_finish(f32x2 x, f32x2 P16) {   // second half: x * Phi(x), Phi = 1/2 + copysign(1/2 - r/2, x)
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(P16.x); r.y = __builtin_amdgcn_rcpf(P16.y);
    const f32x2 h = r * -0.5f + 0.5f;
    f32x2 sg;
    sg.x = copysignf(h.x, x.x); sg.y = copysignf(h.y, x.y);
    return x * sg + x * 0.5f;
}

// GELU for the bf16 mode: x * clamp01(0.5 + x R(x^2)), R fitted to (Phi(x) - 0.5)/x; beyond the fitted range the polynomial runs off in
// the right direction (even degree of R, positive leading coefficient) and the clamp pins Phi to 0 or 1, +-inf included.
//   degree 4 (default, r03): least-squares fit of the GELU error under a N(0, 2) weight on |x| <= 3.2 (where pre-activations live):
//     rms error 2.7e-4 for N(0,1) inputs, 3.7e-4 for N(0,1.5), |error| <= 2.3e-3 (reached near |x| = 3) - the hidden activations are
//     rounded to bf16 (relative 2e-3) right after.  7 VALU instructions per value.
//   degree 6 (-DMASTER_GELU_DEG6, r01/r02): |error| <= 1.9e-4.  9 instructions per value.
// The GELU arithmetic is the exposed VALU time of the fused MLP kernels (18.7 k of 111 k clocks per workgroup with degree 6,
// profiles/r02_mlp_probe_*.txt); exp / rcp forms cost ~29 op slots (transcendentals issue at quarter rate).  The fp32 parity mode keeps gelu_erf.
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) {                   // two values per instruction (v_pk_mul / v_pk_fma)
    const f32x2 t = x * x;
#ifdef MASTER_GELU_DEG6
    f32x2 R = t * 2.278150468e-08f + -1.598605536e-06f;
    R = R * t + 4.795558198e-05f; R = R * t + -8.140160353e-04f; R = R * t + 8.772390895e-03f;
    R = R * t + -6.457311660e-02f; R = R * t + 3.978833556e-01f;
#else
    f32x2 R = t * 1.579773891e-05f + -5.356991246e-04f;
    R = R * t + 7.651804271e-03f; R = R * t + -6.262873859e-02f; R = R * t + 3.968234232e-01f;
#endif
    f32x2 p;
    asm("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(p) : "v"(x), "v"(R));
    return x * p;
}
__device__ __forceinline__ float gelu_poly1(float xx) {                  // one value (the fine-grained form of mlp_bf16.hip)
    const float t = xx * xx;
#ifdef MASTER_GELU_DEG6
    float R = fmaf(t, 2.278150468e-08f, -1.598605536e-06f);
    R = fmaf(R, t, 4.795558198e-05f); R = fmaf(R, t, -8.140160353e-04f); R = fmaf(R, t, 8.772390895e-03f);
    R = fmaf(R, t, -6.457311660e-02f); R = fmaf(R, t, 3.978833556e-01f);
#else
    float R = fmaf(t, 1.579773891e-05f, -5.356991246e-04f);
    R = fmaf(R, t, 7.651804271e-03f); R = fmaf(R, t, -6.262873859e-02f); R = fmaf(R, t, 3.968234232e-01f);
#endif
    return xx * __builtin_amdgcn_fmed3f(fmaf(xx, R, 0.5f), 0.0f, 1.0f);
}

// ReLU that propagates NaN like torch.relu (v_max_f32 returns the other operand for a NaN: fmaxf(NaN, 0) = 0).  The exact-fp32 and the split
// modes use it: in "f16x3" an activation beyond half's range becomes inf / NaN, and the decoder's ReLUs must not turn that into a clean 0
// (r05: a k = 5 forward on random weights, activations up to 4.5e7, came out FINITE and wrong before this).
__device__ __forceinline__ float relu_np(float x) { return x < 0.0f ? 0.0f : x; }

template <int ACT>
__device__ __forceinline__ float apply_act(float x) {
    if (ACT == MASTER_ACT_GELU) return gelu_erf(x);
    if (ACT == MASTER_ACT_RELU) return relu_np(x);
    return x;
}

// LayerNorm of a token row held as bf16 fragments in registers (fragment s of lane (token, half h) = channels 16 s + 8 h .. +7; lane ^ 32 holds
// the other half of the row), in packed fp32 math (r05: v_pk_add / v_pk_fma, two channels per instruction, normalisation folded to
// v * rstd + (-mean * rstd)): gamma / beta are fp32 vectors in LDS.  Used by the fused MLP kernels; attn_block_bf16.hip has its own copy.
template <int KS, typename F>
__device__ __forceinline__ void layernorm_frags_pk(F (&xf)[KS], const float* lng, const float* lnb, int h, float eps) {
    constexpr int CH = KS * 16;
    f32x2 sum2 = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) sum2 += f32x2{__uint_as_float(xf[s].u[e] << 16), __uint_as_float(xf[s].u[e] & 0xffff0000u)};
    float sum = sum2.x + sum2.y;
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / CH);
    f32x2 sq2 = {0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x2 d = f32x2{__uint_as_float(xf[s].u[e] << 16), __uint_as_float(xf[s].u[e] & 0xffff0000u)} - mean;
            sq2 += d * d;
        }
    float sq = sq2.x + sq2.y;
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = rsqrtf(sq * (1.0f / CH) + eps);
    const float nmr = -mean * rstd;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const float* gp = lng + 16 * s + 8 * h; const float* bp = lnb + 16 * s + 8 * h;
        const f32x4 g0 = *(const f32x4*)gp, g1 = *(const f32x4*)(gp + 4);
        const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const f32x2 v = f32x2{__uint_as_float(xf[s].u[e] << 16), __uint_as_float(xf[s].u[e] & 0xffff0000u)};
            const f32x2 g = e < 2 ? f32x2{g0[2 * e], g0[2 * e + 1]} : f32x2{g1[2 * e - 4], g1[2 * e - 3]};
            const f32x2 b = e < 2 ? f32x2{b0[2 * e], b0[2 * e + 1]} : f32x2{b1[2 * e - 4], b1[2 * e - 3]};
            const f32x2 o = (v * rstd + nmr) * g + b;
            xf[s].u[e] = pack_bf16x2(o.x, o.y);
        }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- host side: per-device "raise the dynamic-LDS limit" cache --------------------------------------------------
// hipFuncSetAttribute acts on the CURRENT device's copy of the kernel, so the cache is one atomic per device (a process may
// drive several GPUs from several threads): no lock, at worst two threads set the same value twice.
#include <atomic>
struct LdsLimit {
    static constexpr int MAX_DEV = 64;
    std::atomic<int> bytes[MAX_DEV];
    void ensure(const void* kernel, int need) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= MAX_DEV) { (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, need); return; }
        if (bytes[dev].load(std::memory_order_acquire) < need) {
            (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, need);
            int cur = bytes[dev].load(std::memory_order_relaxed);
            while (cur < need && !bytes[dev].compare_exchange_weak(cur, need, std::memory_order_release)) {}
        }
    }
};

// ---- host side: error string + profiling hooks (defined in capi.hip) -------------------------
void master_set_error(const char* fmt, ...);
struct ProfScope {            // brackets one launch with events when profiling is on
    int family; hipStream_t s; hipEvent_t e0, e1; bool on;
    ProfScope(int family, hipStream_t s);
    ~ProfScope();
};
int master_check_launch(const char* what);
