// Follow-up issue probes on gfx950: (a) MFMA accumulate chains of depth 1 / 2 / 4, (b) ONE vs TWO waves per SIMD running
// MFMA-only, VALU-only and "MFMA ; 4 VALU" streams (512 threads = two waves per SIMD).  Clocks per loop iteration of wave 0.
// build+run:  hipcc -O3 --offload-arch=gfx950 -o /tmp/issue_probe2 profiles/probes/issue_probe2.hip && /tmp/issue_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
#define NIT 3000
#define MFMA(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(fa), "v"(fb))
#define FMA(X) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(t0), "v"(t1))
#define EXP(X) asm volatile("v_exp_f32 %0, %0" : "+v"(X))

template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, float seed) {
    float s0 = seed + threadIdx.x, s1 = seed * 2, s2 = seed * 3, s3 = seed * 4, s4 = seed * 5, s5 = seed * 6, s6 = seed * 7, s7 = seed * 8, t0 = 0.999f, t1 = 0.001f;
    f16v acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    f4v fa = {seed, seed, seed, seed}, fb = {seed, -seed, seed, seed};
    const int wave = threadIdx.x >> 6;
    const bool second = wave >= 4;                       // the second wave on each SIMD (waves w and w+4 share a SIMD)
    __syncthreads();
    unsigned long long ts = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < NIT; ++it) {
        if (MODE == 0) { MFMA(acc0); MFMA(acc0); MFMA(acc0); MFMA(acc0); }                       // chain depth 1
        if (MODE == 1) { MFMA(acc0); MFMA(acc1); MFMA(acc0); MFMA(acc1); }                       // depth 2
        if (MODE == 2) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }                       // depth 4
        if (MODE == 3) { MFMA(acc0); FMA(s0); FMA(s1); FMA(s2); FMA(s3); MFMA(acc1); FMA(s4); FMA(s5); FMA(s6); FMA(s7);
                         MFMA(acc2); FMA(s0); FMA(s1); FMA(s2); FMA(s3); MFMA(acc3); FMA(s4); FMA(s5); FMA(s6); FMA(s7); }   // every wave
        if (MODE == 4) { if (!second) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
                         else { FMA(s0); FMA(s1); FMA(s2); FMA(s3); FMA(s4); FMA(s5); FMA(s6); FMA(s7); FMA(s0); FMA(s1); FMA(s2); FMA(s3); FMA(s4); FMA(s5); FMA(s6); FMA(s7); } }
        if (MODE == 5) { if (!second) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
                         else { EXP(s0); EXP(s1); EXP(s2); EXP(s3); EXP(s4); EXP(s5); EXP(s6); EXP(s7); } }
        if (MODE == 6) { FMA(s0); FMA(s1); FMA(s2); FMA(s3); FMA(s4); FMA(s5); FMA(s6); FMA(s7); FMA(s0); FMA(s1); FMA(s2); FMA(s3); FMA(s4); FMA(s5); FMA(s6); FMA(s7); }
    }
    unsigned long long te = __builtin_amdgcn_s_memtime();
    float r = s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[wave] = te - ts;
}
template <int MODE> void run(const char* what, int threads, float* out, unsigned long long* cyc) {
    unsigned long long h[8] = {0};
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, 0.001f); hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost); }
    printf("%-66s %3d thr: wave0 %7.1f", what, threads, (double)h[0] / NIT);
    if (threads == 512) printf("   wave4 %7.1f", (double)h[4] / NIT);
    printf("  clocks / iteration\n");
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
    run<0>("4 MFMA, one accumulator (dependent chain)", 256, out, cyc);
    run<1>("4 MFMA, two accumulators alternating", 256, out, cyc);
    run<2>("4 MFMA, four accumulators", 256, out, cyc);
    run<2>("4 MFMA, four accumulators; two waves per SIMD", 512, out, cyc);
    run<6>("16 v_fma; one wave per SIMD", 256, out, cyc);
    run<6>("16 v_fma; two waves per SIMD", 512, out, cyc);
    run<3>("4 x (MFMA ; 4 v_fma); one wave per SIMD", 256, out, cyc);
    run<3>("4 x (MFMA ; 4 v_fma); two waves per SIMD", 512, out, cyc);
    run<4>("wave A: 4 MFMA, wave B: 16 v_fma (same SIMD)", 512, out, cyc);
    run<5>("wave A: 4 MFMA, wave B: 8 v_exp (same SIMD)", 512, out, cyc);
    return 0;
}
