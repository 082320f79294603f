// Issue-rate microbenchmarks on gfx950, one wave per SIMD (256 threads, 1 workgroup); loop bodies are inline asm so the
// instruction order is exactly as written.  Prints s_memtime ticks per loop iteration (4 independent MFMAs = 128 ticks, so a
// tick is one matrix-core clock).  Questions: does a wave's own VALU work overlap its MFMAs?  What do dependent vs independent
// v_pk_fma_f32 chains cost?  What do transcendental ops cost?
// build+run:  hipcc -O3 --offload-arch=gfx950 -o /tmp/issue_probe profiles/probes/issue_probe.hip && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
#define NIT 4000

#define MFMA(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(fa), "v"(fb))
#define FMA(X) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(t0), "v"(t1))
#define PK(X) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(p0), "v"(p1))
#define EXP(X) asm volatile("v_exp_f32 %0, %0" : "+v"(X))
#define RCP(X) asm volatile("v_rcp_f32 %0, %0" : "+v"(X))

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float seed) {
    f2 a = {seed + threadIdx.x, seed * 2}, b = {seed * 3, seed + 1}, c = {seed, seed}, d = {seed + 5, seed}, p0 = {0.999f, 1.001f}, p1 = {0.5f, 0.25f};
    float s0 = seed + threadIdx.x, s1 = seed * 2, s2 = seed * 3, s3 = seed * 4, s4 = seed * 5, s5 = seed * 6, s6 = seed * 7, s7 = seed * 8, t0 = 0.999f, t1 = 0.001f;
    f16v acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    f4v fa = {seed, seed, seed, seed}, fb = {seed, -seed, seed, seed};
    unsigned long long ts = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < NIT; ++it) {
        if (MODE == 0) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); }
        if (MODE == 1) { FMA(s0); FMA(s1); FMA(s2); FMA(s3); FMA(s4); FMA(s5); FMA(s6); FMA(s7); FMA(s0); FMA(s1); FMA(s2); FMA(s3); FMA(s4); FMA(s5); FMA(s6); FMA(s7); }
        if (MODE == 2) { MFMA(acc0); FMA(s0); FMA(s1); FMA(s2); FMA(s3); MFMA(acc1); FMA(s4); FMA(s5); FMA(s6); FMA(s7); MFMA(acc2); FMA(s0); FMA(s1); FMA(s2); FMA(s3); MFMA(acc3); FMA(s4); FMA(s5); FMA(s6); FMA(s7); }
        if (MODE == 3) { MFMA(acc0); MFMA(acc1); MFMA(acc2); MFMA(acc3); FMA(s0); FMA(s1); FMA(s2); FMA(s3); FMA(s4); FMA(s5); FMA(s6); FMA(s7); FMA(s0); FMA(s1); FMA(s2); FMA(s3); FMA(s4); FMA(s5); FMA(s6); FMA(s7); }
        if (MODE == 4) { MFMA(acc0); FMA(s0); FMA(s1); FMA(s2); FMA(s3); FMA(s4); FMA(s5); FMA(s6); MFMA(acc1); FMA(s7); FMA(s0); FMA(s1); FMA(s2); FMA(s3); FMA(s4); FMA(s5); MFMA(acc2); FMA(s6); FMA(s7); FMA(s0); FMA(s1); FMA(s2); FMA(s3); FMA(s4); MFMA(acc3); FMA(s5); FMA(s6); FMA(s7); FMA(s0); FMA(s1); FMA(s2); FMA(s3); }
        if (MODE == 5) { PK(a); PK(a); PK(a); PK(a); PK(a); PK(a); PK(a); PK(a); }
        if (MODE == 6) { PK(a); PK(b); PK(a); PK(b); PK(a); PK(b); PK(a); PK(b); }
        if (MODE == 7) { PK(a); PK(b); PK(c); PK(d); PK(a); PK(b); PK(c); PK(d); }
        if (MODE == 8) { MFMA(acc0); PK(a); PK(b); PK(c); PK(d); MFMA(acc1); PK(a); PK(b); PK(c); PK(d); MFMA(acc2); PK(a); PK(b); PK(c); PK(d); MFMA(acc3); PK(a); PK(b); PK(c); PK(d); }
        if (MODE == 9) { MFMA(acc0); PK(a); PK(b); PK(c); PK(d); PK(a); PK(b); PK(c); MFMA(acc1); PK(d); PK(a); PK(b); PK(c); PK(d); PK(a); PK(b); MFMA(acc2); PK(c); PK(d); PK(a); PK(b); PK(c); PK(d); PK(a); MFMA(acc3); PK(b); PK(c); PK(d); PK(a); PK(b); PK(c); PK(d); }
        if (MODE == 10) { EXP(s0); EXP(s1); EXP(s2); EXP(s3); EXP(s4); EXP(s5); EXP(s6); EXP(s7); }
        if (MODE == 11) { EXP(s0); EXP(s0); EXP(s0); EXP(s0); EXP(s0); EXP(s0); EXP(s0); EXP(s0); }
        if (MODE == 12) { RCP(s0); RCP(s1); RCP(s2); RCP(s3); RCP(s4); RCP(s5); RCP(s6); RCP(s7); }
        if (MODE == 13) { FMA(s0); FMA(s0); FMA(s0); FMA(s0); FMA(s0); FMA(s0); FMA(s0); FMA(s0); }
        if (MODE == 14) { MFMA(acc0); EXP(s0); EXP(s1); EXP(s2); MFMA(acc1); EXP(s3); EXP(s4); EXP(s5); MFMA(acc2); EXP(s6); EXP(s7); EXP(s0); MFMA(acc3); EXP(s1); EXP(s2); EXP(s3); }
    }
    unsigned long long te = __builtin_amdgcn_s_memtime();
    float r = a[0] + a[1] + b[0] + b[1] + c[0] + c[1] + d[0] + d[1] + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
    for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + acc2[i] + acc3[i];
    out[threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[0] = te - ts;
}

template <int MODE> void run(const char* what, float* out, unsigned long long* cyc) {
    unsigned long long h = 0;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), 0, 0, out, cyc, 0.001f); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); }
    printf("%-64s %8.1f ticks / iteration\n", what, (double)h / NIT);
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 1024); hipMalloc(&cyc, 8);
    run<0>("4 MFMA 32x32x16 bf16, independent", out, cyc);
    run<1>("16 v_fma_f32, 8 chains", out, cyc);
    run<13>("8 v_fma_f32, 1 dependent chain", out, cyc);
    run<2>("4 x (MFMA ; 4 v_fma)", out, cyc);
    run<3>("4 MFMA ; then 16 v_fma", out, cyc);
    run<4>("4 x (MFMA ; 7 v_fma)", out, cyc);
    run<5>("8 v_pk_fma_f32, 1 dependent chain", out, cyc);
    run<6>("8 v_pk_fma_f32, 2 chains", out, cyc);
    run<7>("8 v_pk_fma_f32, 4 chains", out, cyc);
    run<8>("4 x (MFMA ; 4 pk_fma)", out, cyc);
    run<9>("4 x (MFMA ; 7 pk_fma)", out, cyc);
    run<10>("8 v_exp_f32, independent", out, cyc);
    run<11>("8 v_exp_f32, dependent", out, cyc);
    run<12>("8 v_rcp_f32, independent", out, cyc);
    run<14>("4 x (MFMA ; 3 v_exp)", out, cyc);
    return 0;
}
