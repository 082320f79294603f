// Do a wave's ds_read_b128 fragment reads overlap its MFMAs?  4 waves (one per SIMD), each iteration = one "k-step group":
// 4 fragment reads for the next group, counted wait for this group's, 4 MFMAs.  Variants: reads only, MFMAs only, both;
// accumulators in VGPRs or AGPRs; conflict-free (swizzled 256-B rows) addresses as in mlp_bf16.hip.
// build+run:  hipcc -O3 --offload-arch=gfx950 -o /tmp/lds_mfma profiles/probes/lds_mfma_probe.hip && /tmp/lds_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define NIT 2000
#define RD(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define MFMA_V(ACC, A) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(fb))
#define MFMA_A(ACC, A) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(fb))
#define WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int stride) {
    extern __shared__ unsigned char lds[];
    for (int i = threadIdx.x; i < 32768 / 4; i += 256) ((unsigned*)lds)[i] = i * 2654435761u >> 20;
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + r * 256 + ((h ^ (r & 15)) << 4);
    u4 a0 = {1, 2, 3, 4}, a1 = a0, a2 = a0, a3 = a0, b0 = a0, b1 = a0, b2 = a0, b3 = a0, fb = {5, 6, 7, 8};
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    unsigned long long ts = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < NIT; ++it) {
        const unsigned ad = base ^ ((it & 7) << 5);
        if (MODE == 0) { RD(b0, ad, 0); RD(b1, ad, 8192); RD(b2, ad, 16384); RD(b3, ad, 24576); WAIT(4); RD(a0, ad, 0); RD(a1, ad, 8192); RD(a2, ad, 16384); RD(a3, ad, 24576); WAIT(4); }
        if (MODE == 1) { MFMA_V(c0, a0); MFMA_V(c1, a1); MFMA_V(c2, a2); MFMA_V(c3, a3); MFMA_V(c0, b0); MFMA_V(c1, b1); MFMA_V(c2, b2); MFMA_V(c3, b3); }
        if (MODE == 2) { RD(b0, ad, 0); RD(b1, ad, 8192); RD(b2, ad, 16384); RD(b3, ad, 24576); WAIT(4); MFMA_V(c0, a0); MFMA_V(c1, a1); MFMA_V(c2, a2); MFMA_V(c3, a3);
                         RD(a0, ad, 0); RD(a1, ad, 8192); RD(a2, ad, 16384); RD(a3, ad, 24576); WAIT(4); MFMA_V(c0, b0); MFMA_V(c1, b1); MFMA_V(c2, b2); MFMA_V(c3, b3); }
        if (MODE == 3) { RD(b0, ad, 0); RD(b1, ad, 8192); RD(b2, ad, 16384); RD(b3, ad, 24576); WAIT(4); MFMA_A(c0, a0); MFMA_A(c1, a1); MFMA_A(c2, a2); MFMA_A(c3, a3);
                         RD(a0, ad, 0); RD(a1, ad, 8192); RD(a2, ad, 16384); RD(a3, ad, 24576); WAIT(4); MFMA_A(c0, b0); MFMA_A(c1, b1); MFMA_A(c2, b2); MFMA_A(c3, b3); }
        if (MODE == 4) {   // reads interleaved between the MFMAs instead of in front of them
                         WAIT(0); MFMA_V(c0, a0); RD(b0, ad, 0); MFMA_V(c1, a1); RD(b1, ad, 8192); MFMA_V(c2, a2); RD(b2, ad, 16384); MFMA_V(c3, a3); RD(b3, ad, 24576);
                         WAIT(0); MFMA_V(c0, b0); RD(a0, ad, 0); MFMA_V(c1, b1); RD(a1, ad, 8192); MFMA_V(c2, b2); RD(a2, ad, 16384); MFMA_V(c3, b3); RD(a3, ad, 24576); }
        if (MODE == 5) {   // each fragment feeds two MFMAs (half the LDS traffic per MFMA)
                         RD(b0, ad, 0); RD(b1, ad, 8192); WAIT(2); MFMA_V(c0, a0); MFMA_V(c1, a1); MFMA_V(c2, a0); MFMA_V(c3, a1);
                         RD(a0, ad, 0); RD(a1, ad, 8192); WAIT(2); MFMA_V(c0, b0); MFMA_V(c1, b1); MFMA_V(c2, b0); MFMA_V(c3, b1); }
    }
    unsigned long long te = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float rr = 0;
    for (int i = 0; i < 16; ++i) rr += c0[i] + c1[i] + c2[i] + c3[i];
    rr += a0[0] + a1[0] + a2[0] + a3[0] + b0[0] + b1[0] + b2[0] + b3[0];
    out[threadIdx.x] = rr + stride;
    if (threadIdx.x == 0) cyc[0] = te - ts;
}
template <int MODE> void run(const char* what, float* out, unsigned long long* cyc) {
    unsigned long long h = 0;
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(256), 65536, 0, out, cyc, 0); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); }
    printf("%-72s %8.1f ticks / iteration (8 MFMAs = 256 ideal)\n", what, (double)h / NIT);
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 1024); hipMalloc(&cyc, 8);
    run<0>("8 ds_read_b128 per wave, 4 waves (8 KB/wave; 256 ticks at 128 B/clk)", out, cyc);
    run<1>("8 MFMA 32x32x16", out, cyc);
    run<2>("2 x (4 reads ; wait older ; 4 MFMA), acc in VGPRs", out, cyc);
    run<3>("2 x (4 reads ; wait older ; 4 MFMA), acc in AGPRs", out, cyc);
    run<4>("2 x 4 x (MFMA ; read)", out, cyc);
    run<5>("2 x (2 reads ; wait ; 4 MFMA): each fragment feeds two MFMAs", out, cyc);
    return 0;
}
