// What does it cost a wave to move 1 KB of weights L2 -> LDS between its MFMAs?  (a) one LDS-DMA piece (global_load_lds_dwordx4: m0 + the
// load), (b) a plain global_load_dwordx4 into registers now and a ds_write_b128 of the piece loaded one iteration earlier.
// 256 workgroups x 256 threads (one wave per SIMD, every CU busy, as in the fused MLP), each iteration = 4 MFMAs + one piece per wave;
// the source is a 1-MB buffer every workgroup walks (L2-resident after the first pass).  Prints clocks per iteration (128 = MFMA only).
// build+run:  hipcc -O3 --offload-arch=gfx950 -o /tmp/dma_vs_load profiles/probes/dma_vs_load_probe.hip && /tmp/dma_vs_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define NIT 512
#define MFMA(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(fa), "v"(fb))

template <int MODE>
__global__ __launch_bounds__(256) void k(const unsigned char* w, float* out, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds + wave * 8192;
    f16v c0 = {}, c1 = {}, c2 = {}, c3 = {};
    u4 fa = {1, 2, 3, 4}, fb = {5, 6, 7, 8}, st = {0, 0, 0, 0};
    const unsigned voff = lane * 16;
    const unsigned char* ptr = w + wave * 262144;                 // wave-uniform running pointer
    unsigned ldsw = lds0 + lane * 16;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int it = 0; it < NIT; ++it) {
        MFMA(c0); MFMA(c1);
        if (MODE == 1) {          // LDS-DMA piece
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ptr + voff),
                                             (__attribute__((address_space(3))) void*)(size_t)(lds0 + (it & 7) * 1024), 16, 0, 0);
            if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        if (MODE == 2 || MODE == 3) {          // write last iteration's piece to LDS, request the next one into the same registers
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE == 2) asm volatile("ds_write_b128 %0, %1" :: "v"(ldsw + ((it & 7) << 10)), "v"(st) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(st) : "v"(voff), "s"(ptr) : "memory");
        }
        if (MODE == 4) {          // two register stages: the load has a whole iteration more to land
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            asm volatile("ds_write_b128 %0, %1" :: "v"(ldsw + ((it & 7) << 10)), "v"((it & 1) ? fa : st) : "memory");
            if (it & 1) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(fa) : "v"(voff), "s"(ptr) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(st) : "v"(voff), "s"(ptr) : "memory");
        }
        MFMA(c2); MFMA(c3);
        ptr += 1024;
        if ((it & 63) == 63) ptr -= 65536;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = st[0] + fa[0];
    for (int i = 0; i < 16; ++i) r += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = r + lds[threadIdx.x];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* what, const unsigned char* w, float* out, unsigned long long* cyc) {
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    std::vector<unsigned long long> h(256);
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 65536, 0, w, out, cyc); hipMemcpy(h.data(), cyc, 2048, hipMemcpyDeviceToHost); }
    std::sort(h.begin(), h.end());
    printf("%-78s %7.1f clocks / iteration (median workgroup; 4 MFMAs = 128)\n", what, (double)h[128] / NIT);
}
int main() {
    unsigned char* w; float* out; unsigned long long* cyc;
    hipMalloc(&w, 2 << 20); hipMemset(w, 1, 2 << 20); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 2048);
    run<0>("4 MFMAs", w, out, cyc);
    run<1>("4 MFMAs + one LDS-DMA piece (1 KB)", w, out, cyc);
    run<3>("4 MFMAs + one global_load_dwordx4 (1 KB) into registers, waited for next iteration", w, out, cyc);
    run<2>("4 MFMAs + global_load_dwordx4 + ds_write_b128 of the previous piece", w, out, cyc);
    run<4>("the same with two register stages (a piece has two iterations to land)", w, out, cyc);
    return 0;
}
