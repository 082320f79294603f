// VERDICT r02 item 3 probe: "two independent 64-token workgroups per CU" for the fused MLP - the MAIN LOOP ONLY (weight stream through
// an LDS ring by LDS-DMA, A fragments by ds_read_b128, v_mfma_f32_16x16x32_bf16 with the token tile in registers), no LayerNorm, GELU,
// prologue or epilogue: does the form have the headroom at all?  Workgroup = 64 tokens = 4 waves x 16 tokens, 2-slot ring of 32-KB steps
// (64 KB of LDS, <= 128 registers per lane), two workgroups resident per CU = two waves per SIMD that share no barrier.
// The stream is the C = 256 MLP's: 32 steps of [128 rows][128 k] bf16 (1 MB per workgroup), 32 MFMAs + 32 fragment reads + 8 DMA pieces
// per wave and step.  Reference point: mlp_bf16_kernel<256> at M = 32 768 takes ~50 us for the same 34.4 GFLOP (27 % of 2.5 PF).
// build+run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -std=c++17 profiles/probes/mlp_2wg_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
union Frag { u32x4 u; bf16x8_t b; };
constexpr int SLOT = 32768, T = 32;

// MODE bit 0: LDS-DMA weight stream on; bit 1: fragment reads on (else a constant fragment); TOK16: 16-token waves (16x16x32) else 32 (32x32x16)
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void probe16(const unsigned short* __restrict__ w, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    Frag xf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) xf[s].u[e] = 0x3c003c00u + (unsigned)(lane * 77 + s * 13 + e) % 511u;
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int row0 = wave * 32 + (lane >> 4), l15 = lane & 15;          // piece i of this wave: rows row0 + 4 i (8 pieces x 4 rows)
    auto issue = [&](int t) __attribute__((always_inline)) {
        if (!(MODE & 1)) return;
        unsigned char* slot = lds + (t & 1) * SLOT + wave * 8 * 1024;
        const unsigned short* base = w + (size_t)t * 128 * 128;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row0 + 4 * i;
            const int off = row * 128 + ((l15 ^ (row & 15)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                             (__attribute__((address_space(3))) void*)(slot + i * 1024), 16, 0, 0);
        }
    };
    issue(0);
    const int r16 = lane & 15, g4 = lane >> 4;
    for (int t2 = 0; t2 < T; t2 += 2) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {                                // two steps per trip: the X fragment index stays a compile-time constant
            const int t = t2 + tt;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + 1 < T) issue(t + 1);
            const unsigned char* slot = lds + tt * SLOT;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int rt = 0; rt < 8; ++rt) {
                    Frag a;
                    if (MODE & 2) {
                        const int row = rt * 16 + r16;
                        a.u = *(const u32x4*)(slot + row * 256 + (((ks * 4 + g4) ^ (row & 15)) << 4));
                    } else a.u = xf[rt].u;
                    acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b, xf[ks + 4 * tt].b, acc[rt], 0, 0, 0);
                }
            asm volatile("" ::: "memory");
        }
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += acc[i];
    out[(size_t)blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
}

int main() {
    unsigned short* w; float* out;
    hipMalloc(&w, (size_t)T * 128 * 128 * 2); hipMalloc(&out, 4096 * 256 * 4);
    std::vector<unsigned short> h((size_t)T * 128 * 128);
    for (auto& v : h) v = 0x3c00 + (rand() & 0x1ff) - ((rand() & 1) << 15);
    hipMemcpy(w, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](int mode, int grid, const char* what) {
        auto launch = [&]() {
            switch (mode) {
                case 0: hipLaunchKernelGGL(probe16<0>, dim3(grid), dim3(256), 2 * SLOT, 0, w, out); break;
                case 1: hipLaunchKernelGGL(probe16<1>, dim3(grid), dim3(256), 2 * SLOT, 0, w, out); break;
                case 2: hipLaunchKernelGGL(probe16<2>, dim3(grid), dim3(256), 2 * SLOT, 0, w, out); break;
                default: hipLaunchKernelGGL(probe16<3>, dim3(grid), dim3(256), 2 * SLOT, 0, w, out); break;
            }
        };
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms / 20 * 1e3, gf = (double)grid * 64 * 128 * 128 * 2 * T / 1e9;
        printf("%-66s grid %4d: %7.1f us  %7.1f TFLOP/s = %4.1f %% of 2.5 PF\n", what, grid, us, gf / us * 1e3, gf / us * 1e3 / 2500 * 100);
    };
    hipFuncSetAttribute((const void*)probe16<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT);
    hipFuncSetAttribute((const void*)probe16<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT);
    hipFuncSetAttribute((const void*)probe16<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT);
    hipFuncSetAttribute((const void*)probe16<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SLOT);
    for (int grid : {512, 1024}) {
        run(0, grid, "MFMA 16x16x32 only (operands in registers), barriers");
        run(2, grid, "+ A fragments by ds_read_b128 (one per MFMA), no weight stream");
        run(1, grid, "+ LDS-DMA weight stream (8 pieces/wave/step), no fragment reads");
        run(3, grid, "+ both: the main loop of the two-workgroups-per-CU form");
    }
    return 0;
}
