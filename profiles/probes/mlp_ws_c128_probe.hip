// Probe: a WEIGHT-STATIONARY form of the Swin stage-1 MLP (C = 128, hidden 512): the 256 KB of W1 / W2 live in the register file of the
// CU (4 waves x 256 registers), 64-token tiles stream through LDS, the hidden activations cross between the waves through LDS.
// Main loop only - no LayerNorm, residual or store (an 8-byte sum per lane keeps the compiler honest):
//   per tile:  X tile (LDS-DMA, prefetched)  ->  fc1: H^T[128 hidden of this wave][64 tok] = W1_w . X^T (64 MFMAs, X fragments from LDS)
//              -> GELU (polynomial, common.h) -> bf16 H[tok][512] into LDS -> barrier -> fc2: O^T[32 out of this wave][64 tok] = W2_w . H^T
//              (64 MFMAs, H fragments from LDS) -> barrier.
// Today's mlp_c128_kernel: ~73 k clocks per 256 tokens = ~18 k per 64 tokens (136 us per 262 144-token launch), of which prologue +
// epilogue ~3 k per 64 tokens.  build+run:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude profiles/probes/mlp_ws_c128_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../mastermetastyletransfer_amd/csrc/common.h"
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
union Frag { u32x4 u; bf16x8_t b; };
void master_set_error(const char*, ...) {}
int master_check_launch(const char*) { return 0; }
ProfScope::ProfScope(int, hipStream_t) {} ProfScope::~ProfScope() {}

constexpr int C = 128, HID = 512, XT = 64 * C * 2, HT = 64 * HID * 2;     // X tile 16 KB, H tile 64 KB
#define LDS_READ128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define LGKM_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

template <int GELU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void probe(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w1,
                                                                                         const unsigned short* __restrict__ w2, float* __restrict__ out, int ntiles,
                                                                                         unsigned long long* __restrict__ clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // X[2] | H
    unsigned char* Xs = lds; unsigned char* Hs = lds + 2 * XT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    Frag a1[4][8], a2[32];                                                // W1 rows [128 wave + 32 ht + r][16 s + 8 h ..], W2 rows [32 wave + r][16 s + 8 h ..]
#pragma unroll
    for (int ht = 0; ht < 4; ++ht)
#pragma unroll
        for (int s = 0; s < 8; ++s) a1[ht][s].u = *(const u32x4*)(w1 + (size_t)(wave * 128 + ht * 32 + r) * C + 16 * s + 8 * h);
#pragma unroll
    for (int s = 0; s < 32; ++s) a2[s].u = *(const u32x4*)(w2 + (size_t)(wave * 32 + r) * HID + 16 * s + 8 * h);
    auto gather = [&](int tile, int buf) __attribute__((always_inline)) {   // 64 rows x 256 B = 16 pieces; wave moves 4; chunk swizzle by row & 15
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int I = wave * 4 + i, row = 4 * I + (lane >> 4);
            const unsigned short* sp = x + ((size_t)tile * 64 + row) * C + (((lane & 15) ^ (row & 15)) << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp, (__attribute__((address_space(3))) void*)(Xs + buf * XT + I * 1024), 16, 0, 0);
        }
    };
    int tile = blockIdx.x, buf = 0;
    if (tile < ntiles) gather(tile, 0);
    float keep = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int done = 0;
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1, ++done) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // X tile landed; previous tile's fc2 reads of H are done
        if (tile + (int)gridDim.x < ntiles) gather(tile + gridDim.x, buf ^ 1);
        // ---- fc1 + GELU + H write, one hidden tile (32 units) at a time ------------------------------------------------------------
        const unsigned xb = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)(Xs + buf * XT + r * 256);
        const int xsw = (h ^ (r & 15)) << 4;
#pragma unroll
        for (int ht = 0; ht < 4; ++ht) {
            f32x16 acc[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
            u32x4 xa[2][2];
            LDS_READ128(xa[0][0], xb + (0 ^ xsw), 0); LDS_READ128(xa[0][1], xb + (0 ^ xsw), 8192);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s + 1 < 8) { const unsigned ad = xb + ((32 * (s + 1)) ^ xsw); LDS_READ128(xa[(s + 1) & 1][0], ad, 0); LDS_READ128(xa[(s + 1) & 1][1], ad, 8192); LGKM_WAIT(2); }
                else LGKM_WAIT(0);
                Frag x0, x1; x0.u = xa[s & 1][0]; x1.u = xa[s & 1][1];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[ht][s].b, x0.b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[ht][s].b, x1.b, acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int tok = tt * 32 + r;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x2 v0 = {acc[tt][4 * g], acc[tt][4 * g + 1]}, v1 = {acc[tt][4 * g + 2], acc[tt][4 * g + 3]};
                    if (GELU) { v0 = gelu_poly2(v0); v1 = gelu_poly2(v1); }
                    u32x2 o; o[0] = pack_bf16x2(v0[0], v0[1]); o[1] = pack_bf16x2(v1[0], v1[1]);
                    *(u32x2*)(Hs + tok * 1024 + (((wave * 16 + ht * 4 + g) ^ (tok & 15)) << 4) + 8 * h) = o;   // hidden 128 wave + 32 ht + 8 g + 4 h ..
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                      // H complete
        // ---- fc2 ---------------------------------------------------------------------------------------------------------------------
        f32x16 oacc[2];
#pragma unroll
        for (int e = 0; e < 16; ++e) { oacc[0][e] = 0.f; oacc[1][e] = 0.f; }
        {
            const unsigned hb = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)(Hs + r * 1024);
            u32x4 ha[2][2];
            LDS_READ128(ha[0][0], hb + (0 ^ xsw), 0); LDS_READ128(ha[0][1], hb + (0 ^ xsw), 32768);
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                if (s + 1 < 32) { const unsigned ad = hb + ((32 * (s + 1)) ^ xsw); LDS_READ128(ha[(s + 1) & 1][0], ad, 0); LDS_READ128(ha[(s + 1) & 1][1], ad, 32768); LGKM_WAIT(2); }
                else LGKM_WAIT(0);
                Frag h0, h1; h0.u = ha[s & 1][0]; h1.u = ha[s & 1][1];
                oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[s].b, h0.b, oacc[0], 0, 0, 0);
                oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[s].b, h1.b, oacc[1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) keep += oacc[0][e] + oacc[1][e];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[(size_t)blockIdx.x * 256 + tid] = keep;
    if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = (unsigned long long)done; }
}

int main() {
    const int M = 262144, ntiles = M / 64;
    unsigned short *x, *w1, *w2; float* out; unsigned long long* clk;
    hipMalloc(&x, (size_t)M * C * 2); hipMalloc(&w1, HID * C * 2); hipMalloc(&w2, C * HID * 2); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 256 * 16);
    std::vector<unsigned short> h((size_t)M * C);
    for (auto& v : h) v = 0x3c00 + (rand() & 0x1ff) - ((rand() & 1) << 15);
    hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    std::vector<unsigned short> ws(HID * C); for (auto& v : ws) v = 0x3a00 + (rand() & 0xff) - ((rand() & 1) << 15);
    hipMemcpy(w1, ws.data(), ws.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w2, ws.data(), ws.size() * 2, hipMemcpyHostToDevice);
    const int lds = 2 * XT + HT;
    hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int gelu = 0; gelu < 2; ++gelu) {
        auto launch = [&]() { if (gelu) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), lds, 0, x, w1, w2, out, ntiles, clk);
                              else hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), lds, 0, x, w1, w2, out, ntiles, clk); };
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(512); hipMemcpy(c.data(), clk, 4096, hipMemcpyDeviceToHost);
        double cyc = 0, tiles = 0; for (int i = 0; i < 256; ++i) { cyc += c[2 * i]; tiles += c[2 * i + 1]; }
        const double us = ms / 20 * 1e3, gf = 4.0 * M * C * HID / 1e9;
        printf("weight-stationary main loop, GELU %s: %7.1f us per launch of %d tokens = %6.1f TFLOP/s (%4.1f %% of 2.5 PF); %6.0f clocks per 64-token tile in the loop (128 MFMAs = 4096)\n",
               gelu ? "on " : "off", us, M, gf / us * 1e3, gf / us * 1e3 / 2500 * 100, cyc / tiles);
    }
    return 0;
}
