// In-kernel clock sections of the fused Swin stage-1 attention block (attn_block_bf16.hip, -DATTN_CLOCK_PROBE): wave 0 of every
// persistent workgroup accumulates s_memtime deltas per phase of the per-window loop.
// build+run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -DATTN_CLOCK_PROBE profiles/probes/attn_block_probe.hip -o /tmp/attn_probe && /tmp/attn_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
__device__ unsigned long long* g_probe = nullptr;
void master_set_error(const char*, ...) {}
int master_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -3; }
#include "../../mastermetastyletransfer_amd/csrc/attn_block_bf16.hip"

int main(int argc, char** argv) {
    // Swin stage 1 at 256x256, content+style of bs=32: B = 64 maps of 64x64 tokens, 128 channels, 7x7 windows padded to 70x70 = 100 windows
    const int B = 64, H = 64, W = 64, T = H * W, Cc = 128, nWh = 10, nW = 100, N = 49, shift = argc > 1 ? atoi(argv[1]) : 3;
    std::vector<int> src(nW * 64, -1); std::vector<unsigned char> region(nW * 64, 0);
    for (int wy = 0; wy < nWh; ++wy) for (int wx = 0; wx < nWh; ++wx) for (int t = 0; t < 49; ++t) {
        int py = (wy * 7 + t / 7 + shift) % 70, px = (wx * 7 + t % 7 + shift) % 70;
        src[(wy * nWh + wx) * 64 + t] = (py < H && px < W) ? py * W + px : -1;
        region[(wy * nWh + wx) * 64 + t] = (unsigned char)((py >= 70 - shift ? 2 : (py >= 63 ? 1 : 0)) * 3 + (px >= 70 - shift ? 2 : (px >= 63 ? 1 : 0)));
    }
    unsigned short *x, *out, *wqkv, *wproj; float *bqkv, *bproj, *bias64, *g, *b; int* dsrc; unsigned char* dreg; unsigned long long* probe;
    hipMalloc(&x, (size_t)B * T * Cc * 2); hipMalloc(&out, (size_t)B * T * Cc * 2); hipMalloc(&wqkv, 3 * Cc * Cc * 2); hipMalloc(&wproj, Cc * Cc * 2);
    hipMalloc(&bqkv, 3 * Cc * 4); hipMalloc(&bproj, Cc * 4); hipMalloc(&bias64, 4 * 64 * 64 * 4); hipMalloc(&g, Cc * 4); hipMalloc(&b, Cc * 4);
    hipMalloc(&dsrc, src.size() * 4); hipMalloc(&dreg, region.size()); hipMalloc(&probe, 256 * 10 * 8);
    std::vector<unsigned short> h((size_t)B * T * Cc);
    for (auto& v : h) v = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15);
    hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(wqkv, h.data(), 3 * Cc * Cc * 2, hipMemcpyHostToDevice); hipMemcpy(wproj, h.data() + 11, Cc * Cc * 2, hipMemcpyHostToDevice);
    std::vector<float> ones(4 * 64 * 64, 0.f); for (int i = 0; i < 4 * 64; ++i) for (int j = 49; j < 64; ++j) ones[i * 64 + j] = -1e30f;
    hipMemcpy(bias64, ones.data(), ones.size() * 4, hipMemcpyHostToDevice);
    hipMemset(bqkv, 0, 3 * Cc * 4); hipMemset(bproj, 0, Cc * 4); hipMemset(b, 0, Cc * 4);
    std::vector<float> gam(Cc, 1.f); hipMemcpy(g, gam.data(), Cc * 4, hipMemcpyHostToDevice);
    hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dreg, region.data(), region.size(), hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 20; ++it) attn_block_bf16(x, g, b, wqkv, bqkv, wproj, bproj, bias64, dsrc, shift ? dreg : nullptr, out, B, T, Cc, 4, nW, N, 1e-5f, 0.1767767f, 0);
    hipEventRecord(e0);
    for (int it = 0; it < 50; ++it) attn_block_bf16(x, g, b, wqkv, bqkv, wproj, bproj, bias64, dsrc, shift ? dreg : nullptr, out, B, T, Cc, 4, nW, N, 1e-5f, 0.1767767f, 0);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> p(256 * 10);
    hipMemcpy(p.data(), probe, p.size() * 8, hipMemcpyDeviceToHost);
    double tot = 0, win = 0, sec[8] = {0};
    for (int i = 0; i < 256; ++i) { tot += p[i * 10]; win += p[i * 10 + 1]; for (int j = 0; j < 8; ++j) sec[j] += p[i * 10 + 2 + j]; }
    printf("shift %d: %.1f us per launch (events); mean cycles per workgroup %.0f for %.1f windows = %.0f per window (88 MFMAs = 2816 of them)\n", shift, ms / 50 * 1e3, tot / 256, win / 256, tot / win);
    const char* names[8] = {"wait gathered tile + barrier", "LayerNorm + barrier + next gather", "QKV (48 MFMAs)", "S^T + bias + mask (8 MFMAs)", "softmax: exp2, pack, denominators (8 MFMAs)", "P.V (8 MFMAs)", "heads meet in LDS (2 barriers)", "proj (16 MFMAs) + residual + store"};
    for (int j = 0; j < 8; ++j) printf("   %-48s %7.0f cycles per window (%4.1f %%)\n", names[j], sec[j] / win, 100.0 * sec[j] / tot);
    return 0;
}
