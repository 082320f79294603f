// In-kernel clock sections of the fused attention + proj kernel (window_attn_proj_bf16.hip, -DATTN_PROJ_PROBE): wave 0 of every
// persistent workgroup accumulates s_memtime deltas per phase of the per-window loop.
// build+run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -mllvm -amdgpu-mfma-vgpr-form -DATTN_PROJ_PROBE profiles/probes/attn_proj_probe.hip -o /tmp/ap_probe && /tmp/ap_probe [B] [shift]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ unsigned long long* g_probe = nullptr;
void master_set_error(const char*, ...) {}
int master_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -3; }
#include "../../mastermetastyletransfer_amd/csrc/window_attn_proj_bf16.hip"

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, shift = argc > 2 ? atoi(argv[2]) : 4;
    const int H = 32, W = 32, T = H * W, Cc = 256, nWh = 5, nW = 25, N = 49;
    std::vector<int> src(nW * 64, -1); std::vector<unsigned char> region(nW * 64, 0);
    for (int wy = 0; wy < nWh; ++wy) for (int wx = 0; wx < nWh; ++wx) for (int t = 0; t < 49; ++t) {
        int py = (wy * 7 + t / 7 + shift) % 35, px = (wx * 7 + t % 7 + shift) % 35;
        src[(wy * nWh + wx) * 64 + t] = (py < H && px < W) ? py * W + px : -1;
        region[(wy * nWh + wx) * 64 + t] = (unsigned char)((py >= 35 - shift ? 2 : (py >= 28 ? 1 : 0)) * 3 + (px >= 35 - shift ? 2 : (px >= 28 ? 1 : 0)));
    }
    unsigned short *qkv, *out, *res, *wproj, *pad; float *bproj, *bias64; int* dsrc; unsigned char* dreg; unsigned long long* probe;
    hipMalloc(&qkv, (size_t)B * T * 3 * Cc * 2); hipMalloc(&out, (size_t)B * T * Cc * 2); hipMalloc(&res, (size_t)B * T * Cc * 2);
    hipMalloc(&wproj, Cc * Cc * 2); hipMalloc(&pad, 3 * Cc * 2); hipMalloc(&bproj, Cc * 4); hipMalloc(&bias64, 8 * 64 * 64 * 4);
    hipMalloc(&dsrc, src.size() * 4); hipMalloc(&dreg, region.size()); hipMalloc(&probe, 256 * 13 * 8);
    std::vector<unsigned short> h((size_t)B * T * 3 * Cc);
    for (auto& v : h) v = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15);
    hipMemcpy(qkv, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(res, h.data() + 7, (size_t)B * T * Cc * 2, hipMemcpyHostToDevice);
    std::vector<unsigned short> wsmall(Cc * Cc); for (auto& v : wsmall) v = 0x3a00 + (rand() & 0xff) - ((rand() & 1) << 15);
    hipMemcpy(wproj, wsmall.data(), Cc * Cc * 2, hipMemcpyHostToDevice); hipMemcpy(pad, h.data() + 3, 3 * Cc * 2, hipMemcpyHostToDevice);
    std::vector<float> bz(8 * 64 * 64, 0.f); for (int i = 0; i < 8 * 64; ++i) for (int j = 49; j < 64; ++j) bz[i * 64 + j] = -1e30f;
    hipMemcpy(bias64, bz.data(), bz.size() * 4, hipMemcpyHostToDevice); hipMemset(bproj, 0, Cc * 4);
    hipMemcpy(dsrc, src.data(), src.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dreg, region.data(), region.size(), hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    auto run = [&]() { return window_attn_proj_bf16(qkv, qkv + Cc, qkv + 2 * Cc, 3 * Cc, 3 * Cc, 3 * Cc, pad + Cc, pad + 2 * Cc, 0, 0, bias64, dsrc, shift ? dreg : nullptr,
                                                    wproj, bproj, res, Cc, out, Cc, B, T, 8, nW, N, 0.1767767f, 0); };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 20; ++it) if (run() != 0) { printf("launch failed\n"); return 1; }
    hipEventRecord(e0);
    for (int it = 0; it < 50; ++it) run();
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int grid = B * nW < 256 ? B * nW : 256;
    std::vector<unsigned long long> p(256 * 13);
    hipMemcpy(p.data(), probe, p.size() * 8, hipMemcpyDeviceToHost);
    double tot = 0, win = 0, setup = 0, sec[10] = {0}, maxwg = 0;
    for (int i = 0; i < grid; ++i) { tot += p[i * 13]; if (p[i * 13] > maxwg) maxwg = p[i * 13]; win += p[i * 13 + 1]; setup += p[i * 13 + 2]; for (int j = 0; j < 10; ++j) sec[j] += p[i * 13 + 3 + j]; }
    printf("B %d shift %d: %.1f us per launch (events); mean cycles per workgroup %.0f (longest %.0f) for %.2f windows; setup before the loop %.0f; %.0f per window in the loop (128 MFMAs per wave = 4096 of them)\n",
           B, shift, ms / 50 * 1e3, tot / grid, maxwg, win / grid, setup / grid, (tot - setup) / win);
    const char* names[10] = {"wait gathered tiles + barrier", "fragment pull (2 heads) + barrier", "next gather issue + residual loads", "head 0: S, bias, mask, softmax, PV, O write",
                             "head 1", "O barrier", "proj (64 MFMAs)", "barrier + fp32 stage + barrier", "row pass: + residual, 16-B stores", ""};
    for (int j = 0; j < 9; ++j) printf("   %-48s %7.0f cycles per window (%4.1f %%)\n", names[j], sec[j] / win, 100.0 * sec[j] / (tot - setup));
    return 0;
}
