// In-kernel clock sections of the pixel-stationary decoder convolution (conv_ring_bf16.hip, -DCONV_CLOCK_PROBE): wave 0 of every workgroup.
// build (here) + run (GPU box):  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -DCONV_CLOCK_PROBE profiles/probes/conv_ring_probe.hip -o profiles/ab/conv_probe
//                                profiles/ab/conv_probe [Cin=128] [Cout=128] [H=64 (output)] [up=1] [B=32]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
__device__ unsigned long long* g_probe = nullptr;
void master_set_error(const char*, ...) {}
int master_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -3; }
#include "../../mastermetastyletransfer_amd/csrc/conv_ring_bf16.hip"

int main(int argc, char** argv) {
    const int Cin = argc > 1 ? atoi(argv[1]) : 128, Cout = argc > 2 ? atoi(argv[2]) : 128, H = argc > 3 ? atoi(argv[3]) : 64, up = argc > 4 ? atoi(argv[4]) : 1,
              B = argc > 5 ? atoi(argv[5]) : 32;
    const int Hin = H >> up;
    const size_t nx = (size_t)B * Hin * Hin * Cin, nw = (size_t)Cout * 9 * Cin, no = (size_t)B * H * H * Cout;
    unsigned short *x, *w, *out; float* bias; unsigned long long* probe;
    hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 2); hipMalloc(&out, no * 2); hipMalloc(&bias, Cout * 4); hipMemset(bias, 0, Cout * 4);
    const int TH = Cin == 256 ? 4 : (Cin == 128 ? 8 : 16), WGS = B * (H / TH) * (H / 32);
    hipMalloc(&probe, (size_t)WGS * 8 * 8);
    std::vector<unsigned short> h(std::max(nx, nw));
    for (auto& v : h) v = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15);
    hipMemcpy(x, h.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(w, h.data(), nw * 2, hipMemcpyHostToDevice);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&]() { return conv_ring_bf16(x, w, bias, out, B, Hin, Hin, Cin, Cout, up, 1, 0); };
    for (int i = 0; i < 10; ++i) if (run() != 0) { printf("launch failed / unsupported\n"); return 1; }
    hipEventRecord(e0);
    for (int i = 0; i < 30; ++i) run();
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> p((size_t)WGS * 8);
    hipMemcpy(p.data(), probe, p.size() * 8, hipMemcpyDeviceToHost);
    double tot = 0, sec[4] = {0};
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int i = 0; i < WGS; ++i) { tot += p[i * 8]; for (int j = 0; j < 4; ++j) sec[j] += p[i * 8 + 1 + j]; tmin = std::min(tmin, p[i * 8 + 5]); tmax = std::max(tmax, p[i * 8 + 5] + p[i * 8]); }
    const double flop = 2.0 * B * H * H * 9.0 * Cin * Cout, us = ms / 30 * 1e3;
    const int steps = 9 * Cin / 64, mf = (Cout / 32) * (TH / 4) * 4 * 32;
    printf("conv %d -> %d, %dx%d out (up %d), B=%d: %d workgroups, %.1f us per launch = %.0f TFLOP/s; mean cycles per workgroup %.0f; MFMA floor %d cycles (%d steps x %d); first start -> last end %.0f cycles (100 MHz clock x?)\n",
           Cin, Cout, H, H, up, B, WGS, us, flop / us / 1e6, tot / WGS, steps * mf, steps, mf, (double)(tmax - tmin));
    const char* names[4] = {"prologue: halo gather, first ring steps, wait", "step heads (weights landed + barrier)", "fragment reads + MFMAs + DMA issue", "epilogue (bias, ReLU, LDS transpose, stores issued)"};
    for (int j = 0; j < 4; ++j) printf("   %-55s %8.0f cycles (%4.1f %%)\n", names[j], sec[j] / WGS, 100.0 * sec[j] / tot);
    return 0;
}
