// In-kernel clock sections of the fused split-bf16 MLP (mlp_x3.hip): s_memtime stamps of wave 0 of every workgroup around wait+barrier,
// DMA issue, phase-1 steps, GELU + split, phase-2 steps, prologue, epilogue (diagnostic build only: the stamps go to a buffer of their own).
// build+run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -DMLPX3_CLOCK_PROBE profiles/probes/mlp_x3_probe.hip -o /tmp/mlp_x3_probe && /tmp/mlp_x3_probe [C=256] [M=32768]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
__device__ unsigned long long* g_probe = nullptr;
void master_set_error(const char*, ...) {}
int master_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -3; }
#include "../../mastermetastyletransfer_amd/csrc/mlp_x3.hip"

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 256, HID = 4 * C, M = argc > 2 ? atoi(argv[2]) : 32768, TOK = C == 256 ? 128 : 256, WGS = (M + TOK - 1) / TOK;
    float *x, *out, *b1, *b2; unsigned short *w1, *w2; unsigned long long* probe;
    hipMalloc(&x, (size_t)M * C * 4); hipMalloc(&out, (size_t)M * C * 4); hipMalloc(&w1, (size_t)HID * C * 4); hipMalloc(&w2, (size_t)HID * C * 4);
    hipMalloc(&b1, HID * 4); hipMalloc(&b2, C * 4); hipMalloc(&probe, (size_t)WGS * 11 * 8);
    std::vector<float> hx((size_t)M * C);
    for (auto& v : hx) v = ((rand() & 0xffff) / 32768.0f - 1.0f);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    std::vector<unsigned short> hw((size_t)HID * C * 2);
    for (auto& v : hw) v = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15) - 0x0400;        // random bf16 of magnitude ~0.03..0.06, both signs
    hipMemcpy(w1, hw.data(), hw.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w2, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemset(b1, 0, HID * 4); hipMemset(b2, 0, C * 4);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    for (int it = 0; it < 200; ++it) mlp_x3(x, nullptr, nullptr, w1, b1, w2, b2, x, out, M, C, HID, C, C, C, 1e-5f, 0);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int it = 0; it < 50; ++it) mlp_x3(x, nullptr, nullptr, w1, b1, w2, b2, x, out, M, C, HID, C, C, C, 1e-5f, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> p((size_t)WGS * 11);
    hipMemcpy(p.data(), probe, p.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (int i = 0; i < WGS; ++i) { double dc = (double)(p[4*i+1] - p[4*i]), dr = (double)(p[4*i+3] - p[4*i+2]); ghz.push_back(dc / dr * 0.1); cyc.push_back(dc); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    printf("C=%d M=%d workgroups %d: %.1f us per launch (probe build); in-kernel clock median %.3f GHz; cycles per workgroup median %.0f (%.1f us)\n", C, M, WGS, ms / 50 * 1e3,
           ghz[WGS/2], cyc[WGS/2], cyc[WGS/2] / ghz[WGS/2] / 1e3);
    double sec[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < WGS; ++i) for (int j = 0; j < 7; ++j) sec[j] += (double)p[(size_t)WGS * 4 + i * 7 + j] / WGS;
    printf("mean cycles per workgroup (wave 0) by section: wait+barrier %.0f | DMA issue %.0f | phase-1 steps %.0f | GELU+split %.0f | phase-2 steps %.0f | prologue %.0f | epilogue %.0f\n",
           sec[0], sec[1], sec[2], sec[3], sec[4], sec[5], sec[6]);
    const double steps = (double)(HID / 64) * (C / 64);
    printf("MFMA floor: %.0f cycles per workgroup (%d steps x 48 MFMAs x 32)\n", steps * 48 * 32, (int)steps);
    return 0;
}
