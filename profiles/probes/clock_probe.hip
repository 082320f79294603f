// In-kernel clock of the fused MLP kernel: stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around the
// whole kernel in wave 0 of every workgroup (diagnostic build only: the stamps go to a buffer of their own).
// Ablation switches (probe builds only): -DMLP_X_NOGELU (note: the phase-1 MFMAs then become dead code too), -DMLP_X_NODMA1, -DMLP_X_NORD1.
// build+run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -DMLP_CLOCK_PROBE profiles/probes/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe [C=256] [M=32768]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
__device__ unsigned long long* g_probe = nullptr;
void master_set_error(const char*, ...) {}
int master_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -3; }
#include "../../mastermetastyletransfer_amd/csrc/mlp_bf16.hip"

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 256, HID = 4 * C, M = argc > 2 ? atoi(argv[2]) : 32768, WGS = M / 128;      // M = 128: one tile alone on the chip
    unsigned short *x, *w1, *w2, *out; float *b1, *b2; unsigned long long* probe;
    hipMalloc(&x, (size_t)M * C * 2); hipMalloc(&out, (size_t)M * C * 2); hipMalloc(&w1, (size_t)HID * C * 2); hipMalloc(&w2, (size_t)HID * C * 2);
    hipMalloc(&b1, HID * 4); hipMalloc(&b2, C * 4); hipMalloc(&probe, WGS * 10 * 8);
    std::vector<unsigned short> h((size_t)M * C);
    for (auto& v : h) v = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15);       // random bf16 around +-0.01..1
    hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(w1, h.data(), (size_t)HID * C * 2, hipMemcpyHostToDevice); hipMemcpy(w2, h.data() + 7, (size_t)HID * C * 2, hipMemcpyHostToDevice);
    hipMemset(b1, 0, HID * 4); hipMemset(b2, 0, C * 4);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    for (int it = 0; it < 200; ++it) mlp_bf16(x, nullptr, nullptr, w1, b1, w2, b2, x, out, M, C, HID, C, C, C, 1e-5f, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> p(WGS * 10);
    hipMemcpy(p.data(), probe, p.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (int i = 0; i < WGS; ++i) { double dc = (double)(p[4*i+1] - p[4*i]), dr = (double)(p[4*i+3] - p[4*i+2]); ghz.push_back(dc / dr * 0.1); cyc.push_back(dc); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    printf("workgroups %d: in-kernel clock median %.3f GHz (min %.3f max %.3f); cycles per workgroup median %.0f (%.1f us)\n", WGS, ghz[WGS/2], ghz[0], ghz[WGS-1], cyc[WGS/2], cyc[WGS/2] / ghz[WGS/2] / 1e3);
    double sec[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < WGS; ++i) for (int j = 0; j < 6; ++j) sec[j] += (double)p[WGS * 4 + i * 6 + j] / WGS;
    printf("mean cycles per workgroup by section: wait+barrier %.0f | DMA issue %.0f | phase-1 steps %.0f | phase-2 steps %.0f | prologue %.0f | epilogue %.0f  (stamps add ~40 cycles each)\n", sec[0], sec[1], sec[2], sec[3], sec[4], sec[5]);
    return 0;
}
