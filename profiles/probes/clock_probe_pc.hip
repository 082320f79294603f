// In-kernel clock sections of the producer/consumer MLP (mlp_pc_bf16.hip): wave 0 (fc1) and wave 4 (fc2) of every workgroup.
// Ablation switches (probe builds only; results are wrong, timings are what is read): -DPC_X_NOGELU, -DPC_X_NODMA, -DPC_X_NOREAD.
// build+run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -DMLP_CLOCK_PROBE profiles/probes/clock_probe_pc.hip -o /tmp/cp_pc && /tmp/cp_pc [C]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ unsigned long long* g_probe = nullptr;
void master_set_error(const char*, ...) {}
int master_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -3; }
#include "mlp_pc_bf16.hip"

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 256, HID = 4 * C, M = 32768, WGS = M / 128;
    unsigned short *x, *w1, *w2, *out; float *b1, *b2; unsigned long long* probe;
    hipMalloc(&x, (size_t)M * C * 2); hipMalloc(&out, (size_t)M * C * 2); hipMalloc(&w1, (size_t)HID * C * 2); hipMalloc(&w2, (size_t)HID * C * 2);
    hipMalloc(&b1, HID * 4); hipMalloc(&b2, C * 4); hipMalloc(&probe, WGS * 18 * 8);
    std::vector<unsigned short> h((size_t)M * C);
    for (auto& v : h) v = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15);
    hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(w1, h.data(), (size_t)HID * C * 2, hipMemcpyHostToDevice); hipMemcpy(w2, h.data() + 7, (size_t)HID * C * 2, hipMemcpyHostToDevice);
    hipMemset(b1, 0, HID * 4); hipMemset(b2, 0, C * 4);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    for (int it = 0; it < 50; ++it) mlp_pc_bf16(x, nullptr, nullptr, w1, b1, w2, b2, x, out, M, C, HID, C, C, C, 1e-5f, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> p(WGS * 18);
    hipMemcpy(p.data(), probe, p.size() * 8, hipMemcpyDeviceToHost);
    const char* names[2] = {"fc1 wave", "fc2 wave"};
    for (int role = 0; role < 2; ++role) {
        double s[9] = {0};
        for (int i = 0; i < WGS; ++i) for (int j = 0; j < 9; ++j) s[j] += (double)p[(i * 2 + role) * 9 + j] / WGS;
        printf("C=%d %s: total %.0f | rest of tick work %.0f | DMA+LDS wait %.0f | barrier %.0f | DMA issue %.0f | fc1 MFMA phase %.0f | epilogue %.0f | GELU math %.0f | packs %.0f  (%d ticks)\n",
               C, names[role], s[8], s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], HID / 32 + 1);
    }
    return 0;
}
