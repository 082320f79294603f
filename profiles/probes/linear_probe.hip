// In-kernel clock sections of the token-stationary Linear (linear_ts_bf16.hip, -DLIN_CLOCK_PROBE): wave 0 of every workgroup.
// build+run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -DLIN_CLOCK_PROBE profiles/probes/linear_probe.hip -o /tmp/lp && /tmp/lp [N=256] [M=32768] [res=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ unsigned long long* g_probe = nullptr;
void master_set_error(const char*, ...) {}
int master_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -3; }
#include "../../mastermetastyletransfer_amd/csrc/linear_ts_bf16.hip"

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 256, M = argc > 2 ? atoi(argv[2]) : 32768, res = argc > 3 ? atoi(argv[3]) : 0, K = 256, WGS = (M + 127) / 128;
    unsigned short *x, *w, *out, *r2; float* bias; unsigned long long* probe;
    hipMalloc(&x, (size_t)M * K * 2); hipMalloc(&out, (size_t)M * N * 2); hipMalloc(&r2, (size_t)M * N * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&bias, N * 4);
    hipMalloc(&probe, (size_t)WGS * 8 * 8);
    std::vector<unsigned short> h((size_t)M * K);
    for (auto& v : h) v = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15);
    hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w, h.data() + 5, (size_t)N * K * 2, hipMemcpyHostToDevice);
    hipMemcpy(r2, h.data(), (size_t)M * (N < K ? N : K) * 2, hipMemcpyHostToDevice); hipMemset(bias, 0, N * 4);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&]() { return linear_ts_bf16(x, nullptr, nullptr, w, bias, nullptr, res ? r2 : nullptr, out, M, N, K, K, N, N, 1e-5f, 0); };
    for (int i = 0; i < 20; ++i) if (run() != 0) { printf("launch failed\n"); return 1; }
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) run();
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> p((size_t)WGS * 8);
    hipMemcpy(p.data(), probe, p.size() * 8, hipMemcpyDeviceToHost);
    double tot = 0, sec[5] = {0};
    for (int i = 0; i < WGS; ++i) { tot += p[i * 8]; for (int j = 0; j < 5; ++j) sec[j] += p[i * 8 + 1 + j]; }
    printf("linear M=%d N=%d K=%d res=%d: %.1f us per launch (events, back to back); mean cycles per workgroup %.0f (incl. the final store drain); MFMA floor %d\n",
           M, N, K, res, ms / 50 * 1e3, tot / WGS, (N / 128) * (K / 128) * 16 * 32);
    const char* names[5] = {"prologue: first DMAs, params, residual + X loads, wait", "waits at step heads (weights landed + barrier)", "DMA issue of later steps", "MFMA steps (incl. fragment reads)", "epilogues (LDS transpose, bias / residual, stores issued)"};
    for (int j = 0; j < 5; ++j) printf("   %-62s %7.0f cycles (%4.1f %%)\n", names[j], sec[j] / WGS, 100.0 * sec[j] / tot);
    return 0;
}
