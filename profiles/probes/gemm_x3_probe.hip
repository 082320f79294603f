// In-kernel clock sections of the split-bf16 GEMM / conv (gemm_x3.hip): s_memtime stamps of wave 0 of every workgroup around the top
// wait, the barrier, the look-ahead issue, the MFMA section, the A-register wait and the split + LDS store (diagnostic build only).
// build+run on the GPU box:  hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude -mllvm -amdgpu-mfma-vgpr-form -DX3_CLOCK_PROBE profiles/probes/gemm_x3_probe.hip -o /tmp/gemm_x3_probe && /tmp/gemm_x3_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ unsigned long long* g_probe = nullptr;
void master_set_error(const char*, ...) {}
int master_check_launch(const char*) { return hipGetLastError() == hipSuccess ? 0 : -3; }
#include "../../mastermetastyletransfer_amd/csrc/gemm_x3.hip"

static void run(const char* name, bool conv, int M, int N, int K, int B, int H, int Cin, int up) {
    float *a, *out, *bias; unsigned short* w; unsigned long long* probe;
    const size_t an = conv ? (size_t)B * H * H * Cin : (size_t)M * K;
    if (conv) { const int Ho = H << up; M = B * Ho * Ho; K = 9 * Cin; }
    hipMalloc(&a, an * 4); hipMalloc(&out, (size_t)M * N * 4); hipMalloc(&bias, N * 4); hipMalloc(&w, (size_t)N * K * 4);
    const int wgs = 65536;
    hipMalloc(&probe, (size_t)wgs * 8 * 8); hipMemset(probe, 0, (size_t)wgs * 8 * 8);
    std::vector<float> ha(an); for (auto& v : ha) v = (rand() & 0xffff) / 32768.0f - 1.0f;
    hipMemcpy(a, ha.data(), an * 4, hipMemcpyHostToDevice);
    std::vector<unsigned short> hw((size_t)N * K * 2); for (auto& v : hw) v = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15) - 0x0400;
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice); hipMemset(bias, 0, N * 4);
    hipMemcpyToSymbol(HIP_SYMBOL(g_probe), &probe, sizeof(probe));
    const int asplit = getenv("X3_PROBE_ASPLIT") ? 1 : 0;        // the A operand is read as if in the split layout (timing only: same bytes)
    auto go = [&]() { if (conv) gemm_x3_conv3x3(a, w, bias, out, B, H, H, Cin, N, up, 1, 0, 0, 0, asplit, 0); else gemm_x3_linear(a, w, bias, nullptr, nullptr, out, M, N, K, K, N, 0, 0, 0, asplit, 0); };
    for (int i = 0; i < 20; ++i) go();
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) go();
    hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> p((size_t)wgs * 8); hipMemcpy(p.data(), probe, p.size() * 8, hipMemcpyDeviceToHost);
    double sec[7] = {0}; int n = 0;
    for (int i = 0; i < wgs; ++i) if (p[(size_t)i * 8 + 6]) { ++n; for (int j = 0; j < 7; ++j) sec[j] += (double)p[(size_t)i * 8 + j]; }
    printf("%s: %.1f us per launch (probe build), %.1f TFLOP/s; %d workgroups; mean cycles per workgroup (wave 0): total-to-loop-end %.0f | vmcnt+lgkm wait %.0f | barrier %.0f | issue %.0f | compute %.0f | A-register wait %.0f | split+store %.0f ; chunks %d, MFMA floor per chunk %d cycles per wave\n",
           name, ms / 20 * 1e3, 2.0 * M * N * K / (ms / 20 * 1e-3) / 1e12, n, sec[6] / n, sec[0] / n, sec[1] / n, sec[2] / n, sec[3] / n, sec[4] / n, sec[5] / n, K / 32, 0);
    hipFree(a); hipFree(out); hipFree(bias); hipFree(w); hipFree(probe);
}
int main() {
    run("conv 128->128 @64^2 bs32", true, 0, 128, 0, 32, 64, 128, 0);
    run("conv 256->128 @32^2 bs32", true, 0, 128, 0, 32, 32, 256, 0);
    run("conv 64->64 @128^2(up) bs32", true, 0, 64, 0, 32, 64, 64, 1);
    run("conv 32->32 @256^2(up) bs32", true, 0, 32, 0, 32, 128, 32, 1);
    run("linear 32768x256x1024", false, 32768, 256, 1024, 0, 0, 0, 0);
    run("linear 65536x768x256", false, 65536, 768, 256, 0, 0, 0, 0);
    run("linear 262144x384x128", false, 262144, 384, 128, 0, 0, 0, 0);
    return 0;
}
