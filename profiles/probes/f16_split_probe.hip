// Does gfx950's v_mfma_f32_32x32x16_f16 honour SUBNORMAL half operands, and how exact is a three-product split GEMM with IEEE-half
// halves against bf16 halves on the real matrix core?  (r05: the CPU simulation profiles/tools/x3_split_sim.py says a half split is
// ~10x closer to float64 than the bf16 split, but only if subnormal `lo` halves are not flushed.)
// build+run:  hipcc -O3 --offload-arch=gfx950 -o /tmp/f16_split_probe profiles/probes/f16_split_probe.hip && /tmp/f16_split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

// (1) subnormal operands: A = 2^-20 (subnormal half), B = 1024 -> every output = 16 * 2^-10 = 2^-6 when honoured, 0 when flushed
__global__ void denorm_kernel(float* out, float a_val, float b_val) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    f16v acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    out[threadIdx.x] = acc[0];
    // conversions: does fp32 -> half produce subnormals, and does half -> fp32 read them?
    _Float16 h = (_Float16)a_val;
    out[64 + threadIdx.x] = (float)h;
}

// (2) C[32x32] = A[32xK] * B[32xK]^T with operands split in registers, one wave; FMT 0 = bf16 halves, 1 = half halves
template <int FMT>
__global__ void split_gemm(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f16v acc = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        float av[8], bv[8];
        for (int i = 0; i < 8; ++i) { av[i] = A[r * K + k0 + 8 * h + i]; bv[i] = B[r * K + k0 + 8 * h + i]; }
        if (FMT == 0) {
            b8 ah, al, bh, bl;
            for (int i = 0; i < 8; ++i) {
                ah[i] = (__bf16)av[i]; al[i] = (__bf16)(av[i] - (float)ah[i]);
                bh[i] = (__bf16)bv[i]; bl[i] = (__bf16)(bv[i] - (float)bh[i]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        } else {
            h8 ah, al, bh, bl;
            for (int i = 0; i < 8; ++i) {
                ah[i] = (_Float16)av[i]; al[i] = (_Float16)(av[i] - (float)ah[i]);
                bh[i] = (_Float16)bv[i]; bl[i] = (_Float16)(bv[i] - (float)bh[i]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        }
    }
    // accumulator layout of 32x32: element e of lane (r, h): row 8*(e>>2) + 4*h + (e&3) of A, column r of B
    for (int e = 0; e < 16; ++e) C[(8 * (e >> 2) + 4 * h + (e & 3)) * 32 + r] = acc[e];
}

int main() {
    float* out; hipMalloc(&out, 1024);
    float h[128];
    const float cases[][2] = {{9.5367431640625e-07f /* 2^-20 */, 1024.f}, {5.9604644775390625e-08f /* 2^-24: the smallest subnormal */, 65504.f},
                              {1024.f, 9.5367431640625e-07f}, {3.0517578125e-05f /* 2^-15 */, 1.f}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, out, c[0], c[1]);
        hipMemcpy(h, out, 512, hipMemcpyDeviceToHost);
        printf("mfma_f16: a = %.6e  b = %.6e : out = %.9e  expected if subnormals are honoured %.9e ; half(a) read back %.9e\n", c[0], c[1], h[0],
               16.0 * (double)c[0] * (double)c[1], h[64]);
    }
    const int K = 1024;
    std::vector<float> A(32 * K), B(32 * K), C(32 * 32);
    float *dA, *dB, *dC; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
    const double scales[][2] = {{1.0, 1.0}, {1.0, 0.03}, {0.01, 0.03}, {30.0, 0.03}, {1e-3, 1e-3}};
    for (auto& sc : scales) {
        srand(1);
        auto nrm = [] { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); };
        for (auto& x : A) x = (float)(nrm() * sc[0]);
        for (auto& x : B) x = (float)(nrm() * sc[1]);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        double ref[32 * 32], mx = 0; float f32[32 * 32];
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double s = 0; float sf = 0;
            for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[j * K + k]; sf = fmaf(A[i * K + k], B[j * K + k], sf); }
            ref[i * 32 + j] = s; f32[i * 32 + j] = sf; mx = fmax(mx, fabs(s));
        }
        double e[3] = {0, 0, 0};
        for (int i = 0; i < 1024; ++i) e[2] = fmax(e[2], fabs(f32[i] - ref[i]));
        for (int fmt = 0; fmt < 2; ++fmt) {
            if (fmt == 0) hipLaunchKernelGGL(split_gemm<0>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
            else hipLaunchKernelGGL(split_gemm<1>, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
            hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
            for (int i = 0; i < 1024; ++i) e[fmt] = fmax(e[fmt], fabs(C[i] - ref[i]));
        }
        printf("split GEMM 32x32x%d, operand scales %g x %g, max|ref| %.3e: max err / max|ref|: bf16 halves %.2e   half halves %.2e   fp32 fma chain %.2e\n",
               K, sc[0], sc[1], mx, e[0] / mx, e[1] / mx, e[2] / mx);
    }
    return 0;
}
