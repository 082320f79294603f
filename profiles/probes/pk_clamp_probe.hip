// Does the VOP3P clamp output modifier act on both halves of v_pk_fma_f32 on gfx950?  (gelu_poly2 in common.h relies on it.)
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_clamp profiles/probes/pk_clamp_probe.hip ; prints the clamped pairs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const f2* in, f2* out) {
    f2 x = in[threadIdx.x], one = {1.f, 1.f}, p;
    asm("v_pk_fma_f32 %0, %1, %2, 0.5 op_sel_hi:[1,1,0] clamp" : "=v"(p) : "v"(x), "v"(one));
    out[threadIdx.x] = p;
}
int main() {
    const float h[8] = {-3.f, 2.f, 0.25f, -0.25f, 1e30f, -1e30f, 0.6f, -0.6f};
    f2 *d, *o; float r[8];
    hipMalloc(&d, 32); hipMalloc(&o, 32); hipMemcpy(d, h, 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, o); hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
    const float want[8] = {0.f, 1.f, 0.75f, 0.25f, 1.f, 0.f, 1.f, 0.f};
    int ok = 1;
    for (int i = 0; i < 8; ++i) { printf("x=%g -> %g (want %g)\n", h[i], r[i], want[i]); ok &= r[i] == want[i]; }
    printf(ok ? "PK_CLAMP_OK\n" : "PK_CLAMP_BROKEN\n");
    return !ok;
}
