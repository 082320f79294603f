"""The metric's named GEMM, measured where it runs: the QK^T of the sigma/mu cross-attention (reference
style_transformer.py:536-569; 6 400 problems of 49x49x32 at bs = 32) lives inside window_attn_pw_kernel<true>, fused with bias, mask,
softmax and two P.V products.  This tool builds a probe variant of the library (window_attn_pw_bf16.hip with -DPW_CLOCK_PROBE: s_memtime
stamps around the 8 QK^T MFMAs of every (window, head) and around QK^T + the 8 bias-through-identity MFMAs), runs the dual-stream
attention at the bs = 32 geometry and prints clocks per phase, the matrix-pipe utilisation inside the phase and the algorithmic rate while
in phase.  Run on the GPU box:  python profiles/tools/pw_probe.py > profiles/r04_pw_probe_mi355x.txt"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "mastermetastyletransfer_amd", "csrc")
LIB = os.path.join(ROOT, "profiles", "ab", "libmaster_hip_pwprobe.so")
os.makedirs(os.path.dirname(LIB), exist_ok=True)
obj = "/tmp/window_attn_pw_probe.o"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", f"-I{ROOT}/include", "-DPW_CLOCK_PROBE", "-c",
                os.path.join(CSRC, "window_attn_pw_bf16.hip"), "-o", obj], check=True)
others = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".o") and f != "window_attn_pw_bf16.o"]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, obj] + others, check=True)
os.environ["MASTER_HIP_LIB"] = LIB
import torch
from mastermetastyletransfer_amd import ops
from mastermetastyletransfer_amd.window_plan import window_plan

dev = torch.device("cuda:0")
lib = ops.load_library()
B, H, W, heads = 32, 32, 32, 8
T, C = H * W, heads * 32
plan = window_plan(H, W, [7, 7], [4, 4], dev)
g = torch.Generator().manual_seed(0)
t = lambda *s: (torch.randn(*s, generator=g)).to(dev).bfloat16()
q, k, v0, v1 = t(B * T, C), t(B * T, C), t(B * T, C), t(B * T, C)
kp, p0, p1 = t(B, C), t(C), t(C)
bias64 = torch.zeros(heads, 64, 64); bias64[:, :, 49:] = -1e30; bias64[:, :49, :49] = torch.randn(heads, 49, 49, generator=g) * 0.02
bias64 = bias64.to(dev)
grid = min(B * plan.nW * 2, 512)
buf = torch.zeros(grid * 4 * 6, dtype=torch.int64, device=dev)
lib.master_pw_probe_buffer.argtypes = [ctypes.c_void_p]
assert lib.master_pw_probe_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
run = lambda: ops.window_attn(q, k, v0, v1, kp, p0, p1, bias64, plan.src, plan.region, B, T, heads, plan.nW, plan.N, 32 ** -0.5)
for _ in range(20):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
d = buf.view(grid * 4, 6).cpu().double()
n = d[:, 2].sum()
qk, qkb = d[:, 0].sum() / n, d[:, 1].sum() / n
kern = d[:, 3].mean()
alg = 2.0 * 49 * 49 * 32                      # algorithmic flops of one (window, head) QK^T
pad = 2.0 * 64 * 64 * 32                      # flops of the padded tile the 8 MFMAs compute
ghz = float((d[:, 3] / d[:, 4].clamp_min(1)).median()) * 0.1      # shader clocks per 100-MHz tick over each wave's lifetime: the clock the chip held
print(f"sigma/mu cross-attention core, bs = {B}, {H}x{W} map, window 7 shift 4: {B * plan.nW * heads} (window, head) problems, dual value stream (window_attn_pw_kernel<true>, probe build)")
print(f"launch {us:.1f} us (probe build; stamps add a few %); waves {grid * 4}, (window, head) problems per wave {n / (grid * 4):.2f}; wave lifetime {kern:.0f} clocks -> in-kernel clock {ghz:.2f} GHz")
print(f"QK^T phase (8 x v_mfma_f32_32x32x16_bf16 back to back + result read): {qk:.1f} clocks per (window, head)  ->  matrix-pipe utilisation in phase {8 * 32 / qk:.3f} (8 MFMAs x 32 clocks / phase clocks)")
print(f"QK^T + bias-through-identity (16 MFMAs back to back): {qkb:.1f} clocks  ->  utilisation in phase {16 * 32 / qkb:.3f}")
per_simd_pad = pad / (qk / (ghz * 1e9)) / 1e12 if ghz else 0
print(f"rate while in the QK^T phase, per SIMD: padded 64x64x32 tile {per_simd_pad * 1e3:.1f} GFLOP/s, algorithmic 49x49x32 {per_simd_pad * alg / pad * 1e3:.1f} GFLOP/s; "
      f"x 1024 SIMDs: padded {per_simd_pad * 1024:.0f} TFLOP/s = {per_simd_pad * 1024 / 2500:.3f} of 2.5 PF, algorithmic {per_simd_pad * 1024 * alg / pad:.0f} TFLOP/s = {per_simd_pad * 1024 * alg / pad / 2500:.3f}")
print(f"share of the wave's lifetime spent in the QK^T phase: {qk * (n / (grid * 4)) / kern:.4f}; whole launch: QK^T algorithmic {B * plan.nW * heads * alg / (us * 1e-6) / 1e12:.2f} TFLOP/s = {B * plan.nW * heads * alg / (us * 1e-6) / 1e12 / 2500:.4f} of peak")
print("two waves share a SIMD (two workgroups per CU): a phase longer than 8 x 32 = 256 clocks is the partner wave's MFMAs / the issue of the 8 instructions, not idle pipe")
import json
print("JSON " + json.dumps({"qk_phase_clocks": round(qk, 1), "qk_phase_mfma_util": round(8 * 32 / qk, 4), "qk_plus_bias_phase_clocks": round(qkb, 1),
                            "in_kernel_clock_ghz": round(ghz, 3), "qk_padded_tflops_in_phase_chip": round(per_simd_pad * 1024, 1),
                            "qk_alg_tflops_in_phase_chip": round(per_simd_pad * 1024 * alg / pad, 1), "qk_alg_frac_of_bf16_peak_in_phase": round(per_simd_pad * 1024 * alg / pad / 2500, 4),
                            "share_of_wave_lifetime_in_qk_phase": round(qk * (n / (grid * 4)) / kern, 4), "launch_us_probe_build": round(us, 1)}))
