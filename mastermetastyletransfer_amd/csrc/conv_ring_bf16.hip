// 3x3 reflect-pad convolution (+ nearest x2 upsample in front, + ReLU) for Cin = 128, Cout = 64 / 128, bf16: the middle layers of the
// decoder (reference codes/decoder.py:26-37: Upsample, Conv(128,128) x3, Conv(128,64)), NHWC in and out.
//
// The implicit GEMM (gemm_bf16.hip) re-gathers every source pixel nine times through L2 and reaches ~550 TFLOP/s on these
// layers; the halo kernel for Cin = 64 (conv_halo_bf16.hip) keeps its weights in registers, which Cin = 128 cannot (288
// registers per 32 output channels).  Here the PIXELS are stationary in LDS and the weights stream:
//   workgroup = 8 x 32 output pixels x all output channels, 4 waves, wave w = output rows 2w, 2w+1 (two 32-pixel MFMA tiles);
//   the (8+2) x (32+2) source halo tile (reflect and upsample folded into the gather) is fetched ONCE by LDS-DMA: 340 pixel
//     rows of 256 B, position p of row i holds channel chunk p ^ (i & 15) (conflict-free B-fragment reads);
//   out^T[cout][pixel] = sum_k W[cout][k] X[pixel][k], k = (tap, channel): A = weights, streamed in 18 steps of
//     [Cout rows][64 k] (128-B rows, chunk p ^ ((row >> 1) & 7)) through a 4-slot LDS ring by LDS-DMA three steps ahead,
//     counted vmcnt + one barrier per step; B = pixel fragments read straight from the halo tile at (y+dy, x+dx);
//   per k-step 4 (2) weight fragments + 2 pixel fragments feed 8 (4) MFMAs;
//   epilogue: + bias, ReLU, bf16, transposed through the idle ring into whole 256-B (128-B) pixel rows.
#include "common.h"

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int CIN = 128, TH = 8, TW = 32, HW_ = TW + 2, HROWS = (TH + 2) * HW_;     // 340 halo pixels
constexpr int HALO_BYTES = HROWS * CIN * 2;                                            // 87 040
constexpr int RING = 4, AHEAD = 3, KSTEP = 64;                                         // ring step = [Cout][64 k]
constexpr int NSTEPS = 9 * CIN / KSTEP;                                                // 18

struct RingArgs {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* out;
    int B, Hin, Win, Hout, Wout, up, relu;
};

union Frag { u32x4 u; bf16x8_t b; };

#define LDS_READ128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define LGKM_WAIT(n) do { asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

template <int COUT>
__global__ __launch_bounds__(256, 1) void conv_ring_kernel(RingArgs p) {
    constexpr int NT = COUT / 32;                    // output-channel tiles
    constexpr int SLOTB = COUT * KSTEP * 2;          // 16 KB (Cout 128) / 8 KB (Cout 64)
    constexpr int PW = SLOTB / 1024 / 4;             // DMA pieces per wave per step: 4 / 2
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // halo tile | ring | bias[COUT]
    unsigned char* halo = lds;
    unsigned char* ring = lds + HALO_BYTES;
    float* bs = (float*)(ring + RING * SLOTB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int tiles_x = p.Wout / TW, tiles_y = p.Hout / TH;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; const int b = bid / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- weight stream: step t = k range [64 t, 64 t + 64) of every Cout row; piece I = 8 rows of 128 B ---------------------------
    const int prow0 = wave * (PW * 8) + (lane >> 3), l7 = lane & 7;
    auto dma_piece = [&](int t, int i) __attribute__((always_inline)) {
        const int row = prow0 + 8 * i;                                   // output channel
        const int q = l7 ^ ((row >> 1) & 7);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.w + (size_t)row * (9 * CIN) + t * KSTEP + q * 8),
                                         (__attribute__((address_space(3))) void*)(ring + (t % RING) * SLOTB + (wave * PW + i) * 1024), 16, 0, 0);
    };
    int t_issue = 0;
    for (; t_issue < AHEAD; ++t_issue)
#pragma unroll
        for (int i = 0; i < PW; ++i) dma_piece(t_issue, i);

    // ---- halo gather: 340 pixel rows of 256 B = 85 pieces of 4 rows; source pixel = reflect at output resolution, then >> up -------
    for (int I = wave; I < HROWS / 4; I += 4) {
        const int hp = 4 * I + (lane >> 4);                              // halo pixel index = ly * 34 + lx
        const int ly = hp / HW_, lx = hp - ly * HW_;
        const int sy = reflect1(y0 + ly - 1, p.Hout) >> p.up, sx = reflect1(x0 + lx - 1, p.Wout) >> p.up;
        const int q = (lane & 15) ^ (hp & 15);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.x + (((size_t)b * p.Hin + sy) * p.Win + sx) * CIN + q * 8),
                                         (__attribute__((address_space(3))) void*)(halo + I * 1024), 16, 0, 0);
    }
    if (tid < COUT) bs[tid] = p.bias ? p.bias[tid] : 0.f;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                        // halo tile, bias and the first ring steps have landed

    auto lds_off = [&](const unsigned char* ptr) { return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)ptr; };
    const unsigned halo0 = lds_off(halo), ring0 = lds_off(ring);
    const int xsw = (h ^ ((r >> 1) & 7)) << 4;                           // weight rows (128 B): chunk (2s+h) ^ ((row>>1)&7)

    f32x16 acc[2][NT];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[tt][nt][e] = 0.f;

    int t_cons = 0;
    for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap - 3 * dy;                       // halo offsets 0..2 (= tap offsets -1..1, the halo starts at -1)
        // halo pixel of this lane's output pixel for this tap, per token tile tt (output row 2 wave + tt)
        unsigned pix[2]; int psw[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int hp = (2 * wave + tt + dy) * HW_ + r + dx;
            pix[tt] = halo0 + hp * (CIN * 2); psw[tt] = hp & 15;
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {                           // two ring steps per tap: channels [64 half, 64 half + 64)
            // my DMA pieces of step t_cons have landed -> barrier -> everyone's have, and everyone left the slot step t_cons+3 goes into
            {
                const int ahead = t_issue - 1 - t_cons;
                if (PW == 4) { if (ahead >= 2) WAIT_VMCNT(8); else if (ahead == 1) WAIT_VMCNT(4); else WAIT_VMCNT(0); }
                else { if (ahead >= 2) WAIT_VMCNT(4); else if (ahead == 1) WAIT_VMCNT(2); else WAIT_VMCNT(0); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            const int t_dma = t_issue < NSTEPS ? t_issue++ : -1;
            const unsigned wb = ring0 + (t_cons % RING) * SLOTB + r * 128;
            ++t_cons;
            u32x4 a[2][NT], xb[2][2];
            auto rd = [&](int buf, int s) {                              // k-step s of this ring step: channels 64 half + 16 s ..
                const unsigned ad = wb + ((32 * s) ^ xsw);
                LDS_READ128(a[buf][0], ad, 0); LDS_READ128(a[buf][1], ad, 4096);
                if (NT == 4) { LDS_READ128(a[buf][NT - 2], ad, 8192); LDS_READ128(a[buf][NT - 1], ad, 12288); }
                const int c = 8 * half + 2 * s + h;                      // 16-B channel chunk within the 256-B pixel row
                LDS_READ128(xb[buf][0], pix[0] + ((c ^ psw[0]) << 4), 0);
                LDS_READ128(xb[buf][1], pix[1] + ((c ^ psw[1]) << 4), 0);
            };
            rd(0, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s + 1 < 4) { rd((s + 1) & 1, s + 1); if (NT == 4) LGKM_WAIT(6); else LGKM_WAIT(4); } else { LGKM_WAIT(0); }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    Frag f, x0f, x1f; f.u = a[s & 1][nt]; x0f.u = xb[s & 1][0]; x1f.u = xb[s & 1][1];
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b, x0f.b, acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b, x1f.b, acc[1][nt], 0, 0, 0);
                    if (nt == 0 && t_dma >= 0 && s < PW) { __builtin_amdgcn_sched_barrier(0); dma_piece(t_dma, s); __builtin_amdgcn_sched_barrier(0); }
                }
            }
        }
    }

    // ---- epilogue: + bias, ReLU, bf16; each wave transposes its [64 pixels][COUT] tile through the idle ring (16-B chunks swizzled
    //      by pixel & 15) and stores whole pixel rows (COUT * 2 bytes) ------------------------------------------------------------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                        // every wave is done with the ring
    {
        constexpr int ROWB = COUT * 2, CH = ROWB / 16;                   // bytes and 16-B chunks per pixel row (16 / 8)
        unsigned char* stage = ring + wave * (64 * ROWB);               // 16 KB / 8 KB per wave: the ring is 64 KB / 32 KB
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int px = 32 * tt + r, ch0 = nt * 32 + 8 * g + 4 * h;      // accumulator register e = channel (e&3) + 8(e>>2) + 4h
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = acc[tt][nt][4 * g + e] + bs[ch0 + e]; if (p.relu) v[e] = fmaxf(v[e], 0.f); }
                    u32x2 o; o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]);
                    const int c = ch0 >> 3;                                          // 16-B chunk, 8-B half (ch0 & 4)
                    *(u32x2*)(stage + px * ROWB + (((c ^ (px & (CH - 1))) << 4)) + (ch0 & 4) * 2) = o;
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // wave-private tile
        constexpr int PPI = 64 / CH;                                     // pixels per wave instruction (4 / 8)
        const int q = lane % CH, psub = lane / CH;
#pragma unroll
        for (int it = 0; it < 64 / PPI; ++it) {
            const int px = it * PPI + psub;                              // 0..63: row 2 wave + (px >> 5), column px & 31
            const u32x4 v = *(const u32x4*)(stage + px * ROWB + ((q ^ (px & (CH - 1))) << 4));
            const int oy = y0 + 2 * wave + (px >> 5), ox = x0 + (px & 31);
            *(u32x4*)(p.out + (((size_t)b * p.Hout + oy) * p.Wout + ox) * COUT + q * 8) = v;
        }
    }
}

template <int COUT>
int launch(const RingArgs& p, hipStream_t s) {
    const int lds = HALO_BYTES + RING * COUT * KSTEP * 2 + COUT * 4;
    static int attr_set = 0;
    if (attr_set < lds) {
        (void)hipFuncSetAttribute((const void*)conv_ring_kernel<COUT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = lds;
    }
    hipLaunchKernelGGL((conv_ring_kernel<COUT>), dim3(p.B * (p.Hout / TH) * (p.Wout / TW)), dim3(256), lds, s, p);
    return master_check_launch("conv_ring_bf16");
}

}  // namespace

// MASTER_ERR_UNSUPPORTED outside Cin = 128, Cout in {64, 128}, Hout % 8 == 0, Wout % 32 == 0 (the caller then uses the implicit GEMM)
int conv_ring_bf16(const bf16_t* x, const bf16_t* w, const float* bias, bf16_t* out, int B, int Hin, int Win, int Cin, int Cout, int up,
                   int relu, hipStream_t s) {
    const int Hout = Hin << up, Wout = Win << up;
    if (Cin != CIN || (Cout != 64 && Cout != 128) || Hout % TH != 0 || Wout % TW != 0 || Hout < 2 || Wout < 2) return MASTER_ERR_UNSUPPORTED;
    RingArgs p{x, w, bias, out, B, Hin, Win, Hout, Wout, up, relu};
    return Cout == 128 ? launch<128>(p, s) : launch<64>(p, s);
}
