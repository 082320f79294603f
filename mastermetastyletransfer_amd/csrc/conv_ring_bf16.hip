// 3x3 reflect-pad convolution (+ nearest x2 upsample in front, + ReLU), bf16, NHWC in and out, PIXEL-stationary: the decoder layers
// the other conv kernels serve badly (reference codes/decoder.py:24-46):
//     Cin 256 -> 128 (first layer, 32x32 maps)      tile  4 x 32, one pixel row per wave
//     Cin 128 -> 128 / 64 (Upsample?, 64x64 maps)   tile  8 x 32, two rows per wave
//     Cin  64 -> 32 (128x128 maps)                  tile  8 x 32, two rows per wave, TWO workgroups per CU (r03; was 16 x 32, one per CU)
//
// The implicit GEMM (gemm_bf16.hip) re-gathers every source pixel nine times through L2 (~350-550 TFLOP/s on these layers); the halo
// kernel for Cin = 64 (conv_halo_bf16.hip) keeps its weights in registers, which Cin >= 128 cannot, and with Cout = 32 it has one
// MFMA per LDS read.  Here the PIXELS are stationary in LDS and the weights stream:
//   workgroup = TH x 32 output pixels x all output channels, 4 waves, wave w = output rows RPW w .. RPW w + RPW-1 (RPW 32-pixel MFMA tiles);
//   the (TH+2) x (32+2) source halo tile (reflect and upsample folded into the gather) is fetched ONCE by LDS-DMA, one pixel = one
//     row of 2 Cin bytes, 16-B channel chunks swizzled (p ^ (pixel & 15), or p ^ ((pixel >> 1) & 7) for 128-B rows): conflict-free
//     B-fragment reads;
//   out^T[cout][pixel] = sum_k W[cout][k] X[pixel][k], k = (tap, channel): A = weights, streamed in 9 Cin/64 steps of [Cout rows][64 k]
//     (128-B rows, chunk p ^ ((row >> 1) & 7)) through an LDS ring by LDS-DMA, counted vmcnt + one barrier per step; B = pixel
//     fragments read straight from the halo tile at (y+dy, x+dx);
//   per k-step Cout/32 weight fragments + RPW pixel fragments feed RPW Cout/32 MFMAs;
//   epilogue: + bias, ReLU, bf16, transposed through the (now idle) halo region into whole pixel rows of 2 Cout bytes.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int TW = 32, HW_ = TW + 2, KSTEP = 64;                                       // tile width, halo width, k per ring step
template <int CIN, int RPW> struct Geo {
    static constexpr int TH = 4 * RPW, HROWS = (TH + 2) * HW_;                          // halo pixels: 204 / 340 / 612
    static constexpr int RPP = 1024 / (CIN * 2);                                        // pixel rows per one-KB DMA piece: 2 / 4 / 8
    static constexpr int HPIECES = (HROWS + RPP - 1) / RPP;
    static constexpr int HALO_BYTES = HPIECES * 1024;                                   // 104 448 / 87 040 / 78 848
    static constexpr int RING = CIN == 256 ? 3 : 4, AHEAD = RING - 1;                   // the 104-KB halo of Cin 256 leaves room for 3 slots
    static constexpr int NSTEPS = 9 * CIN / KSTEP;
};
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); __builtin_amdgcn_sched_barrier(0); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

struct RingArgs {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* out;
    int B, Hin, Win, Hout, Wout, up, relu;
};

union Frag { u32x4 u; bf16x8_t b; };

#define LDS_READ128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#ifdef CONV_CLOCK_PROBE   // diagnostic build only (profiles/probes/conv_ring_probe.hip): s_memtime sections of wave 0
#define CPROBE_MARK() do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); probe_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define CPROBE_ADD(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); probe_sec[i] += t_ - probe_last; probe_last = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CPROBE_MARK() do {} while (0)
#define CPROBE_ADD(i) do {} while (0)
#endif

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// workgroups per CU: the 64 -> 32 layer's 8-row tile needs 60 KB of LDS and 32 accumulator registers - two of them fit and cover each
// other's prologue / epilogue (r03, profiles/r03_conv_ring_probe_mi355x.txt: 47 % of that layer's workgroup life; 47 -> 30 us at bs=32).
// The same trade for 128 -> 64 (4-row tile, 3-slot ring: 76 KB) measured neutral - twice the weight stream per pixel - and is not built.
template <int CIN, int COUT, int RPW> constexpr int ring_occupancy() { return (CIN == 64 && RPW <= 2) ? 2 : 1; }

template <int CIN, int COUT, int RPW>
__global__ __launch_bounds__(256, (ring_occupancy<CIN, COUT, RPW>())) void conv_ring_kernel(RingArgs p) {
    typedef Geo<CIN, RPW> G;
#ifdef CONV_CLOCK_PROBE
    unsigned long long probe_sec[4] = {0, 0, 0, 0}, probe_last = 0;
    const unsigned long long probe_t0 = __builtin_amdgcn_s_memtime();
    CPROBE_MARK();
#endif
    constexpr int TH = G::TH, HROWS = G::HROWS, HALO_BYTES = G::HALO_BYTES, RING = G::RING, AHEAD = G::AHEAD, NSTEPS = G::NSTEPS;
    constexpr int NT = COUT / 32;                    // output-channel tiles
    constexpr int SLOTB = COUT * KSTEP * 2;          // 16 / 8 / 4 KB
    constexpr int PW = SLOTB / 1024 / 4;             // DMA pieces per wave per step: 4 / 2 / 1
    constexpr int CPR = CIN / 8;                     // 16-B channel chunks per pixel row
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
    for (; t_issue < AHEAD && t_issue < NSTEPS; ++t_issue)
#pragma unroll
        for (int i = 0; i < PW; ++i) dma_piece(t_issue, i);

    // ---- halo gather: one-KB pieces of RPP pixel rows; source pixel = reflect at output resolution, then >> up ----------------
    auto pix_swz = [&](int hp) { return CIN == 64 ? (hp >> 1) & 7 : hp & 15; };      // XOR term of a pixel row's chunk positions
    for (int I = wave; I < G::HPIECES; I += 4) {
        int hp = G::RPP * I + lane / CPR;                                // halo pixel index = ly * 34 + lx
        const int hps = hp;                                              // (the last piece of the 16-row tile runs past the tile: any pixel will do)
        hp = hp < HROWS ? hp : HROWS - 1;
        const int ly = hp / HW_, lx = hp - ly * HW_;
        const int sy = reflect1(y0 + ly - 1, p.Hout) >> p.up, sx = reflect1(x0 + lx - 1, p.Wout) >> p.up;
        const int q = (lane % CPR) ^ pix_swz(hps);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.x + (((size_t)b * p.Hin + sy) * p.Win + sx) * CIN + q * 8),
                                         (__attribute__((address_space(3))) void*)(halo + I * 1024), 16, 0, 0);
    }
    if (tid < COUT) bs[tid] = p.bias ? p.bias[tid] : 0.f;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                        // halo tile, bias and the first ring steps have landed
    CPROBE_ADD(0);                                                       // prologue: halo gather + first ring steps

    auto lds_off = [&](const unsigned char* ptr) { return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)ptr; };
    const unsigned halo0 = lds_off(halo), ring0 = lds_off(ring);
    const int xsw = (h ^ ((r >> 1) & 7)) << 4;                           // weight rows (128 B): chunk (2s+h) ^ ((row>>1)&7)

    f32x16 acc[RPW][NT];
#pragma unroll
    for (int tt = 0; tt < RPW; ++tt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[tt][nt][e] = 0.f;

    // ---- main loop: NSTEPS ring steps of 64 k (tap-major: step t = tap t / HALVES, channels 64 (t % HALVES) ..), four 16-k sub-steps each.
    //      r03: the wait + barrier for step t + 1 sits in front of the LAST sub-step of step t (whose fragments are in registers by then),
    //      and the first fragments of step t + 1 are read under that sub-step's MFMAs: the exposed LDS latency at every step boundary
    //      (profiles/r03_conv_ring_probe_mi355x.txt: the step ran 1 524 cycles for 1 024 of MFMAs) is gone. ---------------------------------
    constexpr int HALVES = CIN / KSTEP;
    constexpr int LATE = PW == 4 ? 1 : 0;                                // DMA pieces of the current step not yet issued when the early head runs
    unsigned pix[RPW]; int psw[RPW];
    unsigned wb = 0; int half = 0;
    auto set_step = [&](int t) __attribute__((always_inline)) {
        const int tap = t / HALVES;
        half = t - tap * HALVES;
        const int dy = tap / 3, dx = tap - 3 * dy;                       // halo offsets 0..2 (= tap offsets -1..1, the halo starts at -1)
#pragma unroll
        for (int tt = 0; tt < RPW; ++tt) {                               // halo pixel of this lane's output pixel for this tap, per token tile
            const int hp = (RPW * wave + tt + dy) * HW_ + r + dx;
            pix[tt] = halo0 + hp * (CIN * 2); psw[tt] = pix_swz(hp);
        }
        wb = ring0 + (t % RING) * SLOTB + r * 128;
    };
    auto head = [&](int t, bool early) __attribute__((always_inline)) {
        // my DMA pieces of step t have landed -> barrier -> everyone's have, and everyone left the slot the next DMA step goes into.
        // vmcnt counts in order: "at most n outstanding" with n = the pieces issued AFTER step t's.
        const int ahead = t_issue - 1 - t;                               // steps issued beyond step t (<= AHEAD - 1)
        if (early && LATE) { if (ahead >= 2) wait_vm<2 * PW - LATE>(); else if (ahead == 1) wait_vm<PW - LATE>(); else wait_vm<0>(); }
        else { if (ahead >= 2) wait_vm<2 * PW>(); else if (ahead == 1) wait_vm<PW>(); else wait_vm<0>(); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    u32x4 a[2][NT], xb[2][RPW];
    auto rd = [&](int buf, int s) __attribute__((always_inline)) {       // k-step s of the current ring step: channels 64 half + 16 s ..
        const unsigned ad = wb + ((32 * s) ^ xsw);
        LDS_READ128(a[buf][0], ad, 0);
        if (NT >= 2) LDS_READ128(a[buf][NT >= 2 ? 1 : 0], ad, 4096);
        if (NT == 4) { LDS_READ128(a[buf][NT - 2], ad, 8192); LDS_READ128(a[buf][NT - 1], ad, 12288); }
        const int c = 8 * half + 2 * s + h;                              // 16-B channel chunk within the pixel row
#pragma unroll
        for (int tt = 0; tt < RPW; ++tt) LDS_READ128(xb[buf][tt], pix[tt] + ((c ^ psw[tt]) << 4), 0);
    };
    head(0, false);
    CPROBE_ADD(1);
    int t_dma = t_issue < NSTEPS ? t_issue++ : -1;                       // the step whose pieces go out during step 0
    set_step(0);
    rd(0, 0);
    for (int t = 0; t < NSTEPS; ++t) {
        int t_dma_next = -1;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) { rd((s + 1) & 1, s + 1); wait_lgkm<NT + RPW>(); }
            else {
                wait_lgkm<0>();                                          // the last sub-step's fragments are in registers: this step is done with LDS
                if (t + 1 < NSTEPS) {
                    CPROBE_ADD(2);
                    head(t + 1, true);
                    CPROBE_ADD(1);                                       // step head: weights landed + barrier
                    t_dma_next = t_issue < NSTEPS ? t_issue++ : -1;
                    set_step(t + 1);
                    rd(0, 0);                                            // buffer 0 is free (sub-step 3 computes from buffer 1)
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                Frag f; f.u = a[s & 1][nt];
#pragma unroll
                for (int tt = 0; tt < RPW; ++tt) {
                    Frag xf; xf.u = xb[s & 1][tt];
                    acc[tt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b, xf.b, acc[tt][nt], 0, 0, 0);
                }
                if (nt == 0 && t_dma >= 0 && s < PW) { __builtin_amdgcn_sched_barrier(0); dma_piece(t_dma, s); __builtin_amdgcn_sched_barrier(0); }
            }
        }
        t_dma = t_dma_next;
    }
    CPROBE_ADD(2);                                                       // fragment reads + MFMAs + DMA issue

    // ---- epilogue: + bias, ReLU, bf16; each wave transposes its [32 RPW pixels][COUT] tile through the idle halo region (16-B chunks
    //      swizzled by the pixel index) and stores whole pixel rows (COUT * 2 bytes) -----------------------------------------------------
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                        // every wave is done with the halo tile and the ring
    {
        constexpr int ROWB = COUT * 2, CH = ROWB / 16, NPX = 32 * RPW;   // bytes / 16-B chunks per pixel row (16 / 8 / 4); pixels per wave
        unsigned char* stage = halo + wave * (NPX * ROWB);              // <= 16 KB per wave
#pragma unroll
        for (int tt = 0; tt < RPW; ++tt)
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
        constexpr int PPI = 64 / CH;                                     // pixels per wave instruction (4 / 8 / 16)
        const int q = lane % CH, psub = lane / CH;
#pragma unroll
        for (int it = 0; it < NPX / PPI; ++it) {
            const int px = it * PPI + psub;                              // row RPW wave + (px >> 5), column px & 31
            const u32x4 v = *(const u32x4*)(stage + px * ROWB + ((q ^ (px & (CH - 1))) << 4));
            const int oy = y0 + RPW * wave + (px >> 5), ox = x0 + (px & 31);
            *(u32x4*)(p.out + (((size_t)b * p.Hout + oy) * p.Wout + ox) * COUT + q * 8) = v;
        }
    }
#ifdef CONV_CLOCK_PROBE
    CPROBE_ADD(3);                                                       // epilogue (stores issued)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && g_probe) {
        g_probe[blockIdx.x * 8] = __builtin_amdgcn_s_memtime() - probe_t0;
        for (int i = 0; i < 4; ++i) g_probe[blockIdx.x * 8 + 1 + i] = probe_sec[i];
        g_probe[blockIdx.x * 8 + 5] = probe_t0;
    }
#endif
}

template <int CIN, int COUT, int RPW>
int launch(const RingArgs& p, hipStream_t s) {
    typedef Geo<CIN, RPW> G;
    const int lds = G::HALO_BYTES + G::RING * COUT * KSTEP * 2 + COUT * 4;
    static LdsLimit lds_limit;
    lds_limit.ensure((const void*)conv_ring_kernel<CIN, COUT, RPW>, (int)(lds));
    hipLaunchKernelGGL((conv_ring_kernel<CIN, COUT, RPW>), dim3(p.B * (p.Hout / G::TH) * (p.Wout / TW)), dim3(256), lds, s, p);
    return master_check_launch("conv_ring_bf16");
}

}  // namespace

// MASTER_ERR_UNSUPPORTED outside (Cin, Cout) in {(256,128), (128,128), (128,64), (64,32)} with the output a multiple of the tile
// (4 / 8 / 8 / 16 rows x 32 columns); the caller then uses the other conv kernels
int conv_ring_bf16(const bf16_t* x, const bf16_t* w, const float* bias, bf16_t* out, int B, int Hin, int Win, int Cin, int Cout, int up,
                   int relu, hipStream_t s) {
    const int Hout = Hin << up, Wout = Win << up;
    if (Wout % TW != 0 || Hout < 2) return MASTER_ERR_UNSUPPORTED;
    RingArgs p{x, w, bias, out, B, Hin, Win, Hout, Wout, up, relu};
    if (Cin == 256 && Cout == 128 && Hout % 4 == 0) return launch<256, 128, 1>(p, s);
    if (Cin == 128 && Cout == 128 && Hout % 8 == 0) return launch<128, 128, 2>(p, s);
    if (Cin == 128 && Cout == 64 && Hout % 8 == 0) return launch<128, 64, 2>(p, s);
    if (Cin == 64 && Cout == 32) {
        static int tile8 = -1;
        if (tile8 < 0) { const char* e = getenv("MASTER_CONV_RING_64_TILE8"); tile8 = e ? atoi(e) : 1; }
        if (tile8 && Hout % 8 == 0) return launch<64, 32, 2>(p, s);
        if (Hout % 16 == 0) return launch<64, 32, 4>(p, s);
    }
    return MASTER_ERR_UNSUPPORTED;
}
