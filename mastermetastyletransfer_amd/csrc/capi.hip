// extern "C" surface of libmaster_hip.so (include/master_hip.h): argument checks, dtype dispatch,
// error strings and the optional per-launch event timing used by bench.py.
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <vector>
#include "common.h"
#include <cstring>
#include <cstdlib>

// ---- kernel launchers defined in the other translation units --------------------------------
int gemm_f32_linear(const float*, const float*, const float*, const float*, const float*, float*, int, int, int, int,
                    int, int, int, hipStream_t, float* aux = nullptr, const float* rowscale = nullptr, int rs_rows = 1);
// ---- train step (gemm_f32.hip, gemm_f32_t.hip, norms_bwd.hip, window_attn_bwd_f32.hip, train_misc.hip) ----
int window_attn_proj_bf16(const bf16_t*, const bf16_t*, const bf16_t*, int, int, int, const bf16_t*, const bf16_t*, int, int, const float*,
                          const int32_t*, const uint8_t*, const bf16_t*, const float*, const bf16_t*, int, bf16_t*, int, int, int, int, int,
                          int, float, hipStream_t);
int linear_ts_group_bf16(int, const bf16_t* const*, const bf16_t* const*, const float* const*, bf16_t* const*, const int*, const int*, const int*,
                         const float*, const float*, int, int, float, hipStream_t);
int gemm_f32_conv3x3_dgrad(const float*, const float*, float*, const float*, const float*, int, int, int, int, int, int, hipStream_t);
size_t gemm_f32_t_workspace(int, int);
int gemm_f32_dgrad(const float*, const float*, float*, const float*, const float*, int, int, int, int, int, int, int, hipStream_t, const float* rowscale = nullptr,
                   int rs_rows = 1);
int gemm_f32_wgrad(const float*, const float*, float*, float*, int, int, int, int, int, float*, size_t, hipStream_t, int accumulate = 0,
                   const float* rowscale = nullptr, int rs_rows = 1);
int gemm_f32_conv3x3_wgrad(const float*, const float*, float*, float*, int, int, int, int, int, int, int, int, float*, size_t, hipStream_t);
int reduce_slabs_launch(const float*, float*, size_t, size_t, int, hipStream_t, int accumulate = 0);
int reduce_segments_launch(const float* const*, float* const*, const size_t*, const size_t*, const int*, const int*, int, hipStream_t);
int colsum_slabs(int, int);
int colsum_launch(const float*, const float*, float*, float*, int, int, int, hipStream_t, int accumulate = 0);
template <typename T> int channel_affine_launch(const void*, const float*, const float*, void*, int, int, hipStream_t);
template <typename T> int softmax_rows_launch(const void*, void*, int, int, int, float, hipStream_t);
int softmax_rows_bwd_launch(const float*, const float*, float*, int, int, int, float, hipStream_t);
int relu_mask_launch(const float*, const float*, float*, int, int, int, int, int, hipStream_t);
int fold_reflect_launch(const float*, float*, int, int, int, int, int, hipStream_t);
template <typename T> int maxpool2_bwd_launch(const void*, const void*, void*, int, int, int, int, hipStream_t);
template <typename T> int nhwc_to_image_launch(const void*, float*, int, int, int, hipStream_t);
int gemm_bf16_conv3x3_dgrad(const bf16_t*, const bf16_t*, bf16_t*, const bf16_t*, const bf16_t*, int, int, int, int, int, hipStream_t);
int rowscale_add_launch(const float*, const float*, const float*, float*, int, int, int, hipStream_t);
int mul_add_launch(const float*, const float*, const float*, float*, size_t, hipStream_t);
int mul_add_bwd_launch(const float*, const float*, const float*, float*, float*, size_t, hipStream_t);
int adam_chunk();
int adam_launch(const void*, int, long long, float, float, float, float, float, int, hipStream_t);
int lerp_launch(const void*, int, long long, float, hipStream_t);
int adam_dev_launch(const void*, int, long long, const float*, int*, hipStream_t);
int loss_tap_slices(int);
int loss_terms_launch(const float*, const float*, const float*, int, float*, float*, float*, int, int, int, int, int, hipStream_t);
int conv_weight_pack_launch(const float*, float*, float*, int, int, int, hipStream_t);
int conv_cout3_dgrad_launch(const float*, const float*, float*, int, int, int, hipStream_t);
size_t conv_cout3_wgrad_workspace(int, int, int);
int conv_cout3_wgrad_launch(const float*, const float*, float*, float*, float*, int, int, int, hipStream_t);
int bias_tile_fwd_launch(const float*, const long long*, float*, int, int, hipStream_t);
int bias_tile_bwd_launch(const float*, const long long*, float*, int, int, int, hipStream_t);
template <typename T> int loss_tap_bwd_launch(const void*, const float*, const float*, const float*, const float*, float*, void*, int, int, int,
                                              float, float, const float*, const float*, int, int, hipStream_t);
int layernorm_bwd_blocks(int);
int layernorm_bwd_launch(const float*, const float*, const float*, float*, float*, float*, int, int, float, hipStream_t, int accumulate = 0, const float* add = nullptr);
int instnorm_bwd_launch(const float*, const float*, const float*, const float*, float*, float*, int, int, int, int, float, hipStream_t);
int window_attn_bwd_f32(const float*, const float*, const float*, const float*, int, int, int, const float*, const float*, const float*, int,
                        int, const float*, const int32_t*, const uint8_t*, const int32_t*, const float*, const float*, int, float*, float*,
                        float*, float*, int, int, int, float*, float*, float*, float*, int, int, int, int, int, int, float, hipStream_t);
int gemm_f32_conv3x3(const float*, const float*, const float*, float*, int, int, int, int, int, int, int, int,
                     hipStream_t, int zero_pad = 0);
template <typename T> int vgg_conv1_launch(const float*, const float*, const float*, void*, int, int, int, int, hipStream_t);
template <typename T> int image_to_nhwc_launch(const float*, void*, int, int, int, int, hipStream_t);
template <typename T> int maxpool2_launch(const void*, void*, int, int, int, int, hipStream_t);
template <typename T> int chan_stats_launch(const void*, float*, float*, float*, int, int, int, hipStream_t);
int stats_slices(int HW);
template <typename T> int in_dist_launch(const void*, const float*, const float*, float*, int, int, int, int, float, int, hipStream_t);
int window_attn_f32(const float*, const float*, const float*, const float*, int, int, int, const float*, const float*,
                    const float*, int, int, const float*, const int32_t*, const uint8_t*, float*, float*, int, int,
                    int, int, int, int, float, hipStream_t, int x3 = 0);
int gemm_bf16_linear(const bf16_t*, const bf16_t*, const float*, const bf16_t*, const bf16_t*, bf16_t*, int, int, int,
                     int, int, int, int, hipStream_t);
int gemm_bf16_conv3x3(const bf16_t*, const bf16_t*, const float*, void*, int, int, int, int, int, int, int, int,
                      hipStream_t, int zero_pad = 0);
int window_attn_pw_bf16(const bf16_t*, const bf16_t*, const bf16_t*, const bf16_t*, int, int, int, const bf16_t*,
                        const bf16_t*, const bf16_t*, int, int, const float*, const int32_t*, const uint8_t*, bf16_t*,
                        bf16_t*, int, int, int, int, int, int, float, hipStream_t);
int window_attn_bf16(const bf16_t*, const bf16_t*, const bf16_t*, const bf16_t*, int, int, int, const bf16_t*,
                     const bf16_t*, const bf16_t*, int, int, const float*, const int32_t*, const uint8_t*, bf16_t*,
                     bf16_t*, int, int, int, int, int, int, float, hipStream_t);
template <typename T> int layernorm_launch(const void*, const void*, const void*, void*, int, int, float, int, int, int, int, hipStream_t);
template <typename T> int instnorm_launch(const void*, void*, const void*, void*, int, int, int, int, float, int, hipStream_t);
template <typename T> int patch_embed_launch(const float*, const void*, const void*, const void*, const void*, void*, int, int, int, int, float, hipStream_t);
int convert_launch(const void*, void*, size_t, int, int, hipStream_t);
int linear_ts_bf16(const bf16_t*, const float*, const float*, const bf16_t*, const float*, const bf16_t*, const bf16_t*, bf16_t*,
                   int, int, int, int, int, int, float, hipStream_t, int mH = 0, int mW = 0, int mC = 0);
int attn_block_bf16(const bf16_t*, const float*, const float*, const bf16_t*, const float*, const bf16_t*, const float*, const float*,
                    const int32_t*, const uint8_t*, bf16_t*, int, int, int, int, int, int, float, float, hipStream_t);
int patch_embed_bf16(const float*, const float*, const float*, const float*, const float*, bf16_t*, int, int, int, float, hipStream_t);
int conv_ring_bf16(const bf16_t*, const bf16_t*, const float*, bf16_t*, int, int, int, int, int, int, int, hipStream_t);
int conv_halo_bf16(const bf16_t*, const bf16_t*, const float*, bf16_t*, int, int, int, int, int, int, int, hipStream_t);
int conv_tail_bf16(const bf16_t*, const bf16_t*, const float*, const bf16_t*, const float*, float*, int, int, int, hipStream_t);
int mlp_bf16(const bf16_t*, const float*, const float*, const bf16_t*, const float*, const bf16_t*, const float*,
             const bf16_t*, bf16_t*, int, int, int, int, int, int, float, hipStream_t);
int resize_plan_host(int, int, int32_t*, int32_t*, int*);
int preprocess_launch(const uint8_t*, int, int, int, float*, int, int, const int32_t*, const int32_t*, int, const int32_t*, const int32_t*, int,
                      uint8_t*, const float*, const float*, hipStream_t);
int postprocess_launch(const float*, uint8_t*, int, int, int, hipStream_t);
int mlp_c128_bf16(const bf16_t*, const float*, const float*, const bf16_t*, const float*, const bf16_t*, const float*,
                  const bf16_t*, bf16_t*, int, int, int, int, int, int, float, hipStream_t);
int mlp_2w_bf16(const bf16_t*, const float*, const float*, const bf16_t*, const float*, const bf16_t*, const float*,
                const bf16_t*, bf16_t*, int, int, int, int, int, int, float, hipStream_t);
int mlp_8w_bf16(const bf16_t*, const float*, const float*, const bf16_t*, const float*, const bf16_t*, const float*,
                const bf16_t*, bf16_t*, int, int, int, int, int, int, float, hipStream_t);

int gemm_x3_split_weights(const float*, void*, size_t, size_t, int, hipStream_t, int fmt);
int conv_tail_x3(const void*, int, const void*, const float*, const void*, const float*, float*, int, int, int, hipStream_t, int fmt);
int mlp_x3(const float*, const float*, const float*, const void*, const float*, const void*, const float*, const float*, float*, int, int, int, int, int, int,
           float, hipStream_t, int fmt);
int gemm_x3_linear(const float*, const void*, const float*, const float*, const float*, float*, int, int, int, int, int, int, int, hipStream_t,
                   int a_split, int out_split, int fmt);
int gemm_x3_conv3x3(const float*, const void*, const float*, float*, int, int, int, int, int, int, int, int, hipStream_t, int zero_pad, int a_split,
                    int out_split, int fmt);
int gemm_x3_conv3x3_dgrad(const float*, const void*, float*, const float*, const float*, int, int, int, int, int, hipStream_t);

// the two split-operand parity modes: bf16 halves (MASTER_F32X3) and IEEE-half halves (MASTER_F16X3); the kernels' format code (common.h)
static inline bool is_x3(int dtype) { return dtype == MASTER_F32X3 || dtype == MASTER_F16X3; }
static inline int x3_fmt(int dtype) { return dtype == MASTER_F16X3 ? X3_F16 : X3_BF16; }

// ---- error string -------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void master_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
int master_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { master_set_error("%s: launch failed: %s", what, hipGetErrorString(e)); return MASTER_ERR_LAUNCH; }
    return MASTER_OK;
}

// ---- profiling ----------------------------------------------------------------------------------
namespace {
std::mutex g_prof_mu;
bool g_prof_on = false;
struct ProfRec { int family; hipEvent_t e0, e1; };
std::vector<ProfRec> g_prof;
int64_t g_launches[MASTER_K_COUNT];
double g_ms[MASTER_K_COUNT];
}  // namespace

ProfScope::ProfScope(int f, hipStream_t st) : family(f), s(st), e0(nullptr), e1(nullptr), on(g_prof_on) {
    if (!on) return;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
}
ProfScope::~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e1, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back({family, e0, e1});
}

static void prof_drain() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) {
        (void)hipEventSynchronize(r.e1);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { g_launches[r.family]++; g_ms[r.family] += ms; }
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
}

#define REQUIRE(cond, ...) do { if (!(cond)) { master_set_error(__VA_ARGS__); return MASTER_ERR_INVALID; } } while (0)
static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" {

int master_version(void) { return 100; }
const char* master_last_error_string(void) { return g_err; }

// ================================================ train step ==========================================================================
#define F32_ONLY(what) do { if (dtype != MASTER_F32) { master_set_error(what ": the train step runs in the fp32 mode"); return MASTER_ERR_UNSUPPORTED; } } while (0)

int master_linear_act_fwd(const void* a, const void* w, const void* bias, void* out, void* aux, int M, int N, int K, int lda, int ldo,
                          int act, int dtype, master_stream_t stream) {
    F32_ONLY("linear_act_fwd");
    REQUIRE(a && w && out, "linear_act_fwd: null operand");
    REQUIRE(M > 0 && N > 0 && K > 0 && K % 32 == 0 && lda >= K && ldo >= N, "linear_act_fwd: bad dims M=%d N=%d K=%d", M, N, K);
    REQUIRE(al16(a) && al16(w) && lda % 4 == 0, "linear_act_fwd: a/w must be 16-byte aligned, lda %% 4 == 0");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_LINEAR, s);
    return gemm_f32_linear((const float*)a, (const float*)w, (const float*)bias, nullptr, nullptr, (float*)out, M, N, K, lda, ldo, 0, act, s,
                           (float*)aux);
}

int master_linear_res_fwd(const void* a, const void* w, const void* bias, const void* res, const float* rowscale, int rows_per_scale, void* out,
                          void* aux, int M, int N, int K, int lda, int ldo, int ldr, int act, int dtype, master_stream_t stream) {
    F32_ONLY("linear_res_fwd");
    REQUIRE(a && w && out, "linear_res_fwd: null operand");
    REQUIRE(M > 0 && N > 0 && K > 0 && K % 32 == 0 && lda >= K && ldo >= N, "linear_res_fwd: bad dims M=%d N=%d K=%d", M, N, K);
    REQUIRE(al16(a) && al16(w) && lda % 4 == 0, "linear_res_fwd: a/w must be 16-byte aligned, lda %% 4 == 0");
    REQUIRE(res == nullptr || ldr >= N, "linear_res_fwd: ldr < N");
    REQUIRE(rowscale == nullptr || rows_per_scale > 0, "linear_res_fwd: rows_per_scale must be positive");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_LINEAR, s);
    return gemm_f32_linear((const float*)a, (const float*)w, (const float*)bias, nullptr, (const float*)res, (float*)out, M, N, K, lda, ldo, ldr, act, s,
                           (float*)aux, rowscale, rows_per_scale);
}

int master_linear_dgrad(const void* g, const void* w, void* da, const void* gelu_pre, const void* add, const float* rowscale, int rows_per_scale,
                        int M, int N, int K, int ldg, int ldda, int ld_pre, int ld_add, int dtype, master_stream_t stream) {
    F32_ONLY("linear_dgrad");
    REQUIRE(g && w && da, "linear_dgrad: null operand");
    REQUIRE(M > 0 && N > 0 && K > 0 && N % 32 == 0 && K % 4 == 0 && ldg >= N && ldda >= K, "linear_dgrad: bad dims M=%d N=%d K=%d", M, N, K);
    REQUIRE(al16(g) && al16(w) && ldg % 4 == 0, "linear_dgrad: g/w must be 16-byte aligned, ldg %% 4 == 0");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_LINEAR_BWD, s);
    REQUIRE(add == nullptr || ld_add >= K, "linear_dgrad: ld_add < K");
    REQUIRE(rowscale == nullptr || rows_per_scale > 0, "linear_dgrad: rows_per_scale must be positive");
    return gemm_f32_dgrad((const float*)g, (const float*)w, (float*)da, (const float*)gelu_pre, (const float*)add, M, N, K, ldg, ldda, ld_pre, ld_add, s,
                          rowscale, rows_per_scale);
}

size_t master_gemm_workspace_floats(int rows, int cols) { return gemm_f32_t_workspace(rows, cols); }

int master_linear_wgrad(const void* g, const void* a, void* dw, void* dbias, const float* rowscale, int rows_per_scale, int M, int N, int K, int ldg,
                        int lda, void* ws, size_t ws_floats, int dtype, master_stream_t stream) {
    const int accumulate = (dtype & MASTER_ACCUMULATE) != 0; dtype &= ~MASTER_ACCUMULATE;
    F32_ONLY("linear_wgrad");
    REQUIRE(g && a && dw, "linear_wgrad: null operand");
    REQUIRE(M > 0 && N > 0 && K > 0 && K % 4 == 0 && ldg >= N && lda >= K && lda % 4 == 0, "linear_wgrad: bad dims M=%d N=%d K=%d", M, N, K);
    REQUIRE(al16(a) && al16(dw) && (ws == nullptr || al16(ws)), "linear_wgrad: a/dw/ws must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_LINEAR_BWD, s);
    REQUIRE(rowscale == nullptr || rows_per_scale > 0, "linear_wgrad: rows_per_scale must be positive");
    return gemm_f32_wgrad((const float*)g, (const float*)a, (float*)dw, (float*)dbias, M, N, K, ldg, lda, (float*)ws, ws_floats, s, accumulate, rowscale,
                          rows_per_scale);
}

size_t master_colsum_workspace_floats(int M, int N) { return (size_t)colsum_slabs(M, N) * (size_t)N; }
int master_colsum(const void* x, void* out, void* ws, int M, int N, int ld, int dtype, master_stream_t stream) {
    const int accumulate = (dtype & MASTER_ACCUMULATE) != 0; dtype &= ~MASTER_ACCUMULATE;
    F32_ONLY("colsum");
    REQUIRE(x && out && M > 0 && N > 0 && N % 4 == 0 && ld >= N && ld % 4 == 0 && al16(x), "colsum: bad arguments (M=%d N=%d ld=%d)", M, N, ld);
    REQUIRE(ws || (colsum_slabs(M, N) == 1 && !accumulate), "colsum: workspace required");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return colsum_launch((const float*)x, nullptr, (float*)out, (float*)ws, M, N, ld, s, accumulate);
}

int master_colsum_prod(const void* x, const void* y, void* out, void* ws, int M, int N, int ld, int dtype, master_stream_t stream) {
    F32_ONLY("colsum_prod");
    REQUIRE(x && y && out && M > 0 && N > 0 && N % 4 == 0 && ld >= N && ld % 4 == 0 && al16(x) && al16(y), "colsum_prod: bad arguments");
    REQUIRE(ws || colsum_slabs(M, N) == 1, "colsum_prod: workspace required");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return colsum_launch((const float*)x, (const float*)y, (float*)out, (float*)ws, M, N, ld, s);
}

int master_channel_affine_fwd(const void* x, const float* gamma, const float* beta, void* out, int M, int C, int dtype, master_stream_t stream) {
    REQUIRE(x && gamma && out && M > 0 && C > 0 && C % 4 == 0 && al16(x) && al16(out), "channel_affine: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_INSTNORM, s);
    return dtype == MASTER_F32 ? channel_affine_launch<float>(x, gamma, beta, out, M, C, s) : channel_affine_launch<bf16_t>(x, gamma, beta, out, M, C, s);
}

int master_softmax_rows_fwd(const void* x, void* out, int M, int N, int ld, float scale, int dtype, master_stream_t stream) {
    REQUIRE(x && out && M > 0 && N > 0 && N % 4 == 0 && ld >= N && ld % 4 == 0 && al16(x) && al16(out), "softmax_rows: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_ATTN, s);
    return dtype == MASTER_F32 ? softmax_rows_launch<float>(x, out, M, N, ld, scale, s) : softmax_rows_launch<bf16_t>(x, out, M, N, ld, scale, s);
}

int master_softmax_rows_bwd(const void* p, const void* g, void* dx, int M, int N, int ld, float scale, int dtype, master_stream_t stream) {
    F32_ONLY("softmax_rows_bwd");
    REQUIRE(p && g && dx && M > 0 && N > 0 && N % 4 == 0 && ld >= N && ld % 4 == 0 && al16(p) && al16(g) && al16(dx), "softmax_rows_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_ATTN_BWD, s);
    return softmax_rows_bwd_launch((const float*)p, (const float*)g, (float*)dx, M, N, ld, scale, s);
}

int master_reduce_slabs(const float* part, float* out, size_t n, size_t slab, int splits, master_stream_t stream) {
    REQUIRE(part && out && n > 0 && splits >= 1, "reduce_slabs: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return reduce_slabs_launch(part, out, n, slab, splits, s);
}

int master_reduce_segments(const master_reduce_seg* segs, int count, master_stream_t stream) {
    REQUIRE(segs && count >= 1 && count <= 4, "reduce_segments: 1..4 segments");
    const float* part[4]; float* out[4]; size_t n[4], slab[4]; int splits[4], acc[4];
    for (int q = 0; q < count; ++q) {
        REQUIRE(segs[q].part && segs[q].out && segs[q].n > 0 && segs[q].splits >= 1, "reduce_segments: bad segment %d", q);
        part[q] = segs[q].part; out[q] = segs[q].out; n[q] = segs[q].n; slab[q] = segs[q].slab; splits[q] = segs[q].splits; acc[q] = segs[q].accumulate != 0;
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return reduce_segments_launch(part, out, n, slab, splits, acc, count, s);
}

size_t master_layernorm_bwd_workspace_floats(int M, int C) { return (size_t)layernorm_bwd_blocks(M) * 2 * (size_t)C; }
int master_layernorm_bwd(const void* x, const float* gamma, const void* g, const void* add, void* dx, float* dgb, float* ws, int M, int C, float eps,
                         int dtype, master_stream_t stream) {
    const int accumulate = (dtype & MASTER_ACCUMULATE) != 0; dtype &= ~MASTER_ACCUMULATE;
    F32_ONLY("layernorm_bwd");
    REQUIRE(x && gamma && g && dx && dgb && ws && M > 0, "layernorm_bwd: null operand");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_LAYERNORM, s);
    REQUIRE(add == nullptr || al16(add), "layernorm_bwd: add must be 16-byte aligned");
    return layernorm_bwd_launch((const float*)x, gamma, (const float*)g, (float*)dx, dgb, ws, M, C, eps, s, accumulate, (const float*)add);
}

int master_instnorm_bwd(const void* x, const void* g, const void* pad_row_in, const void* g_pad, void* dx, void* dpad, int B, int T, int C,
                        int n_pad, float eps, int dtype, master_stream_t stream) {
    F32_ONLY("instnorm_bwd");
    REQUIRE(x && g && dx && B > 0 && T > 0 && n_pad >= 0, "instnorm_bwd: bad arguments");
    REQUIRE(n_pad == 0 || pad_row_in, "instnorm_bwd: n_pad > 0 needs pad_row_in");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_INSTNORM, s);
    return instnorm_bwd_launch((const float*)x, (const float*)g, (const float*)pad_row_in, (const float*)g_pad, (float*)dx, (float*)dpad, B, T, C,
                               n_pad, eps, s);
}

int master_window_attn_bwd(const void* q, const void* k, const void* v0, const void* v1, int ldq, int ldk, int ldv, const void* k_pad,
                           const void* v0_pad, const void* v1_pad, int k_pad_bstride, int v_pad_bstride, const float* bias,
                           const int32_t* src, const uint8_t* region, const int32_t* pad_ord, const void* do0, const void* do1, int ldo,
                           void* dq, void* dk, void* dv0, void* dv1, int lddq, int lddk, int lddv, void* dk_pad, void* dv0_pad,
                           void* dv1_pad, float* dbias_ws, int B, int T, int heads, int nW, int N, int n_pad, float scale, int dtype,
                           master_stream_t stream) {
    F32_ONLY("window_attn_bwd");
    REQUIRE(q && k && v0 && bias && src && pad_ord && do0 && dq && dk && dv0, "window_attn_bwd: null operand");
    REQUIRE((v1 == nullptr) == (do1 == nullptr) && (v1 == nullptr) == (dv1 == nullptr), "window_attn_bwd: v1, do1 and dv1 come together");
    REQUIRE(B > 0 && T > 0 && heads > 0 && nW > 0 && N > 0 && N <= 64 && n_pad >= 0, "window_attn_bwd: bad dims");
    REQUIRE(n_pad == 0 || (k_pad && v0_pad && dk_pad && dv0_pad && (!v1 || (v1_pad && dv1_pad))), "window_attn_bwd: padded slots need the pad rows and their gradient buffers");
    const int C = heads * 32;
    REQUIRE(ldq >= C && ldk >= C && ldv >= C && ldo >= C && lddq >= C && lddk >= C && lddv >= C, "window_attn_bwd: row strides < heads*32");
    REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0 && lddq % 4 == 0 && lddk % 4 == 0 && lddv % 4 == 0 &&
            al16(q) && al16(k) && al16(v0) && al16(do0) && al16(dq) && al16(dk) && al16(dv0), "window_attn_bwd: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_ATTN_BWD, s);
    return window_attn_bwd_f32((const float*)q, (const float*)k, (const float*)v0, (const float*)v1, ldq, ldk, ldv, (const float*)k_pad,
                               (const float*)v0_pad, (const float*)v1_pad, k_pad_bstride, v_pad_bstride, bias, src, region, pad_ord,
                               (const float*)do0, (const float*)do1, ldo, (float*)dq, (float*)dk, (float*)dv0, (float*)dv1, lddq, lddk, lddv,
                               (float*)dk_pad, (float*)dv0_pad, (float*)dv1_pad, dbias_ws, B, T, heads, nW, N, n_pad, scale, s);
}

int master_conv3x3_dgrad(const void* g, const void* wf, void* out, const void* relu_src, const void* add, int B, int H, int W, int Cg, int Cx,
                         int full, int dtype, master_stream_t stream) {
    REQUIRE(g && wf && out, "conv3x3_dgrad: null operand");
    if (dtype == MASTER_BF16) {          // the frozen loss network in bf16: zero-padded form only
        REQUIRE(full == 0 && Cg % 32 == 0 && Cx % 4 == 0 && al16(g) && al16(wf), "conv3x3_dgrad(bf16): zero-padded form, Cg %% 32 == 0");
        hipStream_t sb = (hipStream_t)stream;
        ProfScope psb(MASTER_K_CONV_DGRAD, sb);
        return gemm_bf16_conv3x3_dgrad((const bf16_t*)g, (const bf16_t*)wf, (bf16_t*)out, (const bf16_t*)relu_src, (const bf16_t*)add, B, H, W, Cg, Cx, sb);
    }
    if (dtype == MASTER_F32X3) {         // the frozen loss network in split-bf16: fp32 gradients, wf = master_split_weights of the flipped weights
        REQUIRE(full == 0 && B > 0 && H > 0 && W > 0 && Cg > 0 && Cg % 32 == 0 && Cg <= 512 && Cx > 0 && al16(g) && al16(wf) && al16(out),
                "conv3x3_dgrad(x3): zero-padded form, 0 < Cg <= 512, Cg %% 32 == 0, 16-byte aligned g / wf / out");
        hipStream_t sx = (hipStream_t)stream;
        ProfScope psx(MASTER_K_CONV_DGRAD, sx);
        return gemm_x3_conv3x3_dgrad((const float*)g, wf, (float*)out, (const float*)relu_src, (const float*)add, B, H, W, Cg, Cx, sx);
    }
    F32_ONLY("conv3x3_dgrad");
    REQUIRE(B > 0 && H > 0 && W > 0 && Cg > 0 && Cg % 32 == 0 && Cx > 0 && (full == 0 || full == 1), "conv3x3_dgrad: bad dims (Cg=%d)", Cg);
    REQUIRE(al16(g) && al16(wf), "conv3x3_dgrad: g/wf must be 16-byte aligned");
    REQUIRE(Cg <= 512, "conv3x3_dgrad: Cg <= 512 (taps outside the image read a 512-float zero row; got %d)", Cg);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_CONV_DGRAD, s);
    return gemm_f32_conv3x3_dgrad((const float*)g, (const float*)wf, (float*)out, (const float*)relu_src, (const float*)add, B, H, W, Cg, Cx,
                                  full, s);
}

int master_conv3x3_wgrad(const void* g, const void* x, void* dw, void* dbias, int B, int Hin, int Win, int Cin, int Cout, int ldg, int upsample,
                         int zero_pad, void* ws, size_t ws_floats, int dtype, master_stream_t stream) {
    F32_ONLY("conv3x3_wgrad");
    REQUIRE(g && x && dw, "conv3x3_wgrad: null operand");
    REQUIRE(B > 0 && Hin > 0 && Win > 0 && Cin > 0 && Cin % 4 == 0 && Cout > 0 && ldg >= Cout && (upsample == 0 || upsample == 1),
            "conv3x3_wgrad: bad dims");
    REQUIRE(al16(x) && al16(dw) && (ws == nullptr || al16(ws)), "conv3x3_wgrad: x/dw/ws must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_CONV_WGRAD, s);
    return gemm_f32_conv3x3_wgrad((const float*)g, (const float*)x, (float*)dw, (float*)dbias, B, Hin, Win, Cin, Cout, ldg, upsample, zero_pad,
                                  (float*)ws, ws_floats, s);
}

int master_relu_mask(const void* g, const void* relu_out, void* gm, int B, int HW, int C, int Cp, int g_nchw, int dtype, master_stream_t stream) {
    F32_ONLY("relu_mask");
    REQUIRE(g && gm && B > 0 && HW > 0 && C > 0 && Cp >= C, "relu_mask: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return relu_mask_launch((const float*)g, (const float*)relu_out, (float*)gm, B, HW, C, Cp, g_nchw, s);
}

int master_fold_reflect(const void* dxp, void* out, int B, int H, int W, int C, int upsample, int dtype, master_stream_t stream) {
    F32_ONLY("fold_reflect");
    REQUIRE(dxp && out && B > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0 && (upsample == 0 || (upsample == 1 && H % 2 == 0 && W % 2 == 0)),
            "fold_reflect: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return fold_reflect_launch((const float*)dxp, (float*)out, B, H, W, C, upsample, s);
}

int master_conv3x3_cout3_dgrad(const float* g, const float* w, float* dxp, int B, int H, int W, int Cin, master_stream_t stream) {
    REQUIRE(g && w && dxp && B > 0 && H >= 2 && W >= 2, "conv3x3_cout3_dgrad: bad arguments");
    if (Cin != 32) { master_set_error("conv3x3_cout3_dgrad: built for Cin = 32 (got %d)", Cin); return MASTER_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_CONV_DGRAD, s);
    return conv_cout3_dgrad_launch(g, w, dxp, B, H, W, s);
}
size_t master_conv3x3_cout3_wgrad_workspace_floats(int B, int H, int W) { return conv_cout3_wgrad_workspace(B, H, W); }
int master_conv3x3_cout3_wgrad(const float* g, const float* x, float* dw, float* dbias, float* ws, size_t ws_floats, int B, int H, int W, int Cin,
                               master_stream_t stream) {
    REQUIRE(g && x && dw && ws && B > 0 && H >= 2 && W >= 2, "conv3x3_cout3_wgrad: bad arguments");
    if (Cin != 32) { master_set_error("conv3x3_cout3_wgrad: built for Cin = 32 (got %d)", Cin); return MASTER_ERR_UNSUPPORTED; }
    REQUIRE(ws_floats >= conv_cout3_wgrad_workspace(B, H, W), "conv3x3_cout3_wgrad: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_CONV_WGRAD, s);
    return conv_cout3_wgrad_launch(g, x, dw, dbias, ws, B, H, W, s);
}

int master_conv_weight_pack(const float* w, float* wp, float* wf, int Cout, int Cin, int Cp, master_stream_t stream) {
    REQUIRE(w && wp && Cout > 0 && Cin > 0 && Cp >= Cout, "conv_weight_pack: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return conv_weight_pack_launch(w, wp, wf, Cout, Cin, Cp, s);
}

int master_bias_tile_fwd(const float* table, const long long* index, float* tile, int N, int heads, master_stream_t stream) {
    REQUIRE(table && index && tile && N > 0 && N <= 64 && heads > 0, "bias_tile_fwd: bad arguments (N=%d heads=%d)", N, heads);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return bias_tile_fwd_launch(table, index, tile, N, heads, s);
}

int master_bias_tile_bwd(const float* dtile, const long long* index, float* dtable, int N, int heads, int entries, master_stream_t stream) {
    REQUIRE(dtile && index && dtable && N > 0 && N <= 64 && heads > 0 && entries > 0, "bias_tile_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return bias_tile_bwd_launch(dtile, index, dtable, N, heads, entries, s);
}

int master_maxpool2x2_bwd(const void* g, const void* y, void* dx, int N, int H, int W, int C, int dtype, master_stream_t stream) {
    REQUIRE(g && y && dx && N > 0 && H >= 2 && W >= 2 && C > 0, "maxpool2x2_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return dtype == MASTER_F32 ? maxpool2_bwd_launch<float>(g, y, dx, N, H, W, C, s) : maxpool2_bwd_launch<bf16_t>(g, y, dx, N, H, W, C, s);
}

int master_nhwc_to_image(const void* g, float* out, int N, int HW, int Cp, int dtype, master_stream_t stream) {
    REQUIRE(g && out && N > 0 && HW > 0 && Cp >= 3, "nhwc_to_image: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return dtype == MASTER_F32 ? nhwc_to_image_launch<float>(g, out, N, HW, Cp, s) : nhwc_to_image_launch<bf16_t>(g, out, N, HW, Cp, s);
}

int master_rowscale_add(const void* x, const void* res, const float* scale, void* out, int B, int T, int C, int dtype, master_stream_t stream) {
    F32_ONLY("rowscale_add");
    REQUIRE(x && scale && out && B > 0 && T > 0 && C > 0 && C % 4 == 0 && al16(x) && al16(out) && (res == nullptr || al16(res)),
            "rowscale_add: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return rowscale_add_launch((const float*)x, (const float*)res, scale, (float*)out, B, T, C, s);
}

int master_mul_add_fwd(const void* q, const void* sigma, const void* mu, void* out, size_t n, int dtype, master_stream_t stream) {
    F32_ONLY("mul_add_fwd");
    REQUIRE(q && sigma && mu && out && n > 0 && n % 4 == 0 && al16(q) && al16(sigma) && al16(mu) && al16(out), "mul_add_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return mul_add_launch((const float*)q, (const float*)sigma, (const float*)mu, (float*)out, n, s);
}
int master_mul_add_bwd(const void* g, const void* q, const void* sigma, void* dq, void* dsigma, size_t n, int dtype, master_stream_t stream) {
    F32_ONLY("mul_add_bwd");
    REQUIRE(g && q && sigma && dq && dsigma && n > 0 && n % 4 == 0 && al16(g) && al16(q) && al16(sigma) && al16(dq) && al16(dsigma),
            "mul_add_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return mul_add_bwd_launch((const float*)g, (const float*)q, (const float*)sigma, (float*)dq, (float*)dsigma, n, s);
}

size_t master_loss_tap_workspace_floats(int B, int HW, int C) { return (size_t)B * (size_t)(loss_tap_slices(HW) + 1) * 2 * (size_t)C; }
int master_loss_tap_bwd(const void* x, const float* mean, const float* var, const float* coefA, const float* coefB, float* ws, void* dx, int B,
                        int HW, int C, float eps, const float* g_content, const float* g_style, int squared, int relu_gate, int dtype,
                        master_stream_t stream) {
    REQUIRE(x && mean && var && coefA && coefB && ws && dx && B > 0 && HW > 0, "loss_tap_bwd: null operand");
    REQUIRE(C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0, "loss_tap_bwd: C/4 must divide 256 (C=%d)", C);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    const float wc_unit = 1.0f / ((float)B * (float)C * (float)HW);
    return dtype == MASTER_F32 ? loss_tap_bwd_launch<float>(x, mean, var, coefA, coefB, ws, dx, B, HW, C, eps, wc_unit, g_content, g_style, squared, relu_gate, s)
                               : loss_tap_bwd_launch<bf16_t>(x, mean, var, coefA, coefB, ws, dx, B, HW, C, eps, wc_unit, g_content, g_style, squared, relu_gate, s);
}

int master_loss_terms_fwd(const float* mean, const float* var, const float* dist_partial, int n_partial, float* losses, float* coefA, float* coefB,
                          int B, int HW, int C, int squared_style, int first, master_stream_t stream) {
    REQUIRE(mean && var && losses && coefA && coefB && (dist_partial || n_partial == 0), "loss_terms_fwd: null operand");
    REQUIRE(B > 0 && C > 0 && HW > 1 && n_partial >= 0, "loss_terms_fwd: bad dims B=%d HW=%d C=%d", B, HW, C);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return loss_terms_launch(mean, var, dist_partial, n_partial, losses, coefA, coefB, B, HW, C, squared_style, first, s);
}

int master_adam_chunk(void) { return adam_chunk(); }
int master_adam_step(const void* table, int ntensors, long long nblocks, float lr, float beta1, float beta2, float eps, float weight_decay,
                     int step, master_stream_t stream) {
    REQUIRE(table && ntensors > 0 && nblocks > 0 && nblocks < 0x7fffffffLL && step >= 1, "adam_step: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return adam_launch(table, ntensors, nblocks, lr, beta1, beta2, eps, weight_decay, step, s);
}
int master_adam_step_dev(const void* table, int ntensors, long long nblocks, const float* hyper, int* step_done, master_stream_t stream) {
    REQUIRE(table && hyper && step_done && ntensors > 0 && nblocks > 0 && nblocks < 0x7fffffffLL, "adam_step_dev: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return adam_dev_launch(table, ntensors, nblocks, hyper, step_done, s);
}
int master_lerp(const void* table, int ntensors, long long nblocks, float eta, master_stream_t stream) {
    REQUIRE(table && ntensors > 0 && nblocks > 0 && nblocks < 0x7fffffffLL, "lerp: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_TRAIN_MISC, s);
    return lerp_launch(table, ntensors, nblocks, eta, s);
}

int master_prof_enable(int on) { prof_drain(); g_prof_on = on != 0; return MASTER_OK; }
int master_prof_reset(void) {
    prof_drain();
    for (int i = 0; i < MASTER_K_COUNT; ++i) { g_launches[i] = 0; g_ms[i] = 0.0; }
    return MASTER_OK;
}
int master_prof_read(int family, int64_t* launches, double* total_ms) {
    REQUIRE(family >= 0 && family < MASTER_K_COUNT, "prof_read: bad family %d", family);
    prof_drain();
    if (launches) *launches = g_launches[family];
    if (total_ms) *total_ms = g_ms[family];
    return MASTER_OK;
}

int master_linear_fwd(const void* a, const void* w, const void* bias, const void* r1, const void* r2, void* out,
                      int M, int N, int K, int lda, int ldo, int ldr, int act, int dtype, master_stream_t stream) {
    REQUIRE(a && w && out, "linear: null operand");
    REQUIRE(M > 0 && N > 0 && K > 0 && K % 32 == 0, "linear: need M,N>0 and K%%32==0 (M=%d N=%d K=%d)", M, N, K);
    REQUIRE(lda >= K && ldo >= N && ((r1 == nullptr && r2 == nullptr) || ldr >= N), "linear: bad leading dims");
    REQUIRE(act >= MASTER_ACT_NONE && act <= MASTER_ACT_RELU, "linear: bad act %d", act);
    const int io = dtype & (MASTER_A_SPLIT | MASTER_OUT_SPLIT);
    dtype &= ~(MASTER_A_SPLIT | MASTER_OUT_SPLIT);
    REQUIRE(io == 0 || is_x3(dtype), "linear: MASTER_A_SPLIT / MASTER_OUT_SPLIT go with MASTER_F32X3 / MASTER_F16X3");
    REQUIRE(!(io & MASTER_A_SPLIT) || lda % 32 == 0, "linear: a split A operand has whole 32-wide chunks per row (lda %% 32 == 0)");
    REQUIRE(!(io & MASTER_OUT_SPLIT) || (N % 32 == 0 && ldo % 32 == 0 && al16(out)), "linear: a split output needs N, ldo %% 32 == 0");
    // the split epilogue exists in the whole-row form only: everything it loads must be addressable in 16-byte pieces
    REQUIRE(!(io & MASTER_OUT_SPLIT) || ((!bias || al16(bias)) && ((!r1 && !r2) || (ldr % 4 == 0 && (!r1 || al16(r1)) && (!r2 || al16(r2))))),
            "linear: a split output needs 16-byte aligned bias / r1 / r2 and ldr %% 4 == 0");
    const int ea = dtype == MASTER_BF16 ? 8 : 4;   // elements per 16 B
    REQUIRE(al16(a) && al16(w) && lda % ea == 0, "linear: a/w must be 16-byte aligned, lda%%%d==0", ea);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_LINEAR, s);
    if (is_x3(dtype))
        return gemm_x3_linear((const float*)a, w, (const float*)bias, (const float*)r1, (const float*)r2, (float*)out, M, N, K, lda, ldo,
                              ldr, act, s, (io & MASTER_A_SPLIT) != 0, (io & MASTER_OUT_SPLIT) != 0, x3_fmt(dtype));
    if (dtype == MASTER_F32)
        return gemm_f32_linear((const float*)a, (const float*)w, (const float*)bias, (const float*)r1,
                               (const float*)r2, (float*)out, M, N, K, lda, ldo, ldr, act, s);
    if (dtype == MASTER_BF16) {
        if (act == MASTER_ACT_NONE && ldo % 4 == 0 && (ldr % 4 == 0)) {
            // token-stationary kernel (weights streamed by LDS-DMA) where it applies, tiled GEMM otherwise
            const int rc = linear_ts_bf16((const bf16_t*)a, nullptr, nullptr, (const bf16_t*)w, (const float*)bias,
                                          (const bf16_t*)r1, (const bf16_t*)r2, (bf16_t*)out, M, N, K, lda, ldr, ldo, 0.f, s);
            if (rc != MASTER_ERR_UNSUPPORTED) return rc;
        }
        return gemm_bf16_linear((const bf16_t*)a, (const bf16_t*)w, (const float*)bias, (const bf16_t*)r1,
                                (const bf16_t*)r2, (bf16_t*)out, M, N, K, lda, ldo, ldr, act, s);
    }
    master_set_error("linear: bad dtype %d", dtype);
    return MASTER_ERR_INVALID;
}

int master_split_weights(const float* w, void* out, size_t rows, size_t K, int layout, master_stream_t stream) {
    REQUIRE(w && out && rows > 0 && K > 0 && K % 32 == 0, "split_weights: need K%%32==0 (rows=%zu K=%zu)", rows, K);
    const int fmt = (layout & MASTER_SPLIT_F16) ? X3_F16 : X3_BF16;
    layout &= ~MASTER_SPLIT_F16;
    REQUIRE(layout == 0 || layout == 1, "split_weights: layout 0 (plain) or 1 (accumulator k order), optionally | MASTER_SPLIT_F16");
    REQUIRE(al16(w) && al16(out), "split_weights: 16-byte alignment");
    return gemm_x3_split_weights(w, out, rows, K, layout, (hipStream_t)stream, fmt);
}

int master_linear_ln_fwd(const void* x, const float* ln_gamma, const float* ln_beta, const void* w, const void* bias,
                         const void* r1, const void* r2, void* out, int M, int N, int K, int ldx, int ldo, int ldr,
                         float eps, int dtype, master_stream_t stream) {
    REQUIRE(x && ln_gamma && ln_beta && w && out, "linear_ln: null operand");
    REQUIRE(M > 0 && N > 0 && K > 0 && ldx >= K && ldo >= N && ((r1 == nullptr && r2 == nullptr) || ldr >= N), "linear_ln: bad dims");
    if (dtype != MASTER_BF16) { master_set_error("linear_ln: fused kernel is bf16-only; use master_layernorm_fwd + master_linear_fwd"); return MASTER_ERR_UNSUPPORTED; }
    REQUIRE(al16(x) && al16(w) && ldx % 8 == 0 && ldo % 4 == 0 && ldr % 4 == 0, "linear_ln: alignment");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_LINEAR, s);
    const int rc = linear_ts_bf16((const bf16_t*)x, ln_gamma, ln_beta, (const bf16_t*)w, (const float*)bias, (const bf16_t*)r1,
                                  (const bf16_t*)r2, (bf16_t*)out, M, N, K, ldx, ldr, ldo, eps, s);
    if (rc == MASTER_ERR_UNSUPPORTED) master_set_error("linear_ln: shape M=%d N=%d K=%d not covered by the fused kernel", M, N, K);
    return rc;
}

int master_linear_group_fwd(int n, const void* const* x, const void* const* w, const void* const* bias, void* const* out, const int* N,
                            const int* ldx, const int* ldo, const float* ln_gamma, const float* ln_beta, int M, int K, float eps, int dtype,
                            master_stream_t stream) {
    REQUIRE(n >= 1 && n <= 3 && x && w && bias && out && N && ldx && ldo, "linear_group: 1..3 problems, non-null tables");
    REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), "linear_group: ln_gamma and ln_beta come together");
    REQUIRE(M > 0 && K > 0, "linear_group: bad dims");
    if (dtype != MASTER_BF16) { master_set_error("linear_group: bf16 only (the fp32 mode launches master_linear_fwd per problem)"); return MASTER_ERR_UNSUPPORTED; }
    for (int i = 0; i < n; ++i) {
        REQUIRE(x[i] && w[i] && out[i] && N[i] > 0 && ldx[i] >= K && ldo[i] >= N[i], "linear_group: problem %d: null operand or bad leading dimension", i);
        REQUIRE(al16(x[i]) && al16(w[i]) && ldx[i] % 8 == 0 && ldo[i] % 4 == 0, "linear_group: problem %d: alignment", i);
    }
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_LINEAR, s);
    const int rc = linear_ts_group_bf16(n, (const bf16_t* const*)x, (const bf16_t* const*)w, (const float* const*)bias, (bf16_t* const*)out, N, ldx,
                                        ldo, ln_gamma, ln_beta, M, K, eps, s);
    if (rc == MASTER_ERR_UNSUPPORTED) master_set_error("linear_group: shapes (K=%d) not covered by the token-stationary kernel", K);
    return rc;
}

int master_resize_plan(int in_size, int out_size, int32_t* bounds, int32_t* coeffs, int* ksize) {
    REQUIRE(in_size > 0 && out_size > 0 && ksize, "resize_plan: sizes must be positive and ksize non-null");
    return resize_plan_host(in_size, out_size, bounds, coeffs, ksize);
}

int master_preprocess_fwd(const uint8_t* img, int B, int H, int W, float* out, int OH, int OW, const int32_t* h_bounds,
                          const int32_t* h_coeffs, int h_ksize, const int32_t* v_bounds, const int32_t* v_coeffs, int v_ksize,
                          uint8_t* tmp, const float* mean, const float* stdv, master_stream_t stream) {
    REQUIRE(img && out && v_bounds && v_coeffs, "preprocess: null operand");
    REQUIRE(B > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && v_ksize > 0, "preprocess: bad dims");
    REQUIRE(OW == W || (h_bounds && h_coeffs && h_ksize > 0 && tmp), "preprocess: the horizontal pass needs its tables and the [B,H,OW,3] workspace");
    REQUIRE((mean == nullptr) == (stdv == nullptr), "preprocess: mean and std come together");
    return preprocess_launch(img, B, H, W, out, OH, OW, h_bounds, h_coeffs, h_ksize, v_bounds, v_coeffs, v_ksize, tmp, mean, stdv,
                             (hipStream_t)stream);
}

int master_postprocess_fwd(const float* img, uint8_t* out, int B, int H, int W, master_stream_t stream) {
    REQUIRE(img && out && B > 0 && H > 0 && W > 0, "postprocess: bad operand");
    return postprocess_launch(img, out, B, H, W, (hipStream_t)stream);
}

int master_patch_merge_linear_fwd(const void* x, const float* gamma, const float* beta, const void* w, void* out, int B, int H,
                                  int W, int C, int N, float eps, int dtype, master_stream_t stream) {
    REQUIRE(x && gamma && beta && w && out, "patch_merge_linear: null operand");
    REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && N > 0, "patch_merge_linear: H, W must be even");
    if (dtype != MASTER_BF16) { master_set_error("patch_merge_linear: fused kernel is bf16-only; use master_patch_merge_ln_fwd + master_linear_fwd"); return MASTER_ERR_UNSUPPORTED; }
    REQUIRE(al16(x) && al16(w), "patch_merge_linear: alignment");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_LINEAR, s);
    const int M = B * (H / 2) * (W / 2);
    const int rc = linear_ts_bf16((const bf16_t*)x, gamma, beta, (const bf16_t*)w, nullptr, nullptr, nullptr, (bf16_t*)out, M, N, 4 * C,
                                  4 * C, 0, N, eps, s, H, W, C);
    if (rc == MASTER_ERR_UNSUPPORTED) master_set_error("patch_merge_linear: C=%d N=%d not covered by the fused kernel", C, N);
    return rc;
}

int master_mlp_fwd(const void* x, const float* ln_gamma, const float* ln_beta, const void* w1, const float* b1,
                   const void* w2, const float* b2, const void* res, void* out, int M, int C, int HID, int ldx, int ldr,
                   int ldo, float eps, int dtype, master_stream_t stream) {
    REQUIRE(x && w1 && b1 && w2 && b2 && out, "mlp: null operand");
    REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), "mlp: ln_gamma and ln_beta come together");
    REQUIRE(M > 0 && C > 0 && HID > 0 && ldx >= C && ldo >= C && (res == nullptr || ldr >= C), "mlp: bad dims");
    if (is_x3(dtype)) {
        REQUIRE(al16(x) && al16(w1) && al16(w2) && al16(out) && (!res || al16(res)) && ldx % 4 == 0 && ldo % 4 == 0 && ldr % 4 == 0,
                "mlp(x3): alignment (x/w/res/out 16 B, ldx, ldo, ldr %% 4)");
        hipStream_t s3 = (hipStream_t)stream;
        ProfScope ps3(MASTER_K_MLP, s3);
        return mlp_x3((const float*)x, ln_gamma, ln_beta, w1, b1, w2, b2, (const float*)res, (float*)out, M, C, HID, ldx, ldr, ldo, eps, s3, x3_fmt(dtype));
    }
    if (dtype != MASTER_BF16) { master_set_error("mlp: fused kernel is bf16 / f32x3 only; use two master_linear_fwd calls in fp32 mode"); return MASTER_ERR_UNSUPPORTED; }
    REQUIRE(al16(x) && al16(w1) && al16(w2) && ldx % 8 == 0 && ldo % 4 == 0 && ldr % 4 == 0, "mlp: alignment (x/w 16 B, ldx%%8, ldo%%4, ldr%%4)");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_MLP, s);
    // forms of the same kernel.  Default: the 256-token specialisation for C = 128 (mlp_c128_bf16.hip), else the one-wave-per-SIMD
    // form (mlp_bf16.hip).  MASTER_MLP_FORM=mono forces the generic form.
    const char* form = getenv("MASTER_MLP_FORM");              // read per call: tests cover every form in one process
    if (form && !strcmp(form, "2w")) {                        // C = 128: two workgroups per CU, GELU dealt out behind the phase-1 MFMAs (mlp_2w_bf16.hip)
        const int rc = mlp_2w_bf16((const bf16_t*)x, ln_gamma, ln_beta, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, (const bf16_t*)res,
                                   (bf16_t*)out, M, C, HID, ldx, ldr, ldo, eps, s);
        if (rc != MASTER_ERR_UNSUPPORTED) return rc;
    }
    if (form && !strcmp(form, "8w")) {                        // C = 256: eight waves, two per SIMD on the same tokens (mlp_8w_bf16.hip, r05)
        const int rc = mlp_8w_bf16((const bf16_t*)x, ln_gamma, ln_beta, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, (const bf16_t*)res,
                                   (bf16_t*)out, M, C, HID, ldx, ldr, ldo, eps, s);
        if (rc != MASTER_ERR_UNSUPPORTED) return rc;
    }
    if (!form || strcmp(form, "mono")) {
        const int rc = mlp_c128_bf16((const bf16_t*)x, ln_gamma, ln_beta, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, (const bf16_t*)res,
                                     (bf16_t*)out, M, C, HID, ldx, ldr, ldo, eps, s);
        if (rc != MASTER_ERR_UNSUPPORTED) return rc;
    }
    return mlp_bf16((const bf16_t*)x, ln_gamma, ln_beta, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, (const bf16_t*)res,
                    (bf16_t*)out, M, C, HID, ldx, ldr, ldo, eps, s);
}

int master_conv3x3_reflect_fwd(const void* x, const void* w, const void* bias, void* out, int B, int Hin, int Win,
                               int Cin, int Cout, int upsample, int relu, int out_nchw, int dtype,
                               master_stream_t stream) {
    REQUIRE(x && w && out, "conv3x3: null operand");
    REQUIRE(B > 0 && Hin > 0 && Win > 0 && Cout > 0 && Cin > 0 && Cin % 32 == 0, "conv3x3: need Cin%%32==0 (Cin=%d)", Cin);
    REQUIRE(upsample == 0 || upsample == 1, "conv3x3: upsample must be 0/1");
    REQUIRE((Hin << upsample) >= 2 && (Win << upsample) >= 2, "conv3x3: reflect pad needs H,W >= 2");
    REQUIRE(al16(x) && al16(w), "conv3x3: x/w must be 16-byte aligned");
    const int io = dtype & (MASTER_A_SPLIT | MASTER_OUT_SPLIT);
    dtype &= ~(MASTER_A_SPLIT | MASTER_OUT_SPLIT);
    REQUIRE(io == 0 || is_x3(dtype), "conv3x3: MASTER_A_SPLIT / MASTER_OUT_SPLIT go with MASTER_F32X3 / MASTER_F16X3");
    REQUIRE(!(io & MASTER_OUT_SPLIT) || (Cout % 32 == 0 && !out_nchw && al16(out) && (!bias || al16(bias))),
            "conv3x3: a split output is NHWC with Cout %% 32 == 0 and a 16-byte aligned bias");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_CONV, s);
    if (is_x3(dtype))
        return gemm_x3_conv3x3((const float*)x, w, (const float*)bias, (float*)out, B, Hin, Win, Cin, Cout, upsample, relu, out_nchw, s, 0,
                               (io & MASTER_A_SPLIT) != 0, (io & MASTER_OUT_SPLIT) != 0, x3_fmt(dtype));
    if (dtype == MASTER_F32)
        return gemm_f32_conv3x3((const float*)x, (const float*)w, (const float*)bias, (float*)out, B, Hin, Win, Cin,
                                Cout, upsample, relu, out_nchw, s);
    if (dtype == MASTER_BF16) {
        if (!out_nchw) {      // tile kernels where they apply: pixel-stationary ring (Cin 256/128, and 64 -> 32), then the Cin = 64 halo kernel
            int rc = conv_ring_bf16((const bf16_t*)x, (const bf16_t*)w, (const float*)bias, (bf16_t*)out, B, Hin, Win, Cin, Cout, upsample, relu, s);
            if (rc != MASTER_ERR_UNSUPPORTED) return rc;
            rc = conv_halo_bf16((const bf16_t*)x, (const bf16_t*)w, (const float*)bias, (bf16_t*)out, B, Hin, Win, Cin, Cout, upsample, relu, s);
            if (rc != MASTER_ERR_UNSUPPORTED) return rc;
        }
        return gemm_bf16_conv3x3((const bf16_t*)x, (const bf16_t*)w, (const float*)bias, out, B, Hin, Win, Cin, Cout,
                                 upsample, relu, out_nchw, s);
    }
    master_set_error("conv3x3: bad dtype %d", dtype);
    return MASTER_ERR_INVALID;
}

int master_conv3x3_zero_fwd(const void* x, const void* w, const void* bias, void* out, int B, int H, int W, int Cin, int Cout,
                            int relu, int dtype, master_stream_t stream) {
    REQUIRE(x && w && out, "conv3x3_zero: null operand");
    REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && Cin > 0 && Cin % 32 == 0, "conv3x3_zero: need Cin%%32==0 (Cin=%d; pad the 3-channel image to 32)", Cin);
    REQUIRE(al16(x) && al16(w), "conv3x3_zero: x/w must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_CONV, s);
    if (dtype == MASTER_F32) {
        REQUIRE(Cin <= 512, "conv3x3_zero: Cin <= 512 (taps outside the image read a 512-float zero row; got %d)", Cin);
        return gemm_f32_conv3x3((const float*)x, (const float*)w, (const float*)bias, (float*)out, B, H, W, Cin, Cout, 0, relu, 0, s, 1);
    }
    if (dtype == MASTER_BF16)
        return gemm_bf16_conv3x3((const bf16_t*)x, (const bf16_t*)w, (const float*)bias, out, B, H, W, Cin, Cout, 0, relu, 0, s, 1);
    if (is_x3(dtype)) {                  // fp32 maps, w = master_split_weights of the packed weights
        // taps outside the image read a 512-float zero row at channel offsets up to Cin - 32 (gemm_x3.hip: g_zero_row_x3)
        REQUIRE(Cin <= 512, "conv3x3_zero(x3): Cin <= 512 (got %d)", Cin);
        return gemm_x3_conv3x3((const float*)x, w, (const float*)bias, (float*)out, B, H, W, Cin, Cout, 0, relu, 0, s, 1, 0, 0, x3_fmt(dtype));
    }
    master_set_error("conv3x3_zero: bad dtype %d", dtype);
    return MASTER_ERR_INVALID;
}

int master_image_to_nhwc_fwd(const float* img, void* out, int N, int H, int W, int Cpad, int dtype, master_stream_t stream) {
    REQUIRE(img && out && N > 0 && H > 0 && W > 0 && Cpad >= 3, "image_to_nhwc: bad operand");
    if (dtype == MASTER_F32) return image_to_nhwc_launch<float>(img, out, N, H, W, Cpad, (hipStream_t)stream);
    if (dtype == MASTER_BF16) return image_to_nhwc_launch<bf16_t>(img, out, N, H, W, Cpad, (hipStream_t)stream);
    master_set_error("image_to_nhwc: bad dtype %d", dtype); return MASTER_ERR_INVALID;
}

int master_vgg_conv1_fwd(const float* img, const float* w, const float* bias, void* out, int N, int H, int W, int relu, int dtype,
                         master_stream_t stream) {
    REQUIRE(img && w && bias && out && N > 0 && H > 0 && W > 0, "vgg_conv1: bad operand");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_CONV, s);
    if (dtype == MASTER_F32) return vgg_conv1_launch<float>(img, w, bias, out, N, H, W, relu, s);
    if (dtype == MASTER_BF16) return vgg_conv1_launch<bf16_t>(img, w, bias, out, N, H, W, relu, s);
    master_set_error("vgg_conv1: bad dtype %d", dtype); return MASTER_ERR_INVALID;
}

int master_maxpool2x2_fwd(const void* x, void* out, int N, int H, int W, int C, int dtype, master_stream_t stream) {
    REQUIRE(x && out && N > 0 && H >= 2 && W >= 2 && C > 0, "maxpool2x2: bad operand");
    if (dtype == MASTER_F32) return maxpool2_launch<float>(x, out, N, H, W, C, (hipStream_t)stream);
    if (dtype == MASTER_BF16) return maxpool2_launch<bf16_t>(x, out, N, H, W, C, (hipStream_t)stream);
    master_set_error("maxpool2x2: bad dtype %d", dtype); return MASTER_ERR_INVALID;
}

int master_stats_slices(int HW) { return stats_slices(HW); }

int master_channel_stats_fwd(const void* x, float* mean, float* var, float* work, int N, int HW, int C, int dtype, master_stream_t stream) {
    REQUIRE(x && mean && var && work && N > 0 && HW > 0 && C > 0, "channel_stats: bad operand");
    const int V = dtype == MASTER_F32 ? 4 : 8;
    REQUIRE(C % V == 0 && C / V <= 256 && 256 % (C / V) == 0 && al16(x), "channel_stats: C=%d must be a power-of-two multiple of the 16-byte vector, at most 256 vectors", C);
    if (dtype == MASTER_F32) return chan_stats_launch<float>(x, mean, var, work, N, HW, C, (hipStream_t)stream);
    if (dtype == MASTER_BF16) return chan_stats_launch<bf16_t>(x, mean, var, work, N, HW, C, (hipStream_t)stream);
    master_set_error("channel_stats: bad dtype %d", dtype); return MASTER_ERR_INVALID;
}

int master_instnorm_distance_fwd(const void* x, const float* mean, const float* var, float* partial, int pairs, int pair_offset, int HW,
                                 int C, float eps, int squared, int dtype, master_stream_t stream) {
    REQUIRE(x && mean && var && partial && pairs > 0 && pair_offset > 0 && HW > 0 && C > 0, "instnorm_distance: bad operand");
    { const int V = dtype == MASTER_F32 ? 4 : 8; REQUIRE(C % V == 0 && C / V <= 256 && 256 % (C / V) == 0 && al16(x), "instnorm_distance: C=%d unsupported", C); }
    if (dtype == MASTER_F32) return in_dist_launch<float>(x, mean, var, partial, pairs, pair_offset, HW, C, eps, squared, (hipStream_t)stream);
    if (dtype == MASTER_BF16) return in_dist_launch<bf16_t>(x, mean, var, partial, pairs, pair_offset, HW, C, eps, squared, (hipStream_t)stream);
    master_set_error("instnorm_distance: bad dtype %d", dtype); return MASTER_ERR_INVALID;
}

int master_decoder_tail_fwd(const void* x, const void* w8, const float* b8, const void* w9, const float* b9, float* out,
                            int B, int Hs, int Ws, int dtype, master_stream_t stream) {
    REQUIRE(x && w8 && b8 && w9 && b9 && out, "decoder_tail: null operand");
    REQUIRE(B > 0 && Hs > 0 && Ws > 0 && al16(x) && al16(w8) && al16(w9), "decoder_tail: bad dims/alignment");
    if (is_x3(dtype & ~MASTER_A_SPLIT)) {           // split-bf16 fast parity mode: x fp32 NHWC, or (MASTER_A_SPLIT) in the split layout
        hipStream_t s3 = (hipStream_t)stream;
        ProfScope ps3(MASTER_K_DECODER_TAIL, s3);
        return conv_tail_x3(x, (dtype & MASTER_A_SPLIT) != 0, w8, b8, w9, b9, out, B, Hs, Ws, s3, x3_fmt(dtype & ~MASTER_A_SPLIT));
    }
    if (dtype != MASTER_BF16) { master_set_error("decoder_tail: fused kernel is bf16-only; use master_conv3x3_reflect_fwd in fp32 mode"); return MASTER_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_DECODER_TAIL, s);
    return conv_tail_bf16((const bf16_t*)x, (const bf16_t*)w8, b8, (const bf16_t*)w9, b9, out, B, Hs, Ws, s);
}

int master_window_attn_fwd(const void* q, const void* k, const void* v0, const void* v1, int ldq, int ldk, int ldv,
                           const void* k_pad, const void* v0_pad, const void* v1_pad, int k_pad_bstride,
                           int v_pad_bstride, const float* bias, const int32_t* src, const uint8_t* region, void* out0,
                           void* out1, int ldo, int B, int T, int heads, int nW, int N, float scale, int dtype,
                           master_stream_t stream) {
    REQUIRE(q && k && v0 && out0 && bias && src, "window_attn: null operand");
    REQUIRE(k_pad && v0_pad, "window_attn: pad rows are required (pass the projection bias)");
    REQUIRE((v1 == nullptr) == (out1 == nullptr) && (v1 == nullptr || v1_pad), "window_attn: v1/out1/v1_pad must come together");
    REQUIRE(B > 0 && T > 0 && heads > 0 && nW > 0 && N > 0 && N <= 64, "window_attn: window of %d tokens unsupported (max 64)", N);
    const int ea = dtype == MASTER_BF16 ? 8 : 4;
    REQUIRE(ldq % ea == 0 && ldk % ea == 0 && ldv % ea == 0 && ldo % ea == 0, "window_attn: leading dims must be multiples of %d", ea);
    REQUIRE(al16(q) && al16(k) && al16(v0) && al16(out0) && al16(k_pad) && al16(bias), "window_attn: operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_ATTN, s);
    if (dtype == MASTER_F32 || is_x3(dtype))                // F32X3 / F16X3: fp32 tensors, both products as three 16-bit MFMAs on split operands
        return window_attn_f32((const float*)q, (const float*)k, (const float*)v0, (const float*)v1, ldq, ldk, ldv,
                               (const float*)k_pad, (const float*)v0_pad, (const float*)v1_pad, k_pad_bstride,
                               v_pad_bstride, bias, src, region, (float*)out0, (float*)out1, ldo, B, T, heads, nW, N,
                               scale, s, dtype == MASTER_F16X3 ? 2 : dtype == MASTER_F32X3);
    if (dtype == MASTER_BF16) {
        // persistent window-per-workgroup kernel for 8-head maps (StyleTransformer, Swin stage 2); one-problem-per-wave otherwise
        const int rc = window_attn_pw_bf16((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v0, (const bf16_t*)v1, ldq, ldk, ldv,
                                           (const bf16_t*)k_pad, (const bf16_t*)v0_pad, (const bf16_t*)v1_pad, k_pad_bstride,
                                           v_pad_bstride, bias, src, region, (bf16_t*)out0, (bf16_t*)out1, ldo, B, T, heads, nW, N, scale, s);
        if (rc != MASTER_ERR_UNSUPPORTED) return rc;
        return window_attn_bf16((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v0, (const bf16_t*)v1, ldq, ldk,
                                ldv, (const bf16_t*)k_pad, (const bf16_t*)v0_pad, (const bf16_t*)v1_pad, k_pad_bstride,
                                v_pad_bstride, bias, src, region, (bf16_t*)out0, (bf16_t*)out1, ldo, B, T, heads, nW,
                                N, scale, s);
    }
    master_set_error("window_attn: bad dtype %d", dtype);
    return MASTER_ERR_INVALID;
}

int master_window_attn_proj_fwd(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, const void* k_pad,
                                const void* v_pad, int k_pad_bstride, int v_pad_bstride, const float* bias, const int32_t* src,
                                const uint8_t* region, const void* wproj, const float* bproj, const void* res, int ldr, void* out,
                                int ldo, int B, int T, int heads, int nW, int N, float scale, int dtype, master_stream_t stream) {
    REQUIRE(q && k && v && out && bias && src && wproj && bproj, "window_attn_proj: null operand");
    REQUIRE(k_pad && v_pad, "window_attn_proj: pad rows are required (pass the projection bias)");
    REQUIRE(B > 0 && T > 0 && heads > 0 && nW > 0 && N > 0 && N <= 64, "window_attn_proj: window of %d tokens unsupported (max 64)", N);
    REQUIRE(out != q && out != k && out != v, "window_attn_proj: in-place call");
    if (dtype != MASTER_BF16) { master_set_error("window_attn_proj: bf16 only (the fp32 mode runs master_window_attn_fwd + master_linear_fwd)"); return MASTER_ERR_UNSUPPORTED; }
    REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0 && (res == nullptr || ldr % 4 == 0), "window_attn_proj: leading dims must be multiples of 8 (q/k/v) and 4 (res/out)");
    REQUIRE(al16(q) && al16(k) && al16(v) && al16(k_pad) && al16(v_pad) && al16(wproj) && al16(out) && (res == nullptr || al16(res)),
            "window_attn_proj: operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_ATTN, s);
    const int rc = window_attn_proj_bf16((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, ldk, ldv, (const bf16_t*)k_pad,
                                         (const bf16_t*)v_pad, k_pad_bstride, v_pad_bstride, bias, src, region, (const bf16_t*)wproj, bproj,
                                         (const bf16_t*)res, ldr, (bf16_t*)out, ldo, B, T, heads, nW, N, scale, s);
    if (rc == MASTER_ERR_UNSUPPORTED) master_set_error("window_attn_proj: geometry outside the fused kernel (heads %d, %d windows per image)", heads, nW);
    return rc;
}

int master_attn_block_fwd(const void* x, const float* ln_gamma, const float* ln_beta, const void* wqkv, const float* bqkv,
                          const void* wproj, const float* bproj, const float* bias64, const int32_t* src,
                          const uint8_t* region, void* out, int B, int T, int C, int heads, int nW, int N, float eps,
                          float scale, int dtype, master_stream_t stream) {
    REQUIRE(x && wqkv && bqkv && wproj && bproj && bias64 && src && out, "attn_block: null operand");
    REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), "attn_block: ln_gamma and ln_beta come together");
    REQUIRE(B > 0 && T > 0 && nW > 0 && N > 0 && N <= 64 && x != out, "attn_block: bad dims (or in-place call)");
    if (dtype != MASTER_BF16) { master_set_error("attn_block: fused kernel is bf16-only"); return MASTER_ERR_UNSUPPORTED; }
    REQUIRE(al16(x) && al16(wqkv) && al16(wproj) && al16(out) && al16(bias64), "attn_block: operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_ATTN_BLOCK, s);
    const int rc = attn_block_bf16((const bf16_t*)x, ln_gamma, ln_beta, (const bf16_t*)wqkv, bqkv, (const bf16_t*)wproj, bproj,
                                   bias64, src, region, (bf16_t*)out, B, T, C, heads, nW, N, eps, scale, s);
    if (rc == MASTER_ERR_UNSUPPORTED) master_set_error("attn_block: C=%d heads=%d not built (128 / 4)", C, heads);
    return rc;
}

int master_layernorm_fwd(const void* x, const void* gamma, const void* beta, void* out, int M, int C, float eps,
                         int dtype, master_stream_t stream) {
    REQUIRE(x && gamma && beta && out, "layernorm: null operand");
    REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 1024, "layernorm: need C%%4==0 and C<=1024 (C=%d)", C);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_LAYERNORM, s);
    if (dtype == MASTER_F32) return layernorm_launch<float>(x, gamma, beta, out, M, C, eps, 0, 0, 0, C, s);
    if (dtype == MASTER_BF16) return layernorm_launch<bf16_t>(x, gamma, beta, out, M, C, eps, 0, 0, 0, C, s);
    master_set_error("layernorm: bad dtype %d", dtype);
    return MASTER_ERR_INVALID;
}

int master_instnorm_fwd(const void* x, void* out, const void* pad_row_in, void* pad_row_out, int B, int T, int C,
                        int n_pad, float eps, int repeat, int dtype, master_stream_t stream) {
    REQUIRE(x && out, "instnorm: null operand");
    REQUIRE(B > 0 && T > 0 && C > 0 && C % 32 == 0, "instnorm: need C%%32==0 (C=%d)", C);
    REQUIRE(n_pad >= 0 && (n_pad == 0 || pad_row_in), "instnorm: n_pad>0 needs pad_row_in");
    REQUIRE(repeat == 1 || (repeat == 2 && n_pad == 0), "instnorm: repeat must be 1, or 2 without pad rows");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_INSTNORM, s);
    if (dtype == MASTER_F32) return instnorm_launch<float>(x, out, pad_row_in, pad_row_out, B, T, C, n_pad, eps, repeat, s);
    if (dtype == MASTER_BF16) return instnorm_launch<bf16_t>(x, out, pad_row_in, pad_row_out, B, T, C, n_pad, eps, repeat, s);
    master_set_error("instnorm: bad dtype %d", dtype);
    return MASTER_ERR_INVALID;
}

int master_patch_embed_fwd(const float* img, const void* w, const void* bias, const void* gamma, const void* beta,
                           void* out, int B, int H, int W, int E, float eps, int dtype, master_stream_t stream) {
    REQUIRE(img && w && bias && gamma && beta && out, "patch_embed: null operand");
    REQUIRE(B > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0, "patch_embed: H,W must be multiples of 4");
    if (E != 128 && E != 96) { master_set_error("patch_embed: embed dim %d not built (128 = swin_B, 96 = swin_T / swin_S)", E); return MASTER_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_PATCH_EMBED, s);
    if (dtype == MASTER_F32) return patch_embed_launch<float>(img, w, bias, gamma, beta, out, B, H, W, E, eps, s);
    if (dtype == MASTER_BF16 && E != 128) { master_set_error("patch_embed(bf16): the matrix-core kernel is built for embed 128 (swin_B); embed %d runs in the fp32 mode", E); return MASTER_ERR_UNSUPPORTED; }
    if (dtype == MASTER_BF16)      // matrix-core version (bf16 operands, fp32 accumulate and LayerNorm)
        return patch_embed_bf16(img, (const float*)w, (const float*)bias, (const float*)gamma, (const float*)beta, (bf16_t*)out, B, H, W, eps, s);
    master_set_error("patch_embed: bad dtype %d", dtype);
    return MASTER_ERR_INVALID;
}

int master_patch_merge_ln_fwd(const void* x, const void* gamma, const void* beta, void* out, int B, int H, int W,
                              int C, float eps, int dtype, master_stream_t stream) {
    REQUIRE(x && gamma && beta && out, "patch_merge: null operand");
    REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "patch_merge: H,W must be even (pad-to-even not needed for /8 images)");
    REQUIRE(C % 4 == 0 && 4 * C <= 1024, "patch_merge: need C%%4==0 and 4C<=1024 (C=%d)", C);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_PATCH_MERGE, s);
    const int M = B * (H / 2) * (W / 2);
    if (dtype == MASTER_F32) return layernorm_launch<float>(x, gamma, beta, out, M, 4 * C, eps, 1, H, W, C, s);
    if (dtype == MASTER_BF16) return layernorm_launch<bf16_t>(x, gamma, beta, out, M, 4 * C, eps, 1, H, W, C, s);
    master_set_error("patch_merge: bad dtype %d", dtype);
    return MASTER_ERR_INVALID;
}

int master_convert(const void* src, void* dst, size_t n, int src_dtype, int dst_dtype, master_stream_t stream) {
    REQUIRE(src && dst, "convert: null operand");
    REQUIRE((src_dtype == MASTER_F32 || src_dtype == MASTER_BF16) && (dst_dtype == MASTER_F32 || dst_dtype == MASTER_BF16), "convert: bad dtype");
    if (n == 0) return MASTER_OK;
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(MASTER_K_CONVERT, s);
    return convert_launch(src, dst, n, src_dtype, dst_dtype, s);
}

}  // extern "C"
