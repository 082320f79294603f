// Placeholder launchers for the bf16 throughput mode (replaced by gemm_bf16.hip / window_attn_bf16.hip).
#include "common.h"
int gemm_bf16_linear(const bf16_t*, const bf16_t*, const float*, const bf16_t*, const bf16_t*, bf16_t*, int, int, int,
                     int, int, int, int, hipStream_t) { master_set_error("linear: bf16 mode not built"); return MASTER_ERR_UNSUPPORTED; }
int gemm_bf16_conv3x3(const bf16_t*, const bf16_t*, const float*, void*, int, int, int, int, int, int, int, int,
                      hipStream_t) { master_set_error("conv3x3: bf16 mode not built"); return MASTER_ERR_UNSUPPORTED; }
int window_attn_bf16(const bf16_t*, const bf16_t*, const bf16_t*, const bf16_t*, int, int, int, const bf16_t*,
                     const bf16_t*, const bf16_t*, int, int, const float*, const int32_t*, const uint8_t*, bf16_t*,
                     bf16_t*, int, int, int, int, int, int, float, hipStream_t) { master_set_error("window_attn: bf16 mode not built"); return MASTER_ERR_UNSUPPORTED; }
