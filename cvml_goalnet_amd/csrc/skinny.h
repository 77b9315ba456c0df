// M <= 16 row linear layers as weight-streaming kernels (skinny.hip); dispatched from the goalnet_linear_* entry points.
#pragma once
#include "gemm_common.h"

namespace goalnet {

constexpr int SKINNY_MAX_M = 16;

size_t skinny_fwd_ws_bytes(int M, int64_t K, int J);
int skinny_linear_fwd(const float* x, int64_t ldx, const float* scale, const float* shift, int bnC, const float* w,
                      const EpiP& efinal, int M, int64_t K, int J, void* ws, hipStream_t st);
int skinny_linear_dx(const float* dy, int64_t lddy, const float* w, const float* mult, int64_t ldmult, float* dx, int64_t lddx,
                     int M, int64_t K, int J, hipStream_t st);
int skinny_linear_dw(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* scale, const float* shift, int bnC,
                     float* dw, float* db, int M, int64_t K, int J, hipStream_t st);

}  // namespace goalnet
