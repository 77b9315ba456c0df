// Layout converters between the reference's torch-native layouts (OIHW weights, NCHW activations,
// NCHW-flatten linear5 columns; SURVEY.md §8(b) "Persistence") and the device layouts (OHWI, NHWC).
#include "common.h"

using namespace goalnet;

namespace {

// [B][R][C] -> [B][C][R] through a padded 32x32 LDS tile; both sides coalesced.
__global__ __launch_bounds__(256) void transpose_inner_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             int64_t R, int64_t C, int64_t tiles_c) {
    __shared__ float tile[32][33];
    const int64_t b = blockIdx.z;
    const int64_t tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const float* s = src + b * R * C;
    float* d = dst + b * R * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t r = tr * 32 + ty + 8 * i, c = tc * 32 + tx;
        if (r < R && c < C) tile[ty + 8 * i][tx] = s[r * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t c = tc * 32 + ty + 8 * i, r = tr * 32 + tx;
        if (r < R && c < C) d[c * R + r] = tile[tx][ty + 8 * i];
    }
}

// wt[ci][8 - t][co] = w[co][t][ci]: per tap a [Cout][Cin] -> [Cin][Cout] transpose through a 32 x 33 LDS tile, coalesced on both
// sides (the element-per-thread gather it replaces read one 4-byte word per cache line: 9.6 us for conv3's 1.2 M weights, in
// every backward of the 10-frame loop). blockIdx.y = tap.
__global__ __launch_bounds__(256) void weight_flip_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin) {
    __shared__ float tile[32][33];
    const int t = blockIdx.y;
    const int tiles_ci = (Cin + 31) / 32;
    const int co0 = (blockIdx.x / tiles_ci) * 32, ci0 = (blockIdx.x % tiles_ci) * 32;
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
    for (int r = r0; r < 32; r += 8)
        tile[r][c] = (co0 + r < Cout && ci0 + c < Cin) ? w[((int64_t)(co0 + r) * 9 + t) * Cin + ci0 + c] : 0.f;
    __syncthreads();
#pragma unroll
    for (int r = r0; r < 32; r += 8)
        if (ci0 + r < Cin && co0 + c < Cout) wt[((int64_t)(ci0 + r) * 9 + (8 - t)) * Cout + co0 + c] = tile[c][r];
}

// two weight tensors in one launch (the 10-frame step flips conv3's and conv2's weights in every backward): blocks
// [0, tiles_a) serve the first, the rest the second
__global__ __launch_bounds__(256) void weight_flip2_kernel(const float* __restrict__ wa, float* __restrict__ wta, int CoutA, int CinA, int tiles_a,
                                                          const float* __restrict__ wb, float* __restrict__ wtb, int CoutB, int CinB) {
    __shared__ float tile[32][33];
    const bool first = (int)blockIdx.x < tiles_a;
    const float* w = first ? wa : wb;
    float* wt = first ? wta : wtb;
    const int Cout = first ? CoutA : CoutB, Cin = first ? CinA : CinB;
    const int b = first ? blockIdx.x : blockIdx.x - tiles_a;
    const int t = blockIdx.y;
    const int tiles_ci = (Cin + 31) / 32;
    const int co0 = (b / tiles_ci) * 32, ci0 = (b % tiles_ci) * 32;
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
#pragma unroll
    for (int r = r0; r < 32; r += 8)
        tile[r][c] = (co0 + r < Cout && ci0 + c < Cin) ? w[((int64_t)(co0 + r) * 9 + t) * Cin + ci0 + c] : 0.f;
    __syncthreads();
#pragma unroll
    for (int r = r0; r < 32; r += 8)
        if (ci0 + r < Cin && co0 + c < Cout) wt[((int64_t)(ci0 + r) * 9 + (8 - t)) * Cout + co0 + c] = tile[c][r];
}

}  // namespace

extern "C" {

int goalnet_conv3x3_weight_flip2(const float* wa, float* wta, int CoutA, int CinA, const float* wb, float* wtb, int CoutB, int CinB, void* stream) {
    GN_REQUIRE(wa && wta && wb && wtb, GOALNET_E_NULL, "conv3x3_weight_flip2: null pointer");
    GN_REQUIRE(CoutA > 0 && CinA > 0 && CoutB > 0 && CinB > 0 && (int64_t)CoutA * CinA * 9 < (1ll << 30) && (int64_t)CoutB * CinB * 9 < (1ll << 30),
               GOALNET_E_SHAPE, "conv3x3_weight_flip2: bad dims");
    const int ta = ((CoutA + 31) / 32) * ((CinA + 31) / 32), tb = ((CoutB + 31) / 32) * ((CinB + 31) / 32);
    hipLaunchKernelGGL(weight_flip2_kernel, dim3(ta + tb, 9), dim3(256), 0, (hipStream_t)stream, wa, wta, CoutA, CinA, ta, wb, wtb, CoutB, CinB);
    GN_LAUNCH_CHECK("conv3x3_weight_flip2");
    return 0;
}

int goalnet_transpose_inner(const float* src, float* dst, int64_t B, int64_t R, int64_t C, void* stream) {
    GN_REQUIRE(src && dst, GOALNET_E_NULL, "transpose_inner: null pointer");
    GN_REQUIRE(B > 0 && R > 0 && C > 0 && B <= 65535, GOALNET_E_SHAPE, "transpose_inner: bad dims (B <= 65535)");
    const int64_t tiles_r = (R + 31) / 32, tiles_c = (C + 31) / 32;
    GN_REQUIRE(tiles_r * tiles_c < (1ll << 31), GOALNET_E_SHAPE, "transpose_inner: too many tiles");
    hipLaunchKernelGGL(transpose_inner_kernel, dim3((unsigned)(tiles_r * tiles_c), 1, (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, src, dst, R, C, tiles_c);
    GN_LAUNCH_CHECK("transpose_inner");
    return 0;
}

int goalnet_conv3x3_weight_flip(const float* w_ohwi, float* wt, int Cout, int Cin, void* stream) {
    GN_REQUIRE(w_ohwi && wt, GOALNET_E_NULL, "conv3x3_weight_flip: null pointer");
    GN_REQUIRE(Cout > 0 && Cin > 0 && (int64_t)Cout * Cin * 9 < (1ll << 30), GOALNET_E_SHAPE, "conv3x3_weight_flip: bad dims");
    const int tiles = ((Cout + 31) / 32) * ((Cin + 31) / 32);
    hipLaunchKernelGGL(weight_flip_kernel, dim3(tiles, 9), dim3(256), 0, (hipStream_t)stream, w_ohwi, wt, Cout, Cin);
    GN_LAUNCH_CHECK("conv3x3_weight_flip");
    return 0;
}

}  // extern "C"
