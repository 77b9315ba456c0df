// Synthetic data and dropout-mask generators: the device twin of cvml_goalnet_amd/synth.py
// (counter-based splitmix64). Element i of stream `tensor_id` depends only on
// (seed, tensor_id, i), so any launch geometry produces the same bits as numpy.
#include "common.h"
#include "rng.h"

using namespace goalnet;

namespace {

__global__ __launch_bounds__(256) void fill_uniform_kernel(float* dst, int64_t n, uint64_t key, float lo, float span) {
    // numpy rounds the product, then the sum. hipcc's default -ffp-contract=fast would fuse `lo + span * u` into one
    // fma (and __fmul_rn/__fadd_rn are plain operators in HIP); fma(span, u, 0) is the rounded product and cannot
    // be contracted with the following add.
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = lo + __builtin_fmaf(span, unit24(key, i), 0.0f);
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(float* dst, int64_t n, uint64_t key, float p, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = unit24(key, i) >= p ? scale : 0.f;
}

unsigned grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" {

int goalnet_fill_uniform(float* dst, int64_t n, uint64_t seed, uint32_t tensor_id, float lo, float hi, void* stream) {
    GN_REQUIRE(dst, GOALNET_E_NULL, "fill_uniform: null pointer");
    GN_REQUIRE(n >= 0, GOALNET_E_SHAPE, "fill_uniform: negative count");
    if (n == 0) return 0;
    hipLaunchKernelGGL(fill_uniform_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dst, n,
                       stream_key(seed, tensor_id), lo, hi - lo);
    GN_LAUNCH_CHECK("fill_uniform");
    return 0;
}

int goalnet_dropout_mask(float* dst, int64_t n, uint64_t seed, uint32_t tensor_id, float p, void* stream) {
    GN_REQUIRE(dst, GOALNET_E_NULL, "dropout_mask: null pointer");
    GN_REQUIRE(n >= 0 && p >= 0.f && p < 1.f, GOALNET_E_SHAPE, "dropout_mask: bad count or p");
    if (n == 0) return 0;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dst, n,
                       stream_key(seed, tensor_id), p, 1.0f / (1.0f - p));
    GN_LAUNCH_CHECK("dropout_mask");
    return 0;
}

}  // extern "C"
