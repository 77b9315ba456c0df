// Frame pre-processing of the reference's loader, /root/reference/utils.py:274-292:
//     image = ((image - image.min()) / (image.max() - image.min() + 1e-7)).astype(np.float32)     # per decoded BGR frame, uint8 HWC
//     image = cv2.resize(image, (40, 40))                                                         # INTER_LINEAR on float32
//     np.transpose(np.array(frames), (0, 3, 1, 2))                                                # -> (N, 3, H, W), channel order BGR
// on the device, for frames that are already decoded (cv2.VideoCapture stays the reference's I/O).
// PARITY UNPINNED: OpenCV is not in the image, so neither the oracle restatement (oracle/preproc_ref.py) nor this kernel can
// be checked against cv2 here. Both follow OpenCV's published bilinear resize for float32 (imgproc/resize.cpp: half-pixel
// centres fx = (dx + 0.5) * scale - 0.5, floor, clamp to the border, horizontal pass then vertical pass, no antialiasing)
// and agree with each other bit for bit.
#include "common.h"

using namespace goalnet;

namespace {

// per-frame min / max of the uint8 pixels (all three channels together, as image.min() / image.max())
__global__ __launch_bounds__(256) void frame_minmax_kernel(const uint8_t* __restrict__ frames, int64_t frame_bytes, int32_t* __restrict__ minmax) {
    __shared__ int smn[4], smx[4];
    const uint8_t* f = frames + (int64_t)blockIdx.x * frame_bytes;
    int mn = 255, mx = 0;
    for (int64_t i = threadIdx.x; i < frame_bytes; i += 256) { const int v = f[i]; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const int a = __shfl_xor(mn, o, 64), b = __shfl_xor(mx, o, 64); mn = a < mn ? a : mn; mx = b > mx ? b : mx; }
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { mn = smn[w] < mn ? smn[w] : mn; mx = smx[w] > mx ? smx[w] : mx; }
        minmax[2 * blockIdx.x] = mn;
        minmax[2 * blockIdx.x + 1] = mx;
    }
}

// rounded product / sum without fma contraction (OpenCV's scalar path multiplies and adds separately)
__device__ __forceinline__ float mul_rn(float a, float b) { return __builtin_fmaf(a, b, 0.0f); }

__global__ __launch_bounds__(256) void frame_resize_kernel(const uint8_t* __restrict__ frames, const int32_t* __restrict__ minmax,
                                                          float* __restrict__ out, int N, int H0, int W0, int H, int W,
                                                          double scale_x, double scale_y) {
    const int64_t total = (int64_t)N * 3 * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int dx = (int)(i % W), dy = (int)((i / W) % H), c = (int)((i / ((int64_t)W * H)) % 3);
        const int64_t n = i / ((int64_t)3 * H * W);
        // source coordinates, resize.cpp: fx = (float)((dx + 0.5) * scale_x - 0.5); sx = floor(fx); fx -= sx; border clamp
        float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { sx = 0; fx = 0.f; }
        if (sx >= W0 - 1) { sx = W0 - 1; fx = 0.f; }
        float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        if (sy < 0) { sy = 0; fy = 0.f; }
        if (sy >= H0 - 1) { sy = H0 - 1; fy = 0.f; }
        const int sx1 = sx + 1 < W0 ? sx + 1 : W0 - 1, sy1 = sy + 1 < H0 ? sy + 1 : H0 - 1;
        const int mn = minmax[2 * n], mx = minmax[2 * n + 1];
        const double den = (double)(mx - mn) + 1e-7;                        // uint8 difference, then + 1e-7 in float64
        const uint8_t* f = frames + n * (int64_t)H0 * W0 * 3;
        auto px = [&](int yy, int xx) -> float { return (float)((double)(f[((int64_t)yy * W0 + xx) * 3 + c] - mn) / den); };
        const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
        const float h0 = mul_rn(px(sy, sx), a0) + mul_rn(px(sy, sx1), a1);   // horizontal pass on the two source rows
        const float h1 = mul_rn(px(sy1, sx), a0) + mul_rn(px(sy1, sx1), a1);
        out[i] = mul_rn(h0, b0) + mul_rn(h1, b1);                           // vertical pass
    }
}

}  // namespace

extern "C" {

int goalnet_frames_preprocess(const uint8_t* frames_hwc, int N, int H0, int W0, float* out_nchw, int H, int W,
                              int32_t* minmax, void* stream) {
    GN_REQUIRE(frames_hwc && out_nchw && minmax, GOALNET_E_NULL, "frames_preprocess: null pointer");
    GN_REQUIRE(N > 0 && H0 > 0 && W0 > 0 && H > 0 && W > 0, GOALNET_E_SHAPE, "frames_preprocess: non-positive dim");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(frame_minmax_kernel, dim3(N), dim3(256), 0, st, frames_hwc, (int64_t)H0 * W0 * 3, minmax);
    GN_LAUNCH_CHECK("frames_preprocess.minmax");
    const int64_t total = (int64_t)N * 3 * H * W;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    // cv2.resize: inv_scale = dsize / ssize (double), scale = 1 / inv_scale
    const double scale_x = 1.0 / ((double)W / (double)W0), scale_y = 1.0 / ((double)H / (double)H0);
    hipLaunchKernelGGL(frame_resize_kernel, dim3((unsigned)blocks), dim3(256), 0, st, frames_hwc, (const int32_t*)minmax, out_nchw,
                       N, H0, W0, H, W, scale_x, scale_y);
    GN_LAUNCH_CHECK("frames_preprocess.resize");
    return 0;
}

}  // extern "C"
