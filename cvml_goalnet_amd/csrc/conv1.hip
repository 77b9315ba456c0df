// VisBl conv1: Conv2d(3 -> 64, k3, stride 3, pad 3) + bias + ReLU, and its weight gradient.
// /root/reference/utils.py:151-152, 174-175.
//
// K = 27 per output: this is HBM/LDS-bound direct convolution, not matrix-core work.
// Stride = kernel = 3, so the 3x3x3 input patches of neighbouring outputs are disjoint: every input
// element is read exactly once. Input is the reference's NCHW frame tensor (read coalesced along W);
// output is NHWC for the implicit-GEMM blocks that follow.
#include <stdlib.h>
#include "common.h"

using namespace goalnet;

namespace {

constexpr int CO = 64, KP = 27;
constexpr int WG_PARTS = 768;   // blocks (= partial rows) of the weight-gradient kernel: three 576-thread blocks per CU (512: 1.23 ms, 768: 1.07, 1024: 1.20 at 1024 frames of 224x224)

__device__ __forceinline__ int conv1_out(int x) { return (x + 3) / 3 + 1; }

// patch element k = (kh*3 + kw)*3 + ci  (OHWI order)  of output pixel (n, oh, ow); 0 in the padding
__device__ __forceinline__ float patch_at(const float* __restrict__ x, int n, int oh, int ow, int k, int H, int W) {
    const int ci = k % 3, t = k / 3;
    const int kh = t / 3, kw = t - 3 * kh;
    const int ih = 3 * oh - 3 + kh, iw = 3 * ow - 3 + kw;
    if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
        return x[(((int64_t)n * 3 + ci) * H + ih) * W + iw];
    return 0.f;
}

// 256 threads: 64 pixels x 4 groups of 16 output channels
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       int N, int H, int W, int Ho, int Wo) {
    __shared__ __attribute__((aligned(16))) float ws[KP][CO];
    __shared__ float patch[64][KP + 1];
    const int tid = threadIdx.x;
    for (int i = tid; i < KP * CO; i += 256) {
        const int co = i / KP, k = i - co * KP;      // w is [co][kh][kw][ci]
        ws[k][co] = w[i];
    }
    const int64_t npix = (int64_t)N * Ho * Wo;
    const int pl = tid >> 2, cg = tid & 3;
    for (int64_t p0 = (int64_t)blockIdx.x * 64; p0 < npix; p0 += (int64_t)gridDim.x * 64) {
        __syncthreads();
        for (int i = tid; i < 64 * KP; i += 256) {
            const int pp = i / KP, k = i - pp * KP;
            const int64_t pix = p0 + pp;
            float v = 0.f;
            if (pix < npix) {
                const int ow = (int)(pix % Wo);
                const int oh = (int)((pix / Wo) % Ho);
                const int n = (int)(pix / ((int64_t)Wo * Ho));
                v = patch_at(x, n, oh, ow, k, H, W);
            }
            patch[pp][k] = v;
        }
        __syncthreads();
        const int64_t pix = p0 + pl;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = bias[cg * 16 + j];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const float xv = patch[pl][k];
            const float4* wp = reinterpret_cast<const float4*>(&ws[k][cg * 16]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 wv = wp[q];
                acc[4 * q + 0] = fmaf(xv, wv.x, acc[4 * q + 0]);
                acc[4 * q + 1] = fmaf(xv, wv.y, acc[4 * q + 1]);
                acc[4 * q + 2] = fmaf(xv, wv.z, acc[4 * q + 2]);
                acc[4 * q + 3] = fmaf(xv, wv.w, acc[4 * q + 3]);
            }
        }
        if (pix < npix) {
            float4* o = reinterpret_cast<float4*>(y + pix * CO + cg * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                o[q] = make_float4(fmaxf(acc[4 * q], 0.f), fmaxf(acc[4 * q + 1], 0.f), fmaxf(acc[4 * q + 2], 0.f), fmaxf(acc[4 * q + 3], 0.f));
        }
    }
}

// v2: one output ROW (n, oh) per iteration, like the weight gradient below: the 9 input row segments (3 channels x 3
// kernel rows) the row needs are contiguous in the NCHW frame and are staged coalesced, with the zero padding; v1 gathered
// 27 scalars per pixel with 64-bit divisions (1.05 ms at N = 1024, 224 x 224 against 0.42 ms of HBM time). Thread =
// (4 output channels, one of 16 pixels per pass); the 27-term chain runs in v1's order (bias, then k ascending): same bits.
// Measured: 1.07 -> 0.9-1.0 ms only — what is left is one global round trip per row between two barriers.
__global__ __launch_bounds__(256) void conv1_fwd_v2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          int N, int H, int W, int Ho, int Wo) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int XW = 3 * Wo;
    float* wsh = sm;                 // [KP][CO]: few registers -> 8 waves per SIMD hide the per-row load latency
    float* xs = sm + KP * CO;        // [9][XW]   row r = ci*3 + kh, element j <-> iw = j - 3
    const int tid = threadIdx.x;
    const int cg = tid & 15, pl = tid >> 4;
    for (int i = tid; i < KP * CO; i += 256) {
        const int co = i / KP, k = i - co * KP;      // w is [co][kh][kw][ci]
        wsh[k * CO + co] = w[i];
    }
    const float4 b4 = *reinterpret_cast<const float4*>(bias + cg * 4);
    const int nrows = N * Ho;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int n = row / Ho, oh = row - n * Ho;
        __syncthreads();
        for (int i = tid; i < 9 * XW; i += 256) {
            const int r = i / XW, j = i - r * XW;
            const int ci = r / 3, kh = r - 3 * ci;
            const int ih = 3 * oh - 3 + kh, iw = j - 3;
            float v = 0.f;
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = x[(((int64_t)n * 3 + ci) * H + ih) * W + iw];
            xs[i] = v;
        }
        __syncthreads();
        // hipcc otherwise vectorises two pixels together (v_pk_fma_f32), needs 256 VGPRs (one wave per SIMD) and splits the stores
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
        for (int ow = pl; ow < Wo; ow += 16) {
            float4 a = b4;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int ci = k % 3, t = k / 3, kh = t / 3, kw = t - 3 * kh;
                const float xv = xs[(ci * 3 + kh) * XW + 3 * ow + kw];
                const float4 wk = *reinterpret_cast<const float4*>(&wsh[k * CO + cg * 4]);
                a.x = fmaf(xv, wk.x, a.x); a.y = fmaf(xv, wk.y, a.y); a.z = fmaf(xv, wk.z, a.z); a.w = fmaf(xv, wk.w, a.w);
            }
            *reinterpret_cast<float4*>(y + ((int64_t)row * Wo + ow) * CO + cg * 4) =
                make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
        }
    }
}

// v3 = v2 with the next row's input prefetched: a thread's staging slots (row segment, column) are the same for every output
// row, so their offsets are computed once and the loads of row r + 1 are in flight while row r is computed (v2 paid one global
// round trip per row between two barriers: 1.0 ms against 0.35 ms of HBM time at N = 1024, 224 x 224). Same arithmetic order.
template <int NS>
__global__ __launch_bounds__(256) void conv1_fwd_v3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          int N, int H, int W, int Ho, int Wo) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int XW = 3 * Wo;
    float* wsh = sm;                 // [KP][CO]
    float* xs = sm + KP * CO;        // [9][XW]   row r = ci*3 + kh, element j <-> iw = j - 3
    const int tid = threadIdx.x;
    const int cg = tid & 15, pl = tid >> 4;
    for (int i = tid; i < KP * CO; i += 256) {
        const int co = i / KP, k = i - co * KP;      // w is [co][kh][kw][ci]
        wsh[k * CO + co] = w[i];
    }
    const float4 b4 = *reinterpret_cast<const float4*>(bias + cg * 4);
    // staging slots of this thread: element i = tid + 256 s of the [9][XW] image
    int soff[NS], skh[NS];           // offset inside the frame for oh = 0 (may be negative), kernel row; skh < 0: no element
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int i = tid + 256 * s;
        const int r = i / XW, j = i - r * XW;
        const int ci = r / 3, kh = r - 3 * ci, iw = j - 3;
        const bool ok = i < 9 * XW && (unsigned)iw < (unsigned)W;
        skh[s] = ok ? kh : -1;
        soff[s] = (ci * H + (kh - 3)) * W + iw;
    }
    const int nrows = N * Ho;
    float pre[NS];
    auto fetch = [&](int row) {
        const int n = row / Ho, oh = row - n * Ho;
        const float* f = x + (int64_t)n * 3 * H * W + (int64_t)3 * oh * W;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int ih = 3 * oh - 3 + skh[s];
            pre[s] = (skh[s] >= 0 && (unsigned)ih < (unsigned)H) ? f[soff[s]] : 0.f;
        }
    };
    int row = blockIdx.x;
    if (row < nrows) fetch(row);
    for (; row < nrows; row += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (tid + 256 * s < 9 * XW) xs[tid + 256 * s] = pre[s];
        if (row + (int)gridDim.x < nrows) fetch(row + gridDim.x);
        __syncthreads();
#pragma clang loop vectorize(disable) interleave(disable) unroll(disable)
        for (int ow = pl; ow < Wo; ow += 16) {
            float4 a = b4;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int ci = k % 3, t = k / 3, kh = t / 3, kw = t - 3 * kh;
                const float xv = xs[(ci * 3 + kh) * XW + 3 * ow + kw];
                const float4 wk = *reinterpret_cast<const float4*>(&wsh[k * CO + cg * 4]);
                a.x = fmaf(xv, wk.x, a.x); a.y = fmaf(xv, wk.y, a.y); a.z = fmaf(xv, wk.z, a.z); a.w = fmaf(xv, wk.w, a.w);
            }
            *reinterpret_cast<float4*>(y + ((int64_t)row * Wo + ow) * CO + cg * 4) =
                make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
        }
    }
}

// dW[co][k] = sum_pix dy[pix][co] * patch[pix][k]; db[co] = sum_pix dy[pix][co].
// thread = (co = tid & 63, kg = tid >> 6): k in [7*kg, 7*kg + 7); partial row per block, summed by a second kernel.
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ partial, int N, int H, int W, int Ho, int Wo) {
    __shared__ float dys[64][CO + 1];
    __shared__ float patch[64][KP + 1];
    const int tid = threadIdx.x;
    const int co = tid & 63, kg = tid >> 6;
    const int64_t npix = (int64_t)N * Ho * Wo;
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float accb = 0.f;
    for (int64_t p0 = (int64_t)blockIdx.x * 64; p0 < npix; p0 += (int64_t)gridDim.x * 64) {
        __syncthreads();
        for (int i = tid; i < 64 * CO; i += 256) {
            const int pp = i >> 6, c = i & 63;
            const int64_t pix = p0 + pp;
            dys[pp][c] = pix < npix ? dy[pix * CO + c] : 0.f;
        }
        for (int i = tid; i < 64 * KP; i += 256) {
            const int pp = i / KP, k = i - pp * KP;
            const int64_t pix = p0 + pp;
            float v = 0.f;
            if (pix < npix) {
                const int ow = (int)(pix % Wo);
                const int oh = (int)((pix / Wo) % Ho);
                const int n = (int)(pix / ((int64_t)Wo * Ho));
                v = patch_at(x, n, oh, ow, k, H, W);
            }
            patch[pp][k] = v;
        }
        if (tid < 64) patch[tid][KP] = 0.f;
        __syncthreads();
#pragma unroll 4
        for (int pp = 0; pp < 64; ++pp) {
            const float d = dys[pp][co];
            accb += d;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int k = 7 * kg + j;
                acc[j] = fmaf(d, patch[pp][k < KP ? k : KP], acc[j]);   // column KP is zero padding
            }
        }
    }
    float* row = partial + (int64_t)blockIdx.x * (CO * KP + CO);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int k = 7 * kg + j;
        if (k < KP) row[co * KP + k] = acc[j];
    }
    if (kg == 0) row[CO * KP + co] = accb;
}

// v2: one output ROW (n, oh) per iteration. The 9 input row segments (3 channels x 3 kernel rows) a whole output row
// needs are contiguous in the NCHW frame and are staged coalesced; v1 gathered 27 scattered scalars per pixel with
// 64-bit divisions and ran at 12.7 ms for 20 GFLOP.
__global__ __launch_bounds__(256) void conv1_wgrad_v2_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ partial, int N, int H, int W, int Ho, int Wo) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int XW = 3 * Wo;
    float* xs = sm;                 // [9][XW]   row r = ci*3 + kh, element j <-> iw = j - 3
    float* dys = sm + 9 * XW;       // [Wo][64]
    const int tid = threadIdx.x;
    const int co = tid & 63, kg = tid >> 6;
    int off[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int k = 7 * kg + j < KP ? 7 * kg + j : 0;
        const int ci = k % 3, t = k / 3, kh = t / 3, kw = t - 3 * kh;
        off[j] = (ci * 3 + kh) * XW + kw;
    }
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float accb = 0.f;
    const int nrows = N * Ho;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int n = row / Ho, oh = row - n * Ho;
        __syncthreads();
        for (int i = tid; i < 9 * XW; i += 256) {
            const int r = i / XW, j = i - r * XW;
            const int ci = r / 3, kh = r - 3 * ci;
            const int ih = 3 * oh - 3 + kh, iw = j - 3;
            float v = 0.f;
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = x[(((int64_t)n * 3 + ci) * H + ih) * W + iw];
            xs[i] = v;
        }
        const float4* src = reinterpret_cast<const float4*>(dy + (int64_t)row * Wo * CO);
        for (int i = tid; i < Wo * (CO / 4); i += 256) reinterpret_cast<float4*>(dys)[i] = src[i];
        __syncthreads();
        for (int ow = 0; ow < Wo; ++ow) {
            const float d = dys[ow * CO + co];
            accb += d;
            const float* xp = xs + 3 * ow;
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[j] = fmaf(d, xp[off[j]], acc[j]);
        }
    }
    float* prow = partial + (int64_t)blockIdx.x * (CO * KP + CO);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int k = 7 * kg + j;
        if (k < KP) prow[co * KP + k] = acc[j];
    }
    if (kg == 0) prow[CO * KP + co] = accb;
}

// v3 (Wo % 4 == 0): thread = (co, r = ci*3 + kh) owns the three taps kw = 0..2 of one staged input row segment, so the x
// values of four consecutive outputs are 12 consecutive floats = three ds_read_b128 (v2: seven scalar reads per output
// for seven unrelated taps): 133 LDS reads per row and thread instead of 608 — v2 was LDS-bound (1.9 ms against 0.4 ms of
// HBM time). Each tap still accumulates its products in output order, rows in the same order: same bits as v2.
__global__ __launch_bounds__(576) void conv1_wgrad_v3_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ partial, int N, int H, int W, int Ho, int Wo) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int XW = 3 * Wo;
    float* xs = sm;                 // [9][XW]
    float* dys = sm + 9 * XW;       // [Wo][64]
    const int tid = threadIdx.x;
    const int co = tid & 63, r = tid >> 6;          // r = ci*3 + kh  (0..8)
    float acc[3] = {0.f, 0.f, 0.f};
    float accb = 0.f;
    const int nrows = N * Ho;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int n = row / Ho, oh = row - n * Ho;
        __syncthreads();
        for (int i = tid; i < 9 * XW; i += 576) {
            const int rr = i / XW, j = i - rr * XW;
            const int ci = rr / 3, kh = rr - 3 * ci;
            const int ih = 3 * oh - 3 + kh, iw = j - 3;
            float v = 0.f;
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) v = x[(((int64_t)n * 3 + ci) * H + ih) * W + iw];
            xs[i] = v;
        }
        const float4* src = reinterpret_cast<const float4*>(dy + (int64_t)row * Wo * CO);
        for (int i = tid; i < Wo * (CO / 4); i += 576) reinterpret_cast<float4*>(dys)[i] = src[i];
        __syncthreads();
        const float4* xr = reinterpret_cast<const float4*>(xs + r * XW);
        for (int o4 = 0; o4 < Wo / 4; ++o4) {
            const float4 x0 = xr[3 * o4], x1 = xr[3 * o4 + 1], x2 = xr[3 * o4 + 2];
            const float xv[12] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = dys[(4 * o4 + i) * CO + co];
                if (r == 0) accb += d;
                acc[0] = fmaf(d, xv[3 * i + 0], acc[0]);
                acc[1] = fmaf(d, xv[3 * i + 1], acc[1]);
                acc[2] = fmaf(d, xv[3 * i + 2], acc[2]);
            }
        }
    }
    float* prow = partial + (int64_t)blockIdx.x * (CO * KP + CO);
    const int ci = r / 3, kh = r - 3 * ci;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) prow[co * KP + (kh * 3 + kw) * 3 + ci] = acc[kw];
    if (r == 0) prow[CO * KP + co] = accb;
}

// 16 columns x 16 part lanes per block; lane-group totals added in a fixed order (deterministic)
__global__ __launch_bounds__(256) void conv1_wgrad_reduce_kernel(const float* __restrict__ partial, int nparts,
                                                                float* __restrict__ dw, float* __restrict__ db) {
    __shared__ double sm[256];
    constexpr int ROW = CO * KP + CO;
    const int cl = threadIdx.x & 15, pg = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + cl;
    double a0 = 0.0, a1 = 0.0;
    if (i < ROW) {
        int p = pg;
        for (; p + 16 < nparts; p += 32) {
            a0 += (double)partial[(int64_t)p * ROW + i];
            a1 += (double)partial[(int64_t)(p + 16) * ROW + i];
        }
        for (; p < nparts; p += 16) a0 += (double)partial[(int64_t)p * ROW + i];
    }
    sm[pg * 16 + cl] = a0 + a1;
    __syncthreads();
    if (pg != 0 || i >= ROW) return;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += sm[k * 16 + cl];
    if (i < CO * KP) dw[i] = (float)s;
    else if (db) db[i - CO * KP] = (float)s;
}

}  // namespace

extern "C" {

int goalnet_conv1_fwd(const float* x_nchw, const float* w_ohwi, const float* bias, float* y_nhwc,
                      int N, int H, int W, void* stream) {
    GN_REQUIRE(x_nchw && w_ohwi && bias && y_nhwc, GOALNET_E_NULL, "conv1_fwd: null pointer");
    GN_REQUIRE(N > 0 && H > 0 && W > 0, GOALNET_E_SHAPE, "conv1_fwd: non-positive dim");
    GN_REQUIRE(aligned16(y_nhwc), GOALNET_E_ALIGN, "conv1_fwd: output must be 16-byte aligned");
    const int Ho = (H + 3) / 3 + 1, Wo = (W + 3) / 3 + 1;
    const int64_t npix = (int64_t)N * Ho * Wo;
    const size_t lds = (size_t)(KP * CO + 9 * 3 * Wo) * sizeof(float);
    const int ns = (9 * 3 * Wo + 255) / 256;                // staging slots per thread
    if (lds <= 64 * 1024 && aligned16(bias) && ns <= 12 && !getenv("GOALNET_CONV1_V1") && !getenv("GOALNET_CONV1_V2")) {
        int64_t blocks = (int64_t)N * Ho;
        if (blocks > 2048) blocks = 2048;
        const dim3 g((unsigned)blocks), b(256);
        hipStream_t st = (hipStream_t)stream;
        if (ns <= 3) hipLaunchKernelGGL(conv1_fwd_v3_kernel<3>, g, b, lds, st, x_nchw, w_ohwi, bias, y_nhwc, N, H, W, Ho, Wo);
        else if (ns <= 6) hipLaunchKernelGGL(conv1_fwd_v3_kernel<6>, g, b, lds, st, x_nchw, w_ohwi, bias, y_nhwc, N, H, W, Ho, Wo);
        else if (ns <= 9) hipLaunchKernelGGL(conv1_fwd_v3_kernel<9>, g, b, lds, st, x_nchw, w_ohwi, bias, y_nhwc, N, H, W, Ho, Wo);
        else hipLaunchKernelGGL(conv1_fwd_v3_kernel<12>, g, b, lds, st, x_nchw, w_ohwi, bias, y_nhwc, N, H, W, Ho, Wo);
    } else if (lds <= 64 * 1024 && aligned16(bias) && !getenv("GOALNET_CONV1_V1")) {
        int64_t blocks = (int64_t)N * Ho;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(conv1_fwd_v2_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x_nchw, w_ohwi, bias,
                           y_nhwc, N, H, W, Ho, Wo);
    } else {
        int64_t blocks = (npix + 63) / 64;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(conv1_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x_nchw, w_ohwi, bias,
                           y_nhwc, N, H, W, Ho, Wo);
    }
    GN_LAUNCH_CHECK("conv1_fwd");
    return 0;
}

size_t goalnet_conv1_wgrad_ws_bytes(int N, int H, int W) {
    (void)N; (void)H; (void)W;
    return (size_t)WG_PARTS * (CO * KP + CO) * sizeof(float);
}

int goalnet_conv1_wgrad(const float* x_nchw, const float* dy_nhwc, float* dw_ohwi, float* dbias,
                        void* ws, size_t ws_bytes, int N, int H, int W, void* stream) {
    GN_REQUIRE(x_nchw && dy_nhwc && dw_ohwi && ws, GOALNET_E_NULL, "conv1_wgrad: null pointer");
    GN_REQUIRE(N > 0 && H > 0 && W > 0, GOALNET_E_SHAPE, "conv1_wgrad: non-positive dim");
    GN_REQUIRE(ws_bytes >= goalnet_conv1_wgrad_ws_bytes(N, H, W), GOALNET_E_WORKSPACE, "conv1_wgrad: workspace too small");
    const int Ho = (H + 3) / 3 + 1, Wo = (W + 3) / 3 + 1;
    const size_t lds = (size_t)(9 * 3 * Wo + Wo * CO) * sizeof(float);
    if (lds <= 64 * 1024 && Wo % 4 == 0 && !getenv("GOALNET_CONV1_V1"))
        hipLaunchKernelGGL(conv1_wgrad_v3_kernel, dim3(WG_PARTS), dim3(576), lds, (hipStream_t)stream, x_nchw, dy_nhwc,
                           (float*)ws, N, H, W, Ho, Wo);
    else if (lds <= 64 * 1024)
        hipLaunchKernelGGL(conv1_wgrad_v2_kernel, dim3(WG_PARTS), dim3(256), lds, (hipStream_t)stream, x_nchw, dy_nhwc,
                           (float*)ws, N, H, W, Ho, Wo);
    else
        hipLaunchKernelGGL(conv1_wgrad_kernel, dim3(WG_PARTS), dim3(256), 0, (hipStream_t)stream, x_nchw, dy_nhwc,
                           (float*)ws, N, H, W, Ho, Wo);
    GN_LAUNCH_CHECK("conv1_wgrad");
    hipLaunchKernelGGL(conv1_wgrad_reduce_kernel, dim3((CO * KP + CO + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                       (const float*)ws, WG_PARTS, dw_ohwi, dbias);
    GN_LAUNCH_CHECK("conv1_wgrad.reduce");
    return 0;
}

}  // extern "C"
