// The classifier-head variant of AVM ("CAVM" / "CVM" in the reference's report, Table 1) — an EXTENSION: in the reference
// it survives only as commented-out lines, which this file restates literally:
//
//     nn.LazyLinear(C) -> nn.Softmax(dim = 1)                       /root/reference/utils.py:255-257 (Sigmoid's alternative)
//     output = 4 * output + 1                                        utils.py:270 (live line, applies to either head)
//     criterion = nn.CrossEntropyLoss()                              main.py:69
//     loss = criterion(predictions, (labels - 1).long())             main.py:96, 189
//     predictions = torch.argmax(predictions, axis = 1) + 1          main.py:97, 190
//
// i.e. class scores s = 4 softmax(z) + 1 in (1, 5), cross entropy applied to s as if they were logits (a second softmax
// inside the loss), classes 1..C from float labels. Oracle: oracle/avm_ref.py (torch CPU ops); the reference itself
// holds no runnable form of it ("parity unpinned"). C <= 8; K (the 128 fusion features) < 1024.
#include "common.h"

using namespace goalnet;

namespace {

constexpr int CMAX = 8;

// one wave per row: C dot products of length K, softmax over the C logits, s = 4 p + 1
__global__ __launch_bounds__(256) void cls_head_fwd_kernel(const float* __restrict__ h, int64_t ldh, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ logits,
                                                          float* __restrict__ out, int N, int K, int C) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float z[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        float acc = 0.f;
        if (c < C)
            for (int k = lane; k < K; k += 64) acc = fmaf(h[(int64_t)n * ldh + k], w[c * K + k], acc);
        z[c] = wave_sum(acc) + (c < C ? b[c] : 0.f);
    }
    if (lane == 0) {
        float mx = z[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, z[c]);
        float e[CMAX], sum = 0.f;
        for (int c = 0; c < C; ++c) { e[c] = expf(z[c] - mx); sum += e[c]; }
        for (int c = 0; c < C; ++c) {
            if (logits) logits[(int64_t)n * C + c] = z[c];
            out[(int64_t)n * C + c] = 4.f * (e[c] / sum) + 1.f;
        }
    }
}

// nn.CrossEntropyLoss()(s (N, C), (labels - 1).long()): mean over rows of logsumexp(s_n) - s_n[label_n - 1]; one block, fixed
// summation order (fp64). ds (optional) = (softmax(s_n) - onehot) / N.
__global__ __launch_bounds__(1024) void cross_entropy_kernel(const float* __restrict__ s, const float* __restrict__ labels,
                                                            float* __restrict__ loss, float* __restrict__ ds, int N, int C) {
    __shared__ double part[1024];
    double acc = 0.0;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const float* r = s + (int64_t)n * C;
        int y = (int)(labels[n] - 1.f);                     // (labels - 1).long(): truncation toward zero
        y = y < 0 ? 0 : (y >= C ? C - 1 : y);               // torch raises for classes outside [0, C); here they are clamped (the labels are device memory: no host check)
        float mx = r[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, r[c]);
        float sum = 0.f;
        for (int c = 0; c < C; ++c) sum += expf(r[c] - mx);
        const float lse = mx + logf(sum);
        acc += (double)(lse - r[y]);
        if (ds)
            for (int c = 0; c < C; ++c) ds[(int64_t)n * C + c] = (expf(r[c] - lse) - (c == y ? 1.f : 0.f)) / (float)N;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int o = blockDim.x >> 1; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss) loss[0] = (float)(part[0] / (double)N);
}

// dz_n = p (.) (g - <g, p>) with p = (s - 1) / 4 and g = 4 ds: backward of s = 4 softmax(z) + 1
__device__ __forceinline__ void dz_of(const float* __restrict__ ds, const float* __restrict__ s, int C, float (&dz)[CMAX]) {
    float p[CMAX], dot = 0.f;
    for (int c = 0; c < C; ++c) { p[c] = (s[c] - 1.f) * 0.25f; dot = fmaf(4.f * ds[c], p[c], dot); }
    for (int c = 0; c < C; ++c) dz[c] = p[c] * (4.f * ds[c] - dot);
}

__global__ __launch_bounds__(256) void cls_head_bwd_dh_kernel(const float* __restrict__ ds, const float* __restrict__ s,
                                                             const float* __restrict__ w, const float* __restrict__ mult, int64_t ldmult,
                                                             float* __restrict__ dh, int64_t lddh, int N, int K, int C) {
    const int64_t total = (int64_t)N * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / K;
        const int k = (int)(i - n * K);
        float dz[CMAX];
        dz_of(ds + n * C, s + n * C, C, dz);
        float v = 0.f;
        for (int c = 0; c < C; ++c) v = fmaf(dz[c], w[c * K + k], v);
        if (mult) v *= mult[n * ldmult + k];
        dh[n * lddh + k] = v;
    }
}

// dw[c][k] = sum_n dz[n][c] h[n][k], db[c] = sum_n dz[n][c]; one thread per (c, k) / (c, bias), rows in order (deterministic)
__global__ __launch_bounds__(256) void cls_head_bwd_dw_kernel(const float* __restrict__ ds, const float* __restrict__ s,
                                                             const float* __restrict__ h, int64_t ldh, float* __restrict__ dw,
                                                             float* __restrict__ db, int N, int K, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * (K + 1)) return;
    const int c = i / (K + 1), k = i - c * (K + 1);
    double acc = 0.0;
    for (int n = 0; n < N; ++n) {
        float dz[CMAX];
        dz_of(ds + (int64_t)n * C, s + (int64_t)n * C, C, dz);
        acc += (double)dz[c] * (k < K ? (double)h[(int64_t)n * ldh + k] : 1.0);
    }
    if (k < K) dw[c * K + k] = (float)acc; else db[c] = (float)acc;
}

// torch.argmax(s, axis = 1) + 1: the first maximal column, as a float class in 1..C
__global__ __launch_bounds__(256) void argmax_plus1_kernel(const float* __restrict__ s, float* __restrict__ cls, int N, int C) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    int best = 0;
    for (int c = 1; c < C; ++c) if (s[(int64_t)n * C + c] > s[(int64_t)n * C + best]) best = c;
    cls[n] = (float)(best + 1);
}

}  // namespace

extern "C" {

int goalnet_cls_head_fwd(const float* h, int64_t ldh, const float* w, const float* b, float* logits, float* out, int N, int K, int C,
                         void* stream) {
    GN_REQUIRE(h && w && b && out, GOALNET_E_NULL, "cls_head_fwd: null pointer");
    GN_REQUIRE(N > 0 && K > 0 && K < 1024 && C >= 2 && C <= CMAX, GOALNET_E_SHAPE, "cls_head_fwd: bad dims (K < 1024, 2 <= C <= 8)");
    hipLaunchKernelGGL(cls_head_fwd_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, h, ldh, w, b, logits, out, N, K, C);
    GN_LAUNCH_CHECK("cls_head_fwd");
    return 0;
}

int goalnet_cross_entropy(const float* scores, const float* labels, float* loss, float* dscores, int N, int C, void* stream) {
    GN_REQUIRE(scores && labels && (loss || dscores), GOALNET_E_NULL, "cross_entropy: null pointer");
    GN_REQUIRE(N > 0 && C >= 2 && C <= CMAX, GOALNET_E_SHAPE, "cross_entropy: bad dims (2 <= C <= 8)");
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, scores, labels, loss, dscores, N, C);
    GN_LAUNCH_CHECK("cross_entropy");
    return 0;
}

int goalnet_cls_head_bwd(const float* dscores, const float* scores, const float* h, int64_t ldh, const float* w, const float* mult,
                         int64_t ldmult, float* dh, int64_t lddh, float* dw, float* db, int N, int K, int C, void* stream) {
    GN_REQUIRE(dscores && scores && h && w && dh && dw && db, GOALNET_E_NULL, "cls_head_bwd: null pointer");
    GN_REQUIRE(N > 0 && K > 0 && K < 1024 && C >= 2 && C <= CMAX, GOALNET_E_SHAPE, "cls_head_bwd: bad dims (K < 1024, 2 <= C <= 8)");
    hipStream_t st = (hipStream_t)stream;
    int64_t blocks = ((int64_t)N * K + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cls_head_bwd_dh_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dscores, scores, w, mult, ldmult, dh, lddh, N, K, C);
    GN_LAUNCH_CHECK("cls_head_bwd.dh");
    hipLaunchKernelGGL(cls_head_bwd_dw_kernel, dim3((C * (K + 1) + 255) / 256), dim3(256), 0, st, dscores, scores, h, ldh, dw, db, N, K, C);
    GN_LAUNCH_CHECK("cls_head_bwd.dw");
    return 0;
}

int goalnet_argmax_plus1(const float* scores, float* classes, int N, int C, void* stream) {
    GN_REQUIRE(scores && classes, GOALNET_E_NULL, "argmax_plus1: null pointer");
    GN_REQUIRE(N > 0 && C >= 1, GOALNET_E_SHAPE, "argmax_plus1: bad dims");
    hipLaunchKernelGGL(argmax_plus1_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, scores, classes, N, C);
    GN_LAUNCH_CHECK("argmax_plus1");
    return 0;
}

}  // extern "C"
