// What the reference runs after the hot path for every video in every epoch, on the device:
//   importance rounding + expansion + per-clip sums   /root/reference/utils.py:608-613, 396-410, 445-463
//   0/1 knapsack over the clips                       /root/reference/utils.py:465-510, 633-635
//   summary mask                                      /root/reference/utils.py:637-641
//   F-score against the annotators                    /root/reference/utils.py:552-580
// Integer work, bit-exact with the reference's Python: int64 sums, the same DP recurrence and back-tracking rule, and
// the same sequence of IEEE double operations for precision / recall / F (one division, 2*p*r/(p+r), a running sum).
// Quirks kept: clip sums and weights use the end-exclusive slice [a:b), the mask the
// end-inclusive range [a, b]. The reference does this in pure Python (a 200 x 15 000 DP table of list-of-lists).
#include "common.h"

using namespace goalnet;

namespace {

// importance of raw frame f: torch.round (half to even) -> int8, then expand_array (repeat, truncate, pad with the last)
__device__ __forceinline__ int importance_at(const float* __restrict__ pred, int n_sampled, int skip, int full_n, int f) {
    int i = n_sampled == full_n ? f : f / skip;
    if (i >= n_sampled) i = n_sampled - 1;
    return (int)(signed char)(int)rintf(pred[i]);
}

// one block per clip: value = sum of importances over [a:b) clamped like a Python slice, length = len of that slice
__global__ __launch_bounds__(256) void clip_info_kernel(const float* __restrict__ pred, int n_sampled, int skip, int full_n,
                                                       const int32_t* __restrict__ cps, int n_clips,
                                                       int64_t* __restrict__ values, int32_t* __restrict__ lengths) {
    __shared__ int64_t red[4];
    const int c = blockIdx.x;
    int a = cps[2 * c], b = cps[2 * c + 1];
    if (a < 0) a = a + full_n < 0 ? 0 : a + full_n;            // Python slice semantics: negative bounds count from the end
    if (b < 0) b = b + full_n < 0 ? 0 : b + full_n;
    a = a > full_n ? full_n : a;
    b = b > full_n ? full_n : b;
    const int len = b > a ? b - a : 0;
    int64_t s = 0;
    for (int f = a + threadIdx.x; f < a + len; f += 256) s += importance_at(pred, n_sampled, skip, full_n, f);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        values[c] = red[0] + red[1] + red[2] + red[3];
        lengths[c] = len;
    }
}

// K[i][w] of utils.py:480-491, one row per iteration, the row's columns spread over the block; then the back-tracking of
// utils.py:493-508 by thread 0. K: (n + 1) x (cap + 1) int64 in global memory (L2-resident: <= 24 MB).
__global__ __launch_bounds__(1024) void knapsack_kernel(const int64_t* __restrict__ values, const int32_t* __restrict__ lengths,
                                                       int weight_scale, const int32_t* __restrict__ weights_in, int n, int cap,
                                                       int64_t* __restrict__ K, int32_t* __restrict__ selected) {
    const int W = cap + 1;
    for (int w = threadIdx.x; w < W; w += 1024) K[w] = 0;
    for (int i = threadIdx.x; i < n; i += 1024) selected[i] = 0;
    __syncthreads();
    for (int i = 1; i <= n; ++i) {
        const int64_t v = values[i - 1];
        const int wt = weights_in ? weights_in[i - 1] : lengths[i - 1] * weight_scale;
        const int64_t* prev = K + (int64_t)(i - 1) * W;
        int64_t* cur = K + (int64_t)i * W;
        for (int w = threadIdx.x; w < W; w += 1024) {
            int64_t r = prev[w];
            if (w == 0) r = 0;
            else if (wt <= w) { const int64_t t = v + prev[w - wt]; r = t > r ? t : r; }
            cur[w] = r;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int64_t res = K[(int64_t)n * W + cap];
        int w = cap;
        for (int i = n; i > 0; --i) {
            if (res <= 0) break;
            if (res == K[(int64_t)(i - 1) * W + w]) continue;
            selected[i - 1] = 1;
            res -= values[i - 1];
            w -= weights_in ? weights_in[i - 1] : lengths[i - 1] * weight_scale;
        }
    }
}

// mask[a .. b] = 1 for the selected clips (end inclusive); a frame outside the video sets status (numpy: IndexError)
__global__ __launch_bounds__(256) void summary_mask_kernel(const int32_t* __restrict__ cps, const int32_t* __restrict__ selected,
                                                          int n_clips, int full_n, uint8_t* __restrict__ mask, int32_t* __restrict__ status) {
    const int c = blockIdx.x;
    if (!selected[c]) return;
    const int a = cps[2 * c], b = cps[2 * c + 1];
    if (threadIdx.x == 0 && b >= a && (a < -full_n || b >= full_n)) atomicOr(status, 1);
    for (int f = a + threadIdx.x; f <= b; f += 256) {
        const int g = f < 0 ? f + full_n : f;                      // numpy wraps negative indices
        if (g >= 0 && g < full_n) mask[g] = 1;
    }
}

// counts[u] = {sum(S and G_u), sum(G_u)}; counts[n_users] = {sum(S), 0}
__global__ __launch_bounds__(256) void fscore_counts_kernel(const uint8_t* __restrict__ gd, const uint8_t* __restrict__ mask, int n_users,
                                                           int full_n, int64_t* __restrict__ counts) {
    __shared__ int64_t red[4][2];
    const int u = blockIdx.x;
    int64_t ov = 0, sg = 0;
    if (u < n_users) {
        const uint8_t* g = gd + (int64_t)u * full_n;
        for (int f = threadIdx.x; f < full_n; f += 256) { const int gv = g[f]; ov += (gv != 0 && mask[f] != 0) ? 1 : 0; sg += gv; }
    } else {
        for (int f = threadIdx.x; f < full_n; f += 256) ov += mask[f];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ov += __shfl_xor(ov, o, 64); sg += __shfl_xor(sg, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = ov; red[threadIdx.x >> 6][1] = sg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        counts[2 * u] = red[0][0] + red[1][0] + red[2][0] + red[3][0];
        counts[2 * u + 1] = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    }
}

// utils.py:566-580 in the reference's order of double operations
__global__ void fscore_final_kernel(const int64_t* __restrict__ counts, int n_users, double* __restrict__ out) {
    const int64_t s_sum = counts[2 * n_users];
    double total = 0.0, best = 0.0;
    for (int u = 0; u < n_users; ++u) {
        const int64_t ov = counts[2 * u], g_sum = counts[2 * u + 1];
        const double precision = s_sum != 0 ? (double)ov / (double)s_sum : 0.0;
        const double recall = g_sum != 0 ? (double)ov / (double)g_sum : 0.0;
        const double pr = precision + recall;
        const double f = pr != 0.0 ? 2.0 * precision * recall / pr : 0.0;
        total += f;
        best = (u == 0 || f > best) ? f : best;
    }
    out[0] = total / (double)n_users;
    out[1] = best;
}

size_t align256(size_t b) { return (b + 255) / 256 * 256; }

}  // namespace

extern "C" {

size_t goalnet_knapsack_ws_bytes(int n_items, int capacity_scaled) {
    if (n_items < 0 || capacity_scaled < 0) return 0;
    return align256((size_t)(n_items + 1) * (size_t)(capacity_scaled + 1) * sizeof(int64_t));
}

int goalnet_knapsack(const int64_t* values, const int32_t* weights_scaled, int n_items, int capacity_scaled,
                     int32_t* selected, void* ws, size_t ws_bytes, void* stream) {
    GN_REQUIRE(values && weights_scaled && selected && ws, GOALNET_E_NULL, "knapsack: null pointer");
    GN_REQUIRE(n_items >= 1 && capacity_scaled >= 0, GOALNET_E_SHAPE, "knapsack: need n_items >= 1, capacity >= 0");
    GN_REQUIRE(ws_bytes >= goalnet_knapsack_ws_bytes(n_items, capacity_scaled), GOALNET_E_WORKSPACE, "knapsack: workspace too small");
    hipLaunchKernelGGL(knapsack_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, values, (const int32_t*)nullptr, 0, weights_scaled,
                       n_items, capacity_scaled, (int64_t*)ws, selected);
    GN_LAUNCH_CHECK("knapsack");
    return 0;
}

int goalnet_fscore(const uint8_t* gd, const uint8_t* mask, int n_users, int full_n_frames, double* fscore, int64_t* counts,
                   void* stream) {
    GN_REQUIRE(gd && mask && fscore && counts, GOALNET_E_NULL, "fscore: null pointer");
    GN_REQUIRE(n_users >= 1 && full_n_frames >= 1, GOALNET_E_SHAPE, "fscore: need n_users >= 1 and frames >= 1");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(fscore_counts_kernel, dim3(n_users + 1), dim3(256), 0, st, gd, mask, n_users, full_n_frames, counts);
    GN_LAUNCH_CHECK("fscore.counts");
    hipLaunchKernelGGL(fscore_final_kernel, dim3(1), dim3(1), 0, st, counts, n_users, fscore);
    GN_LAUNCH_CHECK("fscore.final");
    return 0;
}

size_t goalnet_postprocess_ws_bytes(int n_clips, int capacity_scaled, int n_users) {
    if (n_clips < 0 || capacity_scaled < 0 || n_users < 0) return 0;
    return goalnet_knapsack_ws_bytes(n_clips, capacity_scaled) + align256((size_t)(n_users + 1) * 2 * sizeof(int64_t));
}

int goalnet_postprocess(const float* pred, int n_sampled, int skip_frames, int full_n_frames, const int32_t* change_points,
                        int n_clips, int weight_scale, int capacity_scaled, const uint8_t* gd, int n_users, uint8_t* mask,
                        int32_t* selected, int64_t* clip_values, int32_t* clip_lengths, double* fscore, int32_t* status,
                        void* ws, size_t ws_bytes, void* stream) {
    GN_REQUIRE(pred && change_points && mask && selected && clip_values && clip_lengths && status && ws, GOALNET_E_NULL,
               "postprocess: null pointer");
    GN_REQUIRE((gd == nullptr) == (fscore == nullptr), GOALNET_E_NULL, "postprocess: gd and fscore must both be set or both NULL");
    GN_REQUIRE(n_sampled >= 1 && skip_frames >= 1 && full_n_frames >= 1 && n_clips >= 1 && weight_scale >= 0 && capacity_scaled >= 0 &&
               (gd == nullptr || n_users >= 1), GOALNET_E_SHAPE, "postprocess: bad dims");
    GN_REQUIRE(ws_bytes >= goalnet_postprocess_ws_bytes(n_clips, capacity_scaled, gd ? n_users : 0), GOALNET_E_WORKSPACE,
               "postprocess: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(mask, 0, (size_t)full_n_frames, st);
    if (e == hipSuccess) e = hipMemsetAsync(status, 0, sizeof(int32_t), st);
    if (e != hipSuccess) { set_error("postprocess: memset failed: %s", hipGetErrorString(e)); return (int)e; }
    hipLaunchKernelGGL(clip_info_kernel, dim3(n_clips), dim3(256), 0, st, pred, n_sampled, skip_frames, full_n_frames, change_points,
                       n_clips, clip_values, clip_lengths);
    GN_LAUNCH_CHECK("postprocess.clip_info");
    hipLaunchKernelGGL(knapsack_kernel, dim3(1), dim3(1024), 0, st, (const int64_t*)clip_values, (const int32_t*)clip_lengths, weight_scale,
                       (const int32_t*)nullptr, n_clips, capacity_scaled, (int64_t*)ws, selected);
    GN_LAUNCH_CHECK("postprocess.knapsack");
    hipLaunchKernelGGL(summary_mask_kernel, dim3(n_clips), dim3(256), 0, st, change_points, (const int32_t*)selected, n_clips, full_n_frames,
                       mask, status);
    GN_LAUNCH_CHECK("postprocess.mask");
    if (gd) {
        int64_t* counts = (int64_t*)((char*)ws + goalnet_knapsack_ws_bytes(n_clips, capacity_scaled));
        return goalnet_fscore(gd, mask, n_users, full_n_frames, fscore, counts, stream);
    }
    return 0;
}

}  // extern "C"
