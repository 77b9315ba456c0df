// nn.MSELoss()(pred (n,1), labels (n,)) with its (n,n) broadcast (/root/reference/main.py:68, 191) and its gradient, as the work of
// ONE block of 256 threads: goalnet_mse_bcast's kernel (small.hip) and the tail of the fused fusion-MLP forward (mlp.hip) run the
// same code, so the loss and dL/dpred do not depend on which of the two produced them.
//   loss = 1/n^2 sum_i sum_j (p_i - y_j)^2 = mean_i (p_i^2 - 2 p_i ybar + mean(y^2));   dpred_i = 2/n (p_i - ybar)
#pragma once
#include "common.h"

namespace goalnet {

__device__ __forceinline__ void mse_bcast_block(const float* __restrict__ pred, const float* __restrict__ labels, int N,
                                                float* __restrict__ loss, float* __restrict__ dpred) {
    __shared__ double ybar_s;
    __shared__ double r4[4][4];
    double sy = 0.0, sy2 = 0.0, sp = 0.0, sp2 = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        const double y = labels[i], p = pred[i];
        sy += y; sy2 += y * y; sp += p; sp2 += p * p;
    }
    sy = wave_sum_d(sy); sy2 = wave_sum_d(sy2); sp = wave_sum_d(sp); sp2 = wave_sum_d(sp2);
    if ((threadIdx.x & 63) == 0) {
        const int wv = threadIdx.x >> 6;
        r4[wv][0] = sy; r4[wv][1] = sy2; r4[wv][2] = sp; r4[wv][3] = sp2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[4];
        for (int j = 0; j < 4; ++j) t[j] = r4[0][j] + r4[1][j] + r4[2][j] + r4[3][j];
        const double n = (double)N;
        const double ybar = t[0] / n;
        ybar_s = ybar;
        if (loss) loss[0] = (float)(t[3] / n - 2.0 * (t[2] / n) * ybar + t[1] / n);
    }
    __syncthreads();
    if (dpred) {
        const double ybar = ybar_s;
        for (int i = threadIdx.x; i < N; i += 256) dpred[i] = (float)(2.0 / (double)N * ((double)pred[i] - ybar));
    }
}

}  // namespace goalnet
