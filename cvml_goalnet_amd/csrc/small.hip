// The small ops of the hot path: AudBl's two Conv1d layers, the 128 -> 1 head with Sigmoid and 4y+1,
// the broadcast MSE loss, bias-gradient column sums, elementwise helpers and the fused Adam step.
// /root/reference/utils.py:203-227 (AudBl), 255-256, 270 (head), main.py:68-70, 191-193 (loss, Adam).
//
// Together they are < 1 % of the model's arithmetic and, except Adam, touch
// kilobytes to a few megabytes: they are written for correctness and determinism, not for a roofline.
// Adam is the exception: one pass over the flat parameter arena at 28 B/parameter (HBM-bound), it was
// 57 % of the reference's step at its own sub-batch size.
#include <hip/hip_bf16.h>

#include "common.h"
#include "mse.h"

using namespace goalnet;

namespace {

// ---------------------------------------------------------------------------------------------
// Conv1d (kernel 3), NCL layout.
// ---------------------------------------------------------------------------------------------
// 16 lanes (one DPP row) per output element: the lanes split the input channels, row16_sum adds them in a fixed order.
__global__ __launch_bounds__(256) void conv1d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, int relu, float* __restrict__ y,
                                                        int N, int Cin, int L, int Cout, int Lo, int stride, int pad) {
    const int64_t total = (int64_t)N * Cout * Lo;
    const int g = threadIdx.x & 15;
    const int64_t step = (int64_t)gridDim.x * 16;
    const int64_t rounds = (total + step - 1) / step;          // every lane runs the same number of rounds (DPP needs full rows)
    int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    for (int64_t r = 0; r < rounds; ++r, i += step) {
        const bool ok = i < total;
        const int64_t ic = ok ? i : 0;
        const int lo = (int)(ic % Lo);
        const int co = (int)((ic / Lo) % Cout);
        const int64_t n = ic / ((int64_t)Lo * Cout);
        const float* xs = x + n * Cin * L;
        const float* ws = w + (int64_t)co * Cin * 3;
        float acc = 0.f;
        const int l0 = stride * lo - pad;
        for (int ci = g; ci < Cin; ci += 16) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int l = l0 + k;
                if ((unsigned)l < (unsigned)L) acc = fmaf(xs[ci * L + l], ws[ci * 3 + k], acc);
            }
        }
        acc = row16_sum(acc) + b[co];
        if (ok && g == 0) y[i] = relu ? fmaxf(acc, 0.f) : acc;
    }
}

// dx[n][ci][l] = sum_{co,k : stride*lo - pad + k = l} dz[n][co][lo] * w[co][ci][k]; 16 lanes split the output channels
__global__ __launch_bounds__(256) void conv1d_dx_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                       float* __restrict__ dx, int N, int Cin, int L, int Cout, int Lo,
                                                       int stride, int pad) {
    const int64_t total = (int64_t)N * Cin * L;
    const int g = threadIdx.x & 15;
    const int64_t step = (int64_t)gridDim.x * 16;
    const int64_t rounds = (total + step - 1) / step;
    int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    for (int64_t r = 0; r < rounds; ++r, i += step) {
        const bool ok = i < total;
        const int64_t ic = ok ? i : 0;
        const int l = (int)(ic % L);
        const int ci = (int)((ic / L) % Cin);
        const int64_t n = ic / ((int64_t)L * Cin);
        const float* dzs = dz + n * Cout * Lo;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int t = l + pad - k;
            if (t < 0 || t % stride != 0) continue;
            const int lo = t / stride;
            if (lo >= Lo) continue;
            for (int co = g; co < Cout; co += 16) acc = fmaf(dzs[co * Lo + lo], w[((int64_t)co * Cin + ci) * 3 + k], acc);
        }
        acc = row16_sum(acc);
        if (ok && g == 0) dx[i] = acc;
    }
}

// ---- the same two gradients for many frames (N >= 64: the 1024-frame batches of bench.py) -----------------------------
// dX: one thread per output, channels fastest so that a wave reads the weights w[co][ci..ci+63][k] as one 768-B run
__global__ __launch_bounds__(256) void conv1d_dx_big_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                           float* __restrict__ dx, int N, int Cin, int L, int Cout, int Lo,
                                                           int stride, int pad) {
    const int64_t total = (int64_t)N * L * Cin;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const int l = (int)((i / Cin) % L);
        const int64_t n = i / ((int64_t)Cin * L);
        const float* dzs = dz + n * Cout * Lo;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int t = l + pad - k;
            if (t < 0 || t % stride != 0) continue;
            const int lo = t / stride;
            if (lo >= Lo) continue;
            for (int co = 0; co < Cout; ++co) acc = fmaf(dzs[co * Lo + lo], w[((int64_t)co * Cin + ci) * 3 + k], acc);
        }
        dx[(n * Cin + ci) * L + l] = acc;
    }
}

// dW: block = 4 x 4 (co, ci) pairs x 16 frame lanes; every lane walks its frames (fp32 per frame, fp64 across frames), the 16
// lane totals of a pair are added in a fixed order. The block re-uses each dz / x row 4 times from L1.
__global__ __launch_bounds__(256) void conv1d_dw_big_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                           float* __restrict__ dw, double* __restrict__ part, int N, int Cin, int L,
                                                           int Cout, int Lo, int stride, int pad) {
    __shared__ double red[16][16][3];
    const int pair = threadIdx.x >> 4, lanef = threadIdx.x & 15;
    const int cig = (Cin + 3) / 4;
    const int co = (blockIdx.x / cig) * 4 + (pair >> 2), ci = (blockIdx.x % cig) * 4 + (pair & 3);
    // blockIdx.y = slice of the frames (gridDim.y > 1: fp64 partial sums to `part`, added in slice order by
    // conv1d_dw_sum_kernel; with one slice a lane walked N / 16 frames: 0.42 ms at 1024 frames for 0.2 GFLOP)
    const int n0 = (int)((int64_t)N * blockIdx.y / gridDim.y), n1 = (int)((int64_t)N * (blockIdx.y + 1) / gridDim.y);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (co < Cout && ci < Cin) {
        for (int n = n0 + lanef; n < n1; n += 16) {
            const float* xs = x + ((int64_t)n * Cin + ci) * L;
            const float* ds = dz + ((int64_t)n * Cout + co) * Lo;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            for (int lo = 0; lo < Lo; ++lo) {
                const float d = ds[lo];
                const int l0 = stride * lo - pad;
                if ((unsigned)(l0) < (unsigned)L) s0 = fmaf(d, xs[l0], s0);
                if ((unsigned)(l0 + 1) < (unsigned)L) s1 = fmaf(d, xs[l0 + 1], s1);
                if ((unsigned)(l0 + 2) < (unsigned)L) s2 = fmaf(d, xs[l0 + 2], s2);
            }
            a0 += s0; a1 += s1; a2 += s2;
        }
    }
    red[pair][lanef][0] = a0; red[pair][lanef][1] = a1; red[pair][lanef][2] = a2;
    __syncthreads();
    if (lanef < 3 && co < Cout && ci < Cin) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[pair][j][lanef];
        const int64_t o = ((int64_t)co * Cin + ci) * 3 + lanef;
        if (gridDim.y == 1) dw[o] = (float)t;
        else part[(int64_t)blockIdx.y * Cout * Cin * 3 + o] = t;
    }
}

__global__ __launch_bounds__(256) void conv1d_dw_sum_kernel(const double* __restrict__ part, int slices, int64_t n, float* __restrict__ dw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double t = 0.0;
    for (int s = 0; s < slices; ++s) t += part[(int64_t)s * n + i];
    dw[i] = (float)t;
}

// dw[co][ci][k] = sum_{n,lo} dz[n][co][lo] * x[n][ci][stride*lo - pad + k]; one block per (co, ci), the 256
// threads split the frames, fp64 block reduction (deterministic).
__global__ __launch_bounds__(256) void conv1d_dw_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                       float* __restrict__ dw, int N, int Cin, int L, int Cout, int Lo,
                                                       int stride, int pad) {
    __shared__ double red[4][3];
    const int co = blockIdx.x / Cin, ci = blockIdx.x % Cin;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float* xs = x + ((int64_t)n * Cin + ci) * L;
        const float* ds = dz + ((int64_t)n * Cout + co) * Lo;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int lo = 0; lo < Lo; ++lo) {
            const float d = ds[lo];
            const int l0 = stride * lo - pad;
            if ((unsigned)(l0) < (unsigned)L) s0 = fmaf(d, xs[l0], s0);
            if ((unsigned)(l0 + 1) < (unsigned)L) s1 = fmaf(d, xs[l0 + 1], s1);
            if ((unsigned)(l0 + 2) < (unsigned)L) s2 = fmaf(d, xs[l0 + 2], s2);
        }
        a0 += s0; a1 += s1; a2 += s2;
    }
    a0 = wave_sum_d(a0); a1 = wave_sum_d(a1); a2 = wave_sum_d(a2);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wv][0] = a0; red[wv][1] = a1; red[wv][2] = a2; }
    __syncthreads();
    if (threadIdx.x < 3)
        dw[(int64_t)blockIdx.x * 3 + threadIdx.x] =
            (float)(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// db[co] = sum_{n,lo} dz[n][co][lo]; one block per co
__global__ __launch_bounds__(256) void conv1d_db_kernel(const float* __restrict__ dz, float* __restrict__ db, int N, int Cout, int Lo) {
    __shared__ double red[4];
    const int co = blockIdx.x;
    double a = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float* ds = dz + ((int64_t)n * Cout + co) * Lo;
        float s = 0.f;
        for (int lo = 0; lo < Lo; ++lo) s += ds[lo];
        a += s;
    }
    a = wave_sum_d(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) db[co] = (float)(red[0] + red[1] + red[2] + red[3]);
}

// ---- few frames (N < 64: the reference's 10-frame sub-batches): the whole backward of one Conv1d layer in ONE launch.
// y (nullable) is the layer's own ReLU output: the effective output gradient is dz * (y > 0), so no separate ReLU-backward pass.
// Blocks [0, nA): weight + bias gradient. A block owns 256 / Cin output channels x all Cin input channels, one (co, ci) pair
// per thread; the gated dz rows of its channels sit in LDS, x is read through L1 / L2 (36 KB in all). Per frame the taps are
// summed in fp32, across frames in fp64, frames in index order (conv1d_dw_kernel's arithmetic).
// Blocks [nA, ...): data gradient, 16 lanes per output element splitting the output channels (conv1d_dx_kernel).
__global__ __launch_bounds__(256) void conv1d_bwd_small_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                              const float* __restrict__ y, const float* __restrict__ w,
                                                              float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db,
                                                              int N, int Cin, int L, int Cout, int Lo, int stride, int pad, int nA) {
    extern __shared__ float dzs[];                        // part A: [N][cpb][Lo]
    if ((int)blockIdx.x < nA) {
        const int cpb = 256 / Cin;
        const int co0 = blockIdx.x * cpb;
        const int tot = N * cpb * Lo;
        for (int i = threadIdx.x; i < tot; i += 256) {
            const int lo = i % Lo, cl = (i / Lo) % cpb, n = i / (Lo * cpb);
            float v = 0.f;
            if (co0 + cl < Cout) {
                const int64_t o = ((int64_t)n * Cout + co0 + cl) * Lo + lo;
                v = dz[o];
                if (y && !(y[o] > 0.f)) v = 0.f;
            }
            dzs[i] = v;
        }
        __syncthreads();
        const int cl = threadIdx.x / Cin, ci = threadIdx.x - cl * Cin;
        if (cl < cpb && co0 + cl < Cout) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
            for (int n = 0; n < N; ++n) {
                const float* xs = x + ((int64_t)n * Cin + ci) * L;
                const float* ds = dzs + (n * cpb + cl) * Lo;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f;
                for (int lo = 0; lo < Lo; ++lo) {
                    const float d = ds[lo];
                    const int l0 = stride * lo - pad;
                    if ((unsigned)(l0) < (unsigned)L) s0 = fmaf(d, xs[l0], s0);
                    if ((unsigned)(l0 + 1) < (unsigned)L) s1 = fmaf(d, xs[l0 + 1], s1);
                    if ((unsigned)(l0 + 2) < (unsigned)L) s2 = fmaf(d, xs[l0 + 2], s2);
                }
                a0 += s0; a1 += s1; a2 += s2;
            }
            float* o = dw + ((int64_t)(co0 + cl) * Cin + ci) * 3;
            o[0] = (float)a0; o[1] = (float)a1; o[2] = (float)a2;
        }
        if ((int)threadIdx.x < cpb && co0 + (int)threadIdx.x < Cout) {
            double a = 0.0;
            for (int n = 0; n < N; ++n) {
                const float* ds = dzs + (n * cpb + threadIdx.x) * Lo;
                float t = 0.f;
                for (int lo = 0; lo < Lo; ++lo) t += ds[lo];
                a += t;
            }
            db[co0 + threadIdx.x] = (float)a;
        }
        return;
    }
    const int64_t total = (int64_t)N * Cin * L;
    const int g = threadIdx.x & 15;
    const int64_t step = (int64_t)(gridDim.x - nA) * 16;
    const int64_t rounds = (total + step - 1) / step;
    int64_t i = (int64_t)(blockIdx.x - nA) * 16 + (threadIdx.x >> 4);
    for (int64_t r = 0; r < rounds; ++r, i += step) {
        const bool ok = i < total;
        const int64_t ic = ok ? i : 0;
        const int l = (int)(ic % L);
        const int ci = (int)((ic / L) % Cin);
        const int64_t n = ic / ((int64_t)L * Cin);
        const float* dzr = dz + n * Cout * Lo;
        const float* yr = y ? y + n * Cout * Lo : nullptr;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int t = l + pad - k;
            if (t < 0 || t % stride != 0) continue;
            const int lo = t / stride;
            if (lo >= Lo) continue;
            for (int co = g; co < Cout; co += 16) {
                float d = dzr[co * Lo + lo];
                if (yr && !(yr[co * Lo + lo] > 0.f)) d = 0.f;
                acc = fmaf(d, w[((int64_t)co * Cin + ci) * 3 + k], acc);
            }
        }
        acc = row16_sum(acc);
        if (ok && g == 0) dx[i] = acc;
    }
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ dz, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dz[i] = y[i] > 0.f ? dy[i] : 0.f;
}

__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ m, int64_t ldm,
                                                 float* __restrict__ y, int64_t ldy, int M, int J) {
    const int64_t total = (int64_t)M * J;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / J;
        const int c = (int)(i - r * J);
        y[r * ldy + c] = x[r * ldx + c] * m[r * ldm + c];
    }
}

// out[j] = sum_m x[m][j]: block = 32 columns x 8 row lanes, fp64 accumulation
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ldx, int M, int J, float* __restrict__ out) {
    __shared__ double red[8][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + cx;
    double a = 0.0;
    if (col < J)
        for (int m = ry; m < M; m += 8) a += (double)x[(int64_t)m * ldx + col];
    red[ry][cx] = a;
    __syncthreads();
    if (ry == 0 && col < J) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += red[i][cx];
        out[col] = (float)s;
    }
}

// ---------------------------------------------------------------------------------------------
// head: z = h . w + b ; out = 4 * sigmoid(z) + 1.  One wave per frame.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ h, int64_t ldh, const float* __restrict__ w,
                                                      const float* __restrict__ b, float* __restrict__ logit,
                                                      float* __restrict__ out, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc = fmaf(h[(int64_t)n * ldh + k], w[k], acc);
    acc = wave_sum(acc);
    if (lane == 0) {
        const float z = acc + b[0];
        if (logit) logit[n] = z;
        out[n] = 4.f / (1.f + expf(-z)) + 1.f;
    }
}

__device__ __forceinline__ float dlogit_of(float dout, float out) {
    // d/dz (4*sigmoid(z) + 1) = 4 s (1 - s), s = (out - 1) / 4
    const float s = (out - 1.f) * 0.25f;
    return dout * 4.f * s * (1.f - s);
}

__global__ __launch_bounds__(256) void head_bwd_dh_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                         const float* __restrict__ w, const float* __restrict__ mult, int64_t ldmult,
                                                         float* __restrict__ dh, int64_t lddh, int N, int K) {
    const int64_t total = (int64_t)N * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / K;
        const int k = (int)(i - n * K);
        float v = dlogit_of(dout[n], out[n]) * w[k];
        if (mult) v *= mult[n * ldmult + k];
        dh[n * lddh + k] = v;
    }
}

// dw[k] = sum_n dlogit[n] h[n][k]; db = sum_n dlogit[n] (column K). Block = 64 columns x 16 row groups: a wave reads 64
// consecutive k of one row, every thread sums its rows in fp64, the 16 group totals of a column are added in a fixed order.
// (One thread per column walking all rows took 316 us at 1024 rows: a thousand dependent round trips.)
__global__ __launch_bounds__(1024) void head_bwd_dw_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                          const float* __restrict__ h, int64_t ldh, float* __restrict__ dw,
                                                          float* __restrict__ db, int N, int K) {
    __shared__ double red[16][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + c;
    double a = 0.0;
    if (k <= K)
        for (int n = rg; n < N; n += 16) {
            const float g = dlogit_of(dout[n], out[n]);
            a += (double)(k < K ? g * h[(int64_t)n * ldh + k] : g);
        }
    red[rg][c] = a;
    __syncthreads();
    if (rg != 0 || k > K) return;
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) t += red[j][c];
    if (k < K) dw[k] = (float)t;
    else db[0] = (float)t;
}

// loss = 1/n^2 sum_i sum_j (p_i - y_j)^2 = mean_i (p_i^2 - 2 p_i ybar + mean(y^2)); dpred_i = 2/n (p_i - ybar)
__global__ __launch_bounds__(256) void mse_bcast_kernel(const float* __restrict__ pred, const float* __restrict__ labels, int N,
                                                       float* __restrict__ loss, float* __restrict__ dpred) {
    mse_bcast_block(pred, labels, N, loss, dpred);
}

// ---------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults, single-tensor operation order), one pass over the flat arena.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float b2, float omb1, float omb2,
                                      float eps, float step_size, float bc2_sqrt, float gs) {
    g *= gs;
    m = m + omb1 * (g - m);                    // exp_avg.lerp_(grad, 1 - beta1)
    v = v * b2 + (omb2 * g) * g;               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);           // param.addcdiv_(exp_avg, denom, value = -step_size)
}

// One kernel for both entry points (host step count / device step counter) so that they produce the same bits.
// The scalars torch's _single_tensor_adam computes as python floats (doubles) are computed in fp64 here.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, int64_t n, double lr, double beta1, double beta2,
                                                  float eps, const int64_t* __restrict__ step_dev, int64_t step_host, float gs,
                                                  uint2* __restrict__ shadow, int64_t sh_b4, int64_t sh_e4, int sh_f16,
                                                  const int64_t* __restrict__ bad_step) {
    const int64_t ti = step_dev ? *step_dev + step_host : step_host;               // device counter + bias, or the host count
    // precision = "fp16": goalnet_grad_finite_check stamped this step as overflowed -> the whole update is skipped
    if (bad_step && *bad_step == ti) return;
    const double t = (double)ti;
    const float step_size = (float)(lr / (1.0 - pow(beta1, t)));
    const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2, t));
    const float b2 = (float)beta2, omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        adam1(pp.x, gg.x, mm.x, vv.x, b2, omb1, omb2, eps, step_size, bc2_sqrt, gs);
        adam1(pp.y, gg.y, mm.y, vv.y, b2, omb1, omb2, eps, step_size, bc2_sqrt, gs);
        adam1(pp.z, gg.z, mm.z, vv.z, b2, omb1, omb2, eps, step_size, bc2_sqrt, gs);
        adam1(pp.w, gg.w, mm.w, vv.w, b2, omb1, omb2, eps, step_size, bc2_sqrt, gs);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (shadow && i >= sh_b4 && i < sh_e4) {       // 16-bit copy of the updated parameters of one slice (the next step's GEMM operand)
            if (sh_f16) {
                const _Float16 a = (_Float16)pp.x, b = (_Float16)pp.y, c = (_Float16)pp.z, d = (_Float16)pp.w;
                shadow[i - sh_b4] = make_uint2((unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16),
                                               (unsigned)__builtin_bit_cast(unsigned short, c) | ((unsigned)__builtin_bit_cast(unsigned short, d) << 16));
            } else {
                const __hip_bfloat16 a = __float2bfloat16(pp.x), b = __float2bfloat16(pp.y), c = __float2bfloat16(pp.z), d = __float2bfloat16(pp.w);
                shadow[i - sh_b4] = make_uint2((unsigned)*reinterpret_cast<const unsigned short*>(&a) | ((unsigned)*reinterpret_cast<const unsigned short*>(&b) << 16),
                                               (unsigned)*reinterpret_cast<const unsigned short*>(&c) | ((unsigned)*reinterpret_cast<const unsigned short*>(&d) << 16));
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        adam1(p[i], g[i], m[i], v[i], b2, omb1, omb2, eps, step_size, bc2_sqrt, gs);
    }
}

// x *= s (the loss scale of precision = "fp16" on dL/dpred)
__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, int64_t n, float s) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= s;
}

// any non-finite value in g -> *bad_step = max(*bad_step, this step's 1-based count): the fused Adam of this step then returns
// without touching anything (adam_kernel); any number of blocks may write. The step's last launch
// (goalnet_counters_add4_guarded, stepstate.hip) keeps the step count where it is for a stamped step and clears the stamp.
__global__ __launch_bounds__(256) void grad_finite_check_kernel(const float* __restrict__ g, int64_t n, const int64_t* __restrict__ step,
                                                               int64_t step_bias, int64_t* __restrict__ bad_step,
                                                               int64_t* __restrict__ skipped) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = g[i];
        bad |= !(fabsf(v) <= 3.0e38f);                 // inf or nan
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) {
        const unsigned long long t = (unsigned long long)(*step + step_bias);
        const unsigned long long old = atomicMax(reinterpret_cast<unsigned long long*>(bad_step), t);
        if (old != t && skipped) atomicAdd(reinterpret_cast<unsigned long long*>(skipped), 1ull);       // first reporter of this step
    }
}

unsigned grid1d(int64_t n, int cap = 4096) {
    int64_t b = (n + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" {

int goalnet_conv1d_fwd(const float* x, const float* w, const float* b, int relu, float* y,
                       int N, int Cin, int L, int Cout, int stride, int pad, void* stream) {
    GN_REQUIRE(x && w && b && y, GOALNET_E_NULL, "conv1d_fwd: null pointer");
    GN_REQUIRE(N > 0 && Cin > 0 && L > 0 && Cout > 0 && stride > 0 && pad >= 0 && L + 2 * pad >= 3, GOALNET_E_SHAPE, "conv1d_fwd: bad dims");
    const int Lo = (L + 2 * pad - 3) / stride + 1;
    hipLaunchKernelGGL(conv1d_fwd_kernel, dim3(grid1d((int64_t)N * Cout * Lo * 16)), dim3(256), 0, (hipStream_t)stream, x, w, b, relu, y,
                       N, Cin, L, Cout, Lo, stride, pad);
    GN_LAUNCH_CHECK("conv1d_fwd");
    return 0;
}

constexpr int CONV1D_DW_SLICES = 8;      // frame slices of the many-frame weight gradient

size_t goalnet_conv1d_bwd_ws_bytes(int N, int Cin, int Cout) {
    return N >= 64 * CONV1D_DW_SLICES ? (size_t)CONV1D_DW_SLICES * Cout * Cin * 3 * sizeof(double) : 0;
}

int goalnet_conv1d_bwd(const float* x, const float* dz, const float* w, float* dx, float* dw, float* db,
                       int N, int Cin, int L, int Cout, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
    GN_REQUIRE(x && dz && w && dw && db, GOALNET_E_NULL, "conv1d_bwd: null pointer");
    GN_REQUIRE(N > 0 && Cin > 0 && L > 0 && Cout > 0 && stride > 0 && pad >= 0 && L + 2 * pad >= 3, GOALNET_E_SHAPE, "conv1d_bwd: bad dims");
    const int Lo = (L + 2 * pad - 3) / stride + 1;
    hipStream_t st = (hipStream_t)stream;
    const bool many = N >= 64;          // few frames (the reference's sub-batches): lanes split the channels; many: lanes split the frames
    if (dx) {
        if (many)
            hipLaunchKernelGGL(conv1d_dx_big_kernel, dim3(grid1d((int64_t)N * Cin * L)), dim3(256), 0, st, dz, w, dx, N, Cin, L, Cout, Lo, stride, pad);
        else
            hipLaunchKernelGGL(conv1d_dx_kernel, dim3(grid1d((int64_t)N * Cin * L * 16)), dim3(256), 0, st, dz, w, dx, N, Cin, L, Cout, Lo, stride, pad);
        GN_LAUNCH_CHECK("conv1d_bwd.dx");
    }
    if (many) {
        const size_t need = goalnet_conv1d_bwd_ws_bytes(N, Cin, Cout);
        const int slices = need && ws && ws_bytes >= need ? CONV1D_DW_SLICES : 1;
        GN_REQUIRE(slices == 1 || (reinterpret_cast<uintptr_t>(ws) & 7u) == 0, GOALNET_E_ALIGN, "conv1d_bwd: workspace must be 8-byte aligned");
        hipLaunchKernelGGL(conv1d_dw_big_kernel, dim3(((Cout + 3) / 4) * ((Cin + 3) / 4), slices), dim3(256), 0, st, x, dz, dw, (double*)ws,
                           N, Cin, L, Cout, Lo, stride, pad);
        if (slices > 1) {
            const int64_t n = (int64_t)Cout * Cin * 3;
            hipLaunchKernelGGL(conv1d_dw_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const double*)ws, slices, n, dw);
        }
    } else
        hipLaunchKernelGGL(conv1d_dw_kernel, dim3(Cout * Cin), dim3(256), 0, st, x, dz, dw, N, Cin, L, Cout, Lo, stride, pad);
    GN_LAUNCH_CHECK("conv1d_bwd.dw");
    hipLaunchKernelGGL(conv1d_db_kernel, dim3(Cout), dim3(256), 0, st, dz, db, N, Cout, Lo);
    GN_LAUNCH_CHECK("conv1d_bwd.db");
    return 0;
}

/* goalnet_conv1d_bwd for few frames (N < 64) in ONE launch, with the ReLU backward of the layer's own output folded in:
 * y (nullable) = the layer's ReLU output, effective dz = dz * (y > 0). dx nullable (first layer). Cin <= 256. */
int goalnet_conv1d_bwd_small(const float* x, const float* dz, const float* y, const float* w, float* dx, float* dw, float* db,
                             int N, int Cin, int L, int Cout, int stride, int pad, void* stream) {
    GN_REQUIRE(x && dz && w && dw && db, GOALNET_E_NULL, "conv1d_bwd_small: null pointer");
    GN_REQUIRE(N > 0 && N < 64 && Cin > 0 && Cin <= 256 && L > 0 && Cout > 0 && stride > 0 && pad >= 0 && L + 2 * pad >= 3, GOALNET_E_SHAPE,
               "conv1d_bwd_small: bad dims (N < 64, Cin <= 256)");
    const int Lo = (L + 2 * pad - 3) / stride + 1;
    const int cpb = 256 / Cin;
    const size_t lds = (size_t)N * cpb * Lo * sizeof(float);
    GN_REQUIRE(lds <= 48 * 1024, GOALNET_E_SHAPE, "conv1d_bwd_small: the dz rows of a block do not fit LDS");
    const int nA = (Cout + cpb - 1) / cpb;
    const int nB = dx ? (int)grid1d((int64_t)N * Cin * L * 16) : 0;
    hipLaunchKernelGGL(conv1d_bwd_small_kernel, dim3(nA + nB), dim3(256), lds, (hipStream_t)stream, x, dz, y, w, dx, dw, db,
                       N, Cin, L, Cout, Lo, stride, pad, nA);
    GN_LAUNCH_CHECK("conv1d_bwd_small");
    return 0;
}

int goalnet_relu_bwd(const float* dy, const float* y, float* dz, int64_t n, void* stream) {
    GN_REQUIRE(dy && y && dz, GOALNET_E_NULL, "relu_bwd: null pointer");
    GN_REQUIRE(n > 0, GOALNET_E_SHAPE, "relu_bwd: bad count");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dz, n);
    GN_LAUNCH_CHECK("relu_bwd");
    return 0;
}

int goalnet_mul(const float* x, int64_t ldx, const float* mult, int64_t ldmult, float* y, int64_t ldy,
                int M, int J, void* stream) {
    GN_REQUIRE(x && mult && y, GOALNET_E_NULL, "mul: null pointer");
    GN_REQUIRE(M > 0 && J > 0, GOALNET_E_SHAPE, "mul: bad dims");
    hipLaunchKernelGGL(mul_kernel, dim3(grid1d((int64_t)M * J)), dim3(256), 0, (hipStream_t)stream, x, ldx, mult, ldmult, y, ldy, M, J);
    GN_LAUNCH_CHECK("mul");
    return 0;
}

int goalnet_colsum(const float* x, int64_t ldx, int M, int J, float* out, void* stream) {
    GN_REQUIRE(x && out, GOALNET_E_NULL, "colsum: null pointer");
    GN_REQUIRE(M > 0 && J > 0, GOALNET_E_SHAPE, "colsum: bad dims");
    hipLaunchKernelGGL(colsum_kernel, dim3((J + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, ldx, M, J, out);
    GN_LAUNCH_CHECK("colsum");
    return 0;
}

int goalnet_head_fwd(const float* h, int64_t ldh, const float* w, const float* b, float* logit, float* out,
                     int N, int K, void* stream) {
    GN_REQUIRE(h && w && b && out, GOALNET_E_NULL, "head_fwd: null pointer");
    GN_REQUIRE(N > 0 && K > 0, GOALNET_E_SHAPE, "head_fwd: bad dims");
    hipLaunchKernelGGL(head_fwd_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, h, ldh, w, b, logit, out, N, K);
    GN_LAUNCH_CHECK("head_fwd");
    return 0;
}

int goalnet_head_bwd(const float* dout, const float* out, const float* h, int64_t ldh, const float* w,
                     const float* mult, int64_t ldmult, float* dh, int64_t lddh, float* dw, float* db,
                     int N, int K, void* stream) {
    GN_REQUIRE(dout && out && h && w && dh && dw && db, GOALNET_E_NULL, "head_bwd: null pointer");
    GN_REQUIRE(N > 0 && K > 0 && K < 1024, GOALNET_E_SHAPE, "head_bwd: bad dims (K < 1024)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(head_bwd_dh_kernel, dim3(grid1d((int64_t)N * K)), dim3(256), 0, st, dout, out, w, mult, ldmult, dh, lddh, N, K);
    GN_LAUNCH_CHECK("head_bwd.dh");
    hipLaunchKernelGGL(head_bwd_dw_kernel, dim3((K + 1 + 63) / 64), dim3(1024), 0, st, dout, out, h, ldh, dw, db, N, K);
    GN_LAUNCH_CHECK("head_bwd.dw");
    return 0;
}

int goalnet_mse_bcast(const float* pred, const float* labels, int N, float* loss, float* dpred, void* stream) {
    GN_REQUIRE(pred && labels, GOALNET_E_NULL, "mse_bcast: null pointer");
    GN_REQUIRE(N > 0, GOALNET_E_SHAPE, "mse_bcast: bad count");
    hipLaunchKernelGGL(mse_bcast_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, labels, N, loss, dpred);
    GN_LAUNCH_CHECK("mse_bcast");
    return 0;
}

static int adam_launch(const char* who, float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                       double beta2, double eps, const int64_t* step_dev, int64_t step_host, float grad_scale, void* stream,
                       void* shadow = nullptr, int64_t sh_begin = 0, int64_t sh_count = 0, int sh_f16 = 0, const int64_t* bad_step = nullptr,
                       int max_blocks = 8192) {
    GN_REQUIRE(p && g && m && v, GOALNET_E_NULL, "%s: null pointer", who);
    GN_REQUIRE(n > 0, GOALNET_E_SHAPE, "%s: bad count", who);
    GN_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), GOALNET_E_ALIGN, "%s: arenas must be 16-byte aligned", who);
    hipLaunchKernelGGL(adam_kernel, dim3(grid1d(n >> 2, max_blocks)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2,
                       (float)eps, step_dev, step_host, grad_scale, (uint2*)shadow, sh_begin >> 2, (sh_begin + sh_count) >> 2, sh_f16, bad_step);
    GN_LAUNCH_CHECK(who);
    return 0;
}

/* goalnet_adam_step_dev on at most max_blocks blocks of 256 threads: a BACKGROUND pass. The 10-frame step runs the update of
 * linear5.weight (90 % of the bytes) on its own stream under the rest of backward; on the default 8192 blocks it takes every CU and
 * all of the HBM bandwidth and the latency-bound kernels of the backward chain queue behind it, on ~128 blocks it streams at a
 * third of the bandwidth and leaves the chain alone. */
int goalnet_adam_step_dev_blocks(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                                 double eps, const int64_t* step, int64_t step_bias, float grad_scale, int max_blocks, void* stream) {
    GN_REQUIRE(step, GOALNET_E_NULL, "adam_step_dev_blocks: null pointer");
    GN_REQUIRE(max_blocks >= 1 && max_blocks <= 65535, GOALNET_E_SHAPE, "adam_step_dev_blocks: 1..65535 blocks");
    return adam_launch("adam_step_dev_blocks", p, g, m, v, n, lr, beta1, beta2, eps, step, step_bias, grad_scale, stream, nullptr, 0, 0, 0, nullptr,
                       max_blocks);
}

int goalnet_adam_step_dev_shadow(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                                 double eps, const int64_t* step, int64_t step_bias, float grad_scale, void* shadow_bf16,
                                 int64_t shadow_begin, int64_t shadow_count, int f16, void* stream) {
    GN_REQUIRE(step && shadow_bf16, GOALNET_E_NULL, "adam_step_dev_shadow: null pointer");
    GN_REQUIRE(shadow_begin >= 0 && shadow_count > 0 && shadow_begin + shadow_count <= n && (shadow_begin & 3) == 0 && (shadow_count & 3) == 0,
               GOALNET_E_SHAPE, "adam_step_dev_shadow: the shadowed slice must lie inside the arena, offset and length multiples of 4");
    GN_REQUIRE((reinterpret_cast<uintptr_t>(shadow_bf16) & 7u) == 0, GOALNET_E_ALIGN, "adam_step_dev_shadow: shadow must be 8-byte aligned");
    return adam_launch("adam_step_dev_shadow", p, g, m, v, n, lr, beta1, beta2, eps, step, step_bias, grad_scale, stream, shadow_bf16,
                       shadow_begin, shadow_count, f16);
}

/* precision = "fp16" (loss-scaled gradients): the same two entry points with an overflow guard — the update is skipped when
 * goalnet_grad_finite_check stamped this step (*bad_step == *step + step_bias). shadow may be NULL. */
int goalnet_adam_step_dev_guarded(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                                  double eps, const int64_t* step, int64_t step_bias, float grad_scale, void* shadow_16,
                                  int64_t shadow_begin, int64_t shadow_count, int f16, const int64_t* bad_step, void* stream) {
    GN_REQUIRE(step && bad_step, GOALNET_E_NULL, "adam_step_dev_guarded: null pointer");
    if (shadow_16) {
        GN_REQUIRE(shadow_begin >= 0 && shadow_count > 0 && shadow_begin + shadow_count <= n && (shadow_begin & 3) == 0 && (shadow_count & 3) == 0,
                   GOALNET_E_SHAPE, "adam_step_dev_guarded: the shadowed slice must lie inside the arena, offset and length multiples of 4");
        GN_REQUIRE((reinterpret_cast<uintptr_t>(shadow_16) & 7u) == 0, GOALNET_E_ALIGN, "adam_step_dev_guarded: shadow must be 8-byte aligned");
    }
    return adam_launch("adam_step_dev_guarded", p, g, m, v, n, lr, beta1, beta2, eps, step, step_bias, grad_scale, stream, shadow_16,
                       shadow_begin, shadow_16 ? shadow_count : 0, f16, bad_step);
}

int goalnet_scale(float* x, int64_t n, float s, void* stream) {
    GN_REQUIRE(x, GOALNET_E_NULL, "scale: null pointer");
    GN_REQUIRE(n > 0, GOALNET_E_SHAPE, "scale: bad count");
    hipLaunchKernelGGL(scale_kernel, dim3(grid1d(n, 8192)), dim3(256), 0, (hipStream_t)stream, x, n, s);
    GN_LAUNCH_CHECK("scale");
    return 0;
}

int goalnet_grad_finite_check(const float* g, int64_t n, const int64_t* step, int64_t step_bias, int64_t* bad_step, int64_t* skipped,
                              void* stream) {
    GN_REQUIRE(g && step && bad_step, GOALNET_E_NULL, "grad_finite_check: null pointer");
    GN_REQUIRE(n > 0, GOALNET_E_SHAPE, "grad_finite_check: bad count");
    hipLaunchKernelGGL(grad_finite_check_kernel, dim3(grid1d(n, 1024)), dim3(256), 0, (hipStream_t)stream, g, n, step, step_bias, bad_step, skipped);
    GN_LAUNCH_CHECK("grad_finite_check");
    return 0;
}

int goalnet_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                      double beta2, double eps, int step, float grad_scale, void* stream) {
    GN_REQUIRE(step >= 1, GOALNET_E_SHAPE, "adam_step: step is 1-based");
    return adam_launch("adam_step", p, g, m, v, n, lr, beta1, beta2, eps, nullptr, step, grad_scale, stream);
}

int goalnet_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                          double eps, const int64_t* step, int64_t step_bias, float grad_scale, void* stream) {
    GN_REQUIRE(step, GOALNET_E_NULL, "adam_step_dev: null step counter");
    return adam_launch("adam_step_dev", p, g, m, v, n, lr, beta1, beta2, eps, step, step_bias, grad_scale, stream);
}

}  // extern "C"
