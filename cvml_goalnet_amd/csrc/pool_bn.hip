// MaxPool2d(3, stride 1) + train-mode BatchNorm2d statistics, and their backward pass.
// /root/reference/utils.py:153-154, 158-159, 163-164 (forward), autograd of main.py:192 (backward).
//
// All of these are HBM-bound passes over NHWC tensors (float4 = 4 channels per lane, lanes of a wave walk
// the channel dimension, so every access is a full coalesced line). The BatchNorm *apply* never gets its
// own pass: it is folded into the operand load of the next GEMM (gemm_f32.hip), because training-mode
// statistics are only known after the whole pooled tensor has been produced (SURVEY.md §7 "Hard parts").
//
// Statistics are accumulated in fp64 and reduced through a fixed number (GOALNET_STAT_PARTS) of per-block
// partial rows that a second kernel sums in a fixed order: deterministic, and accurate to the level of
// ATen's CPU implementation (which accumulates float statistics in double).
#include <hip/hip_bf16.h>
#include <stdlib.h>

#include "common.h"

using namespace goalnet;

namespace {

constexpr int PARTS = GOALNET_STAT_PARTS;

struct d4 { double x, y, z, w; };

// block-level reduction of per-thread (4 channels x NV values) doubles over the threads that share a
// channel group g = tid % G; result written by the threads with tid < G.
template <int NV>
__device__ __forceinline__ void block_reduce_store(double (&v)[NV][4], int G, int tid, double* smem,
                                                   double* out_row, int64_t out_stride, int C) {
    // smem: [256][NV*4] doubles
#pragma unroll
    for (int a = 0; a < NV; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) smem[tid * (NV * 4) + a * 4 + c] = v[a][c];
    __syncthreads();
    if (tid < G) {
        const int reps = 256 / G;
#pragma unroll
        for (int a = 0; a < NV; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double s = 0.0;
                for (int rp = 0; rp < reps; ++rp) s += smem[(rp * G + tid) * (NV * 4) + a * 4 + c];
                out_row[a * out_stride + tid * 4 + c] = s;
            }
    }
    (void)C;
}

// Layout of the argmax bytes: [N][C / S][Hp][Wp][S] with S = 32 channels per slice (S = C when C is not a multiple of 32):
// the v2 kernels work on 32-channel slices, and with the slice as the slow index the S bytes of consecutive pixels are
// contiguous, so a wave stores whole cache lines (1 KB per instruction). In pixel-major order ([N][Hp][Wp][C]) the same
// store is 32-byte pieces C bytes apart: measured 8.65 ms vs 4.55 ms without the store for the 72 x 72 x 512 block.
__host__ __device__ __forceinline__ int idx_slice(int C) { return (C & 31) == 0 ? 32 : C; }
__device__ __forceinline__ int64_t idx_off(int64_t n, int64_t pix_in_frame, int c, int64_t frame_pix, int C) {
    const int S = idx_slice(C);
    return ((n * (C / S) + c / S) * frame_pix + pix_in_frame) * S + (c % S);
}

// ---- forward: p = maxpool3x3s1(y), idx = argmax (first maximum in kh,kw scan order, NaN wins, as ATen's
// ---- column sums over the PARTS partial rows. One block = 16 columns x 16 part lanes: lane group pg sums rows pg, pg+16, ...
// (4 independent accumulators), the 16 lane-group totals are then added in a fixed order -> deterministic, and ~64
// dependent loads per thread instead of 1024 (these kernels were 240 us at N = 10 when one thread walked all rows).
__device__ __forceinline__ double column_sum16(const double* __restrict__ col0, int nparts, int64_t stride, bool valid,
                                               double* sm /* [16][16] */) {
    const int cl = threadIdx.x & 15, pg = threadIdx.x >> 4;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (valid) {
        int pt = pg;
        for (; pt + 48 < nparts; pt += 64) {
            a0 += col0[(int64_t)pt * stride];
            a1 += col0[(int64_t)(pt + 16) * stride];
            a2 += col0[(int64_t)(pt + 32) * stride];
            a3 += col0[(int64_t)(pt + 48) * stride];
        }
        for (; pt < nparts; pt += 16) a0 += col0[(int64_t)pt * stride];
    }
    __syncthreads();                       // sm may still be read from a previous call
    sm[pg * 16 + cl] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    double t = 0.0;
    if (pg == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sm[k * 16 + cl];
    }
    return t;
}

// max_pool2d CPU kernel), partial sums of p and p*p per channel.
__global__ __launch_bounds__(256) void pool_bnstats_fwd_kernel(const float* __restrict__ y, float* __restrict__ p,
                                                              uint8_t* __restrict__ idx, double* __restrict__ partials,
                                                              int N, int Hc, int Wc, int C) {
    __shared__ double smem[256 * 8];
    const int tid = threadIdx.x;
    const int G = C >> 2;                 // channel groups of 4; G divides 256
    const int g = tid % G, pl = tid / G;
    const int ppi = 256 / G;              // pixels per block iteration
    const int Hp = Hc - 2, Wp = Wc - 2;
    const int64_t npix = (int64_t)N * Hp * Wp;
    double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int64_t pix = (int64_t)blockIdx.x * ppi + pl; pix < npix; pix += (int64_t)gridDim.x * ppi) {
        const int pw = (int)(pix % Wp);
        const int ph = (int)((pix / Wp) % Hp);
        const int64_t n = pix / ((int64_t)Wp * Hp);
        const float* src = y + ((n * Hc + ph) * Wc + pw) * C + g * 4;
        float4 v[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                v[kh * 3 + kw] = *reinterpret_cast<const float4*>(src + ((int64_t)kh * Wc + kw) * C);
        float4 best = v[0];
        unsigned bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 1; k < 9; ++k) {
            if (v[k].x > best.x || v[k].x != v[k].x) { best.x = v[k].x; bi[0] = k; }
            if (v[k].y > best.y || v[k].y != v[k].y) { best.y = v[k].y; bi[1] = k; }
            if (v[k].z > best.z || v[k].z != v[k].z) { best.z = v[k].z; bi[2] = k; }
            if (v[k].w > best.w || v[k].w != v[k].w) { best.w = v[k].w; bi[3] = k; }
        }
        *reinterpret_cast<float4*>(p + pix * C + g * 4) = best;
        if (idx) *reinterpret_cast<uint32_t*>(idx + idx_off(pix / ((int64_t)Hp * Wp), pix % ((int64_t)Hp * Wp), g * 4, (int64_t)Hp * Wp, C)) =
                bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
        acc[0][0] += (double)best.x; acc[1][0] += (double)best.x * (double)best.x;
        acc[0][1] += (double)best.y; acc[1][1] += (double)best.y * (double)best.y;
        acc[0][2] += (double)best.z; acc[1][2] += (double)best.z * (double)best.z;
        acc[0][3] += (double)best.w; acc[1][3] += (double)best.w * (double)best.w;
    }
    block_reduce_store<2>(acc, G, tid, smem, partials + (int64_t)blockIdx.x * 2 * C, C, C);
}

__device__ __forceinline__ void bn_finalize_one(int c, double s, double q, const float* gamma, const float* beta, float* rmean, float* rvar,
                                                float momentum, float eps, double count, float* mean, float* invstd, float* scale, float* shift) {
    const double m = s / count;
    double var = q / count - m * m;
    if (var < 0.0) var = 0.0;
    const float fm = (float)m;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = fm;
    invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - fm * sc;
    if (rmean) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * fm;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ partials, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ rmean,
                                                         float* __restrict__ rvar, float momentum, float eps, double count,
                                                         int nparts, int C, float* __restrict__ mean, float* __restrict__ invstd,
                                                         float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double sm[256];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const double s = column_sum16(partials + c, nparts, 2 * C, c < C, sm);
    const double q = column_sum16(partials + C + c, nparts, 2 * C, c < C, sm);
    if (c >= C || threadIdx.x >= 16) return;
    bn_finalize_one(c, s, q, gamma, beta, rmean, rvar, momentum, eps, count, mean, invstd, scale, shift);
}

// dz (the gradient wrt the BatchNorm output) arrives as fp32 or, from the bf16 GEMMs' bf16-output epilogue, as bf16
__device__ __forceinline__ float4 load_dz4(const float* q) { return *reinterpret_cast<const float4*>(q); }
__device__ __forceinline__ float4 load_dz4(const __hip_bfloat16* q) {
    const uint2 u = *reinterpret_cast<const uint2*>(q);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}

// IEEE fp16 tensors (precision = "fp16"): same 8 bytes per 4 values as bf16, another conversion
struct h16raw { uint2 u; };
__device__ __forceinline__ float h16_lo(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)); }
__device__ __forceinline__ float h16_hi(unsigned u) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16)); }
__device__ __forceinline__ float4 f4_of_h16(const uint2& u) { return make_float4(h16_lo(u.x), h16_hi(u.x), h16_lo(u.y), h16_hi(u.y)); }
__device__ __forceinline__ float4 load_dz4(const _Float16* q) { return f4_of_h16(*reinterpret_cast<const uint2*>(q)); }

// the same in two steps for software-pipelined loops: the raw load is issued early, the conversion waits for it later
__device__ __forceinline__ float4 load_dz4_raw(const float* q) { return *reinterpret_cast<const float4*>(q); }
__device__ __forceinline__ uint2 load_dz4_raw(const __hip_bfloat16* q) { return *reinterpret_cast<const uint2*>(q); }
__device__ __forceinline__ h16raw load_dz4_raw(const _Float16* q) { return h16raw{*reinterpret_cast<const uint2*>(q)}; }
__device__ __forceinline__ float4 dz4_of(const float4& v) { return v; }
__device__ __forceinline__ float4 dz4_of(const uint2& u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ float4 dz4_of(const h16raw& r) { return f4_of_h16(r.u); }

// ---- backward phase 1: partial sums of dz and dz * xhat
template <typename DZ, typename PT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const DZ* __restrict__ dz, const PT* __restrict__ p,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           double* __restrict__ partials, int64_t npix, int C) {
    __shared__ double smem[256 * 8];
    const int tid = threadIdx.x;
    const int G = C >> 2;
    const int g = tid % G, pl = tid / G;
    const int ppi = 256 / G;
    const float4 mu = *reinterpret_cast<const float4*>(mean + g * 4);
    const float4 is = *reinterpret_cast<const float4*>(invstd + g * 4);
    double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int64_t pix = (int64_t)blockIdx.x * ppi + pl; pix < npix; pix += (int64_t)gridDim.x * ppi) {
        const float4 d = load_dz4(dz + pix * C + g * 4);
        const float4 x = load_dz4(p + pix * C + g * 4);
        acc[0][0] += (double)d.x; acc[1][0] += (double)d.x * (double)((x.x - mu.x) * is.x);
        acc[0][1] += (double)d.y; acc[1][1] += (double)d.y * (double)((x.y - mu.y) * is.y);
        acc[0][2] += (double)d.z; acc[1][2] += (double)d.z * (double)((x.z - mu.z) * is.z);
        acc[0][3] += (double)d.w; acc[1][3] += (double)d.w * (double)((x.w - mu.w) * is.w);
    }
    block_reduce_store<2>(acc, G, tid, smem, partials + (int64_t)blockIdx.x * 2 * C, C, C);
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ partials, const float* __restrict__ gamma,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             double count, int nparts, int C, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ coef3) {
    __shared__ double sm[256];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const double s = column_sum16(partials + c, nparts, 2 * C, c < C, sm);
    const double q = column_sum16(partials + C + c, nparts, 2 * C, c < C, sm);
    if (c >= C || threadIdx.x >= 16) return;
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
    // dp = a*(dz - m1 - xhat*m2) = a*dz + b*p + cc,  xhat = (p - mean)*invstd
    const double a = (double)gamma[c] * (double)invstd[c];
    const double m1 = s / count, m2 = q / count;
    const double b = -a * m2 * (double)invstd[c];
    const double cc = -a * m1 - b * (double)mean[c];
    coef3[c] = (float)a;
    coef3[C + c] = (float)b;
    coef3[2 * C + c] = (float)cc;
}

// ---- backward phase 3: dy[n,h,w,c] = relu'(y) * sum over the (<= 9) pooling windows that contain (h,w) and
// whose argmax is (h,w) of dp[window], dp = a*dz + b*p + cc. A gather (no atomics, deterministic).
__global__ __launch_bounds__(256) void bnpool_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ p,
                                                        const uint8_t* __restrict__ idx,
                                                        const float* __restrict__ coef3, float* __restrict__ dy,
                                                        double* __restrict__ dbias_partials, int N, int Hc, int Wc, int C) {
    __shared__ double smem[256 * 4];
    const int tid = threadIdx.x;
    const int G = C >> 2;
    const int g = tid % G, pl = tid / G;
    const int ppi = 256 / G;
    const int Hp = Hc - 2, Wp = Wc - 2;
    const int64_t npix = (int64_t)N * Hc * Wc;
    const float4 ca = *reinterpret_cast<const float4*>(coef3 + g * 4);
    const float4 cb = *reinterpret_cast<const float4*>(coef3 + C + g * 4);
    const float4 cc = *reinterpret_cast<const float4*>(coef3 + 2 * C + g * 4);
    double accb[1][4] = {{0, 0, 0, 0}};
    for (int64_t pix = (int64_t)blockIdx.x * ppi + pl; pix < npix; pix += (int64_t)gridDim.x * ppi) {
        const int w = (int)(pix % Wc);
        const int h = (int)((pix / Wc) % Hc);
        const int64_t n = pix / ((int64_t)Wc * Hc);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int ph = h - dh, pw = w - dw;                 // window origin; this pixel is tap (dh, dw)
                const bool ok = (unsigned)ph < (unsigned)Hp && (unsigned)pw < (unsigned)Wp;
                const int phc = ok ? ph : 0, pwc = ok ? pw : 0;     // clamped: loads stay unconditional
                const int64_t q = ((n * Hp + phc) * Wp + pwc) * C + g * 4;
                const uint32_t ii = *reinterpret_cast<const uint32_t*>(idx + idx_off(n, (int64_t)phc * Wp + pwc, g * 4, (int64_t)Hp * Wp, C));
                const float4 d = *reinterpret_cast<const float4*>(dz + q);
                const float4 x = *reinterpret_cast<const float4*>(p + q);
                const unsigned k = dh * 3 + dw;
                // ReLU mask from p: a window whose argmax is this pixel has p == y here, so (y > 0) == (p > 0) for every term
                acc.x += (ok && (ii & 0xffu) == k && x.x > 0.f) ? fmaf(ca.x, d.x, fmaf(cb.x, x.x, cc.x)) : 0.f;
                acc.y += (ok && ((ii >> 8) & 0xffu) == k && x.y > 0.f) ? fmaf(ca.y, d.y, fmaf(cb.y, x.y, cc.y)) : 0.f;
                acc.z += (ok && ((ii >> 16) & 0xffu) == k && x.z > 0.f) ? fmaf(ca.z, d.z, fmaf(cb.z, x.z, cc.z)) : 0.f;
                acc.w += (ok && (ii >> 24) == k && x.w > 0.f) ? fmaf(ca.w, d.w, fmaf(cb.w, x.w, cc.w)) : 0.f;
            }
        *reinterpret_cast<float4*>(dy + pix * C + g * 4) = acc;
        accb[0][0] += (double)acc.x; accb[0][1] += (double)acc.y; accb[0][2] += (double)acc.z; accb[0][3] += (double)acc.w;
    }
    block_reduce_store<1>(accb, G, tid, smem, dbias_partials + (int64_t)blockIdx.x * C, C, C);
}

// ------------------------------------------------------------------------------------------------
// small shapes (the reference's 10-frame sub-batches at 40 x 40): every dependent round of memory accesses costs about a
// microsecond there, so these kernels are built to need few of them. <= 32 blocks of 1024 threads walk the pixels in the
// coalesced mapping of the first-generation kernels (lanes over channels, 16 bytes each; two to three pixels per thread), each
// block writes ONE partial row, and the LAST block to arrive (a ticket counter, release / acquire fences) fetches the <= 32
// rows of every column in a single batch of loads, adds them in row order and does the finalise step itself — what
// goalnet_bn_finalize / goalnet_bn_bwd_finalize / goalnet_partials_sum do in a second launch.
// (Round 3 measured the alternatives: a last block walking 80-256 partial rows serially, or one block per four channels
// with 16-byte accesses 2 KB apart, were 3-5 x slower than the two-launch form)
// ------------------------------------------------------------------------------------------------
constexpr int SMALL_T = 1024;        // threads per block
constexpr int SMALL_MAXB = 32;       // blocks = partial rows

// threads with the same tid % G own the same four channels: their NV x 4 doubles are added in thread order and the threads
// tid < G write row[a * stride + tid * 4 + c]. sm: SMALL_T * 4 doubles (32 KB), one pass per value.
template <int NV>
__device__ __forceinline__ void small_block_row(double (&v)[NV][4], int G, double* sm, double* row, int stride) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int a = 0; a < NV; ++a) {
        if (a) __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) sm[tid * 4 + c] = v[a][c];
        __syncthreads();
        if (tid < G) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double t = 0.0;
                for (int q = tid; q < SMALL_T; q += G) t += sm[q * 4 + c];
                st_dev(&row[a * stride + tid * 4 + c], t);        // read by the last block of this launch
            }
        }
    }
}

// Ticket: every block calls it after its partial row is written (st_dev: common.h "inter-block hand-over"). *ctr must be zero
// on entry and is zero again on exit. Returns true in the last block to arrive.
__device__ __forceinline__ bool last_block_ticket(int* ctr, int nblocks) {
    __shared__ int s_last;
    dev_stores_done_block();
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == nblocks - 1;
        if (s_last) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const bool last = s_last != 0;
    if (last) dev_loads_after();
    return last;
}

// last block: tot[col] = sum over the nb (<= SMALL_MAXB) partial rows of column col, rows in index order; all loads of a
// column are issued before the first add. W columns, W doubles of LDS.
__device__ __forceinline__ void small_column_totals(const double* __restrict__ partials, int nb, int W, double* tot) {
    for (int col = threadIdx.x; col < W; col += SMALL_T) {
        double v[SMALL_MAXB];
#pragma unroll
        for (int r = 0; r < SMALL_MAXB; ++r) v[r] = r < nb ? ld_dev(&partials[(int64_t)r * W + col]) : 0.0;
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < SMALL_MAXB; ++r) t += v[r];
        tot[col] = t;
    }
    __syncthreads();
}

__global__ __launch_bounds__(SMALL_T) void pool_bn_fwd_small_kernel(const float* __restrict__ y, float* __restrict__ p, uint8_t* __restrict__ idx,
                                                                   double* __restrict__ partials, int* __restrict__ ctr,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
                                                                   float* __restrict__ mean, float* __restrict__ invstd,
                                                                   float* __restrict__ scale, float* __restrict__ shift,
                                                                   int N, int Hc, int Wc, int C) {
    extern __shared__ double dsm[];                  // SMALL_T * 4 doubles (block reduction), then 2 C doubles (totals)
    const int tid = threadIdx.x;
    const int G = C >> 2, g = tid % G, pl = tid / G, ppi = SMALL_T / G;
    const int Hp = Hc - 2, Wp = Wc - 2;
    const int fp = Hp * Wp, npix = N * fp;
    double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int pix = blockIdx.x * ppi + pl; pix < npix; pix += gridDim.x * ppi) {
        const int n = pix / fp, q = pix - n * fp;
        const int ph = q / Wp, pw = q - ph * Wp;
        const float* src = y + ((int64_t)(n * Hc + ph) * Wc + pw) * C + g * 4;
        float4 v[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) v[kh * 3 + kw] = *reinterpret_cast<const float4*>(src + ((int64_t)kh * Wc + kw) * C);
        float4 best = v[0];
        unsigned bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 1; k < 9; ++k) {                    // first maximum in (kh, kw) scan order, NaN wins: ATen's max_pool2d rule
            if (v[k].x > best.x || v[k].x != v[k].x) { best.x = v[k].x; bi[0] = k; }
            if (v[k].y > best.y || v[k].y != v[k].y) { best.y = v[k].y; bi[1] = k; }
            if (v[k].z > best.z || v[k].z != v[k].z) { best.z = v[k].z; bi[2] = k; }
            if (v[k].w > best.w || v[k].w != v[k].w) { best.w = v[k].w; bi[3] = k; }
        }
        *reinterpret_cast<float4*>(p + (int64_t)pix * C + g * 4) = best;
        if (idx) *reinterpret_cast<uint32_t*>(idx + idx_off(n, q, g * 4, fp, C)) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
        acc[0][0] += (double)best.x; acc[1][0] += (double)best.x * (double)best.x;
        acc[0][1] += (double)best.y; acc[1][1] += (double)best.y * (double)best.y;
        acc[0][2] += (double)best.z; acc[1][2] += (double)best.z * (double)best.z;
        acc[0][3] += (double)best.w; acc[1][3] += (double)best.w * (double)best.w;
    }
    small_block_row<2>(acc, G, dsm, partials + (int64_t)blockIdx.x * 2 * C, C);
    if (!last_block_ticket(ctr, gridDim.x)) return;
    small_column_totals(partials, gridDim.x, 2 * C, dsm);
    for (int c = tid; c < C; c += SMALL_T)
        bn_finalize_one(c, dsm[c], dsm[C + c], gamma, beta, rmean, rvar, momentum, eps, (double)npix, mean, invstd, scale, shift);
}

// backward, first launch: (sum dz, sum dz xhat) per channel -> dgamma, dbeta and the coefficients of dp = a dz + b p + cc
__global__ __launch_bounds__(SMALL_T) void bn_bwd_reduce_small_kernel(const float* __restrict__ dz, const float* __restrict__ p,
                                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                     double* __restrict__ partials, int* __restrict__ ctr,
                                                                     const float* __restrict__ gamma, float* __restrict__ dgamma,
                                                                     float* __restrict__ dbeta, float* __restrict__ coef3, int npix, int C) {
    extern __shared__ double dsm[];
    const int tid = threadIdx.x;
    const int G = C >> 2, g = tid % G, pl = tid / G, ppi = SMALL_T / G;
    const float4 mu = *reinterpret_cast<const float4*>(mean + g * 4);
    const float4 is = *reinterpret_cast<const float4*>(invstd + g * 4);
    double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int pix = blockIdx.x * ppi + pl; pix < npix; pix += gridDim.x * ppi) {
        const float4 d = *reinterpret_cast<const float4*>(dz + (int64_t)pix * C + g * 4);
        const float4 x = *reinterpret_cast<const float4*>(p + (int64_t)pix * C + g * 4);
        acc[0][0] += (double)d.x; acc[1][0] += (double)d.x * (double)((x.x - mu.x) * is.x);
        acc[0][1] += (double)d.y; acc[1][1] += (double)d.y * (double)((x.y - mu.y) * is.y);
        acc[0][2] += (double)d.z; acc[1][2] += (double)d.z * (double)((x.z - mu.z) * is.z);
        acc[0][3] += (double)d.w; acc[1][3] += (double)d.w * (double)((x.w - mu.w) * is.w);
    }
    small_block_row<2>(acc, G, dsm, partials + (int64_t)blockIdx.x * 2 * C, C);
    if (!last_block_ticket(ctr, gridDim.x)) return;
    small_column_totals(partials, gridDim.x, 2 * C, dsm);
    for (int c = tid; c < C; c += SMALL_T) {
        const double s = dsm[c], q = dsm[C + c], count = (double)npix;
        dbeta[c] = (float)s;
        dgamma[c] = (float)q;
        const double a = (double)gamma[c] * (double)invstd[c];
        const double m1 = s / count, m2 = q / count;
        const double b = -a * m2 * (double)invstd[c];
        coef3[c] = (float)a;
        coef3[C + c] = (float)b;
        coef3[2 * C + c] = (float)(-a * m1 - b * (double)mean[c]);
    }
}

// backward, second launch: dy = relu'(y) x (max-pool backward of dp) as a gather over the <= 9 windows of every conv pixel
// (bnpool_bwd_kernel's arithmetic), and the conv bias gradient = sum of dy
__global__ __launch_bounds__(SMALL_T) void bnpool_bwd_small_kernel(const float* __restrict__ dz, const float* __restrict__ p,
                                                                  const uint8_t* __restrict__ idx, const float* __restrict__ coef3,
                                                                  float* __restrict__ dy, double* __restrict__ partials, int* __restrict__ ctr,
                                                                  float* __restrict__ dbias, int N, int Hc, int Wc, int C) {
    extern __shared__ double dsm[];
    const int tid = threadIdx.x;
    const int G = C >> 2, g = tid % G, pl = tid / G, ppi = SMALL_T / G;
    const int Hp = Hc - 2, Wp = Wc - 2, fp = Hp * Wp;
    const int fc = Hc * Wc, nconv = N * fc;
    const float4 ca = *reinterpret_cast<const float4*>(coef3 + g * 4);
    const float4 cb = *reinterpret_cast<const float4*>(coef3 + C + g * 4);
    const float4 cc = *reinterpret_cast<const float4*>(coef3 + 2 * C + g * 4);
    double accb[1][4] = {{0, 0, 0, 0}};
    for (int pix = blockIdx.x * ppi + pl; pix < nconv; pix += gridDim.x * ppi) {
        const int n = pix / fc, q = pix - n * fc;
        const int h = q / Wc, w = q - h * Wc;
        float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int ph = h - dh, pw = w - dw;                 // window origin; this pixel is tap (dh, dw)
                const bool ok = (unsigned)ph < (unsigned)Hp && (unsigned)pw < (unsigned)Wp;
                const int phc = ok ? ph : 0, pwc = ok ? pw : 0;     // clamped: loads stay unconditional
                const int64_t o = ((int64_t)(n * Hp + phc) * Wp + pwc) * C + g * 4;
                const uint32_t ii = *reinterpret_cast<const uint32_t*>(idx + idx_off(n, phc * Wp + pwc, g * 4, fp, C));
                const float4 d = *reinterpret_cast<const float4*>(dz + o);
                const float4 x = *reinterpret_cast<const float4*>(p + o);
                const unsigned k = dh * 3 + dw;
                a4.x += (ok && (ii & 0xffu) == k && x.x > 0.f) ? fmaf(ca.x, d.x, fmaf(cb.x, x.x, cc.x)) : 0.f;
                a4.y += (ok && ((ii >> 8) & 0xffu) == k && x.y > 0.f) ? fmaf(ca.y, d.y, fmaf(cb.y, x.y, cc.y)) : 0.f;
                a4.z += (ok && ((ii >> 16) & 0xffu) == k && x.z > 0.f) ? fmaf(ca.z, d.z, fmaf(cb.z, x.z, cc.z)) : 0.f;
                a4.w += (ok && (ii >> 24) == k && x.w > 0.f) ? fmaf(ca.w, d.w, fmaf(cb.w, x.w, cc.w)) : 0.f;
            }
        *reinterpret_cast<float4*>(dy + (int64_t)pix * C + g * 4) = a4;
        accb[0][0] += (double)a4.x; accb[0][1] += (double)a4.y; accb[0][2] += (double)a4.z; accb[0][3] += (double)a4.w;
    }
    small_block_row<1>(accb, G, dsm, partials + (int64_t)blockIdx.x * C, C);
    if (!last_block_ticket(ctr, gridDim.x)) return;
    small_column_totals(partials, gridDim.x, C, dsm);
    for (int c = tid; c < C; c += SMALL_T) dbias[c] = (float)dsm[c];
}

// (Round 3 also built, measured and dropped three other decompositions of these passes: one block per channel slice over ALL pixels
// with no cross-block step — 68 / 59 us for the forward / reduce against 24 / 20 here, a handful of CUs cannot keep enough loads in
// flight; a max-pool / ReLU backward per (slice, frame) out of LDS — 9 us for block 3 against 13 for the rolling-row kernel, slower for
// blocks 1 and 2, no gain on the step; and the whole backward of a block as one launch around a grid barrier — 75 us against 23 + 9,
// 160 workgroups of 1024 threads polling one arrival counter. DESIGN.md §4.3.)

// ------------------------------------------------------------------------------------------------
// v2 kernels: one block = (frame slot, 32-channel slice). The three image rows a 3x3 / stride-1 window needs are
// kept in a rolling LDS buffer, so every HBM byte is read exactly once (the v1 kernels above re-read each element
// up to 9 times through L2 and lost it across XCDs: 2.2x the algorithmic traffic in rocprof). A 32-channel slice
// of one pixel is one full 128-B line: 8 lanes x float4.
// ------------------------------------------------------------------------------------------------
constexpr int CS = 32;   // channels per block slice

__device__ __forceinline__ void slice_reduce_store(double (&v)[2][4], int nv, int tid, double* red, double* out0, double* out1) {
    // threads with the same (tid & 7) own the same 4 channels; 32 such threads per block
    __syncthreads();
    for (int a = 0; a < nv; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) red[(a * 256 + tid) * 4 + c] = v[a][c];
    __syncthreads();
    if (tid < 8) {
        for (int a = 0; a < nv; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                double s = 0.0;
                for (int t = tid; t < 256; t += 8) s += red[(a * 256 + t) * 4 + c];
                (a == 0 ? out0 : out1)[tid * 4 + c] = s;
            }
    }
}

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    const __hip_bfloat16 x = __float2bfloat16(a), y = __float2bfloat16(b);
    return (unsigned)(*reinterpret_cast<const unsigned short*>(&x)) | ((unsigned)(*reinterpret_cast<const unsigned short*>(&y)) << 16);
}

// the pooled activation p is stored as fp32 or (precision = "bf16", blocks 2 and 3) as bf16: it is read three more times per
// step (BatchNorm apply, BatchNorm backward reduce, fused backward) and every GEMM behind it consumes bf16 anyway. The
// statistics are taken from the values AS STORED, so forward and backward see one and the same activation.
__device__ __forceinline__ float4 store_p4(float* q, const float4& v) { *reinterpret_cast<float4*>(q) = v; return v; }
__device__ __forceinline__ float4 store_p4(__hip_bfloat16* q, const float4& v) {
    const uint2 u = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    *reinterpret_cast<uint2*>(q) = u;
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
template <bool F16>
__device__ __forceinline__ unsigned pack_h16x2(float a, float b) {
    if (F16) {
        const _Float16 x = (_Float16)a, y = (_Float16)b;
        return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
    }
    return pack_bf16x2(a, b);
}
__device__ __forceinline__ float4 store_p4(_Float16* q, const float4& v) {
    const uint2 u = make_uint2(pack_h16x2<true>(v.x, v.y), pack_h16x2<true>(v.z, v.w));
    *reinterpret_cast<uint2*>(q) = u;
    return f4_of_h16(u);
}
__device__ __forceinline__ float4 load_p4(const _Float16* q) { return f4_of_h16(*reinterpret_cast<const uint2*>(q)); }
__device__ __forceinline__ float4 load_p4(const float* q) { return *reinterpret_cast<const float4*>(q); }
__device__ __forceinline__ float4 load_p4(const __hip_bfloat16* q) {
    const uint2 u = *reinterpret_cast<const uint2*>(q);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}

// NPW = passes of 32 pixels per conv row (launcher: ceil(Wc / 32) <= 6)
template <typename YT, typename PT, int NPW>
__global__ __launch_bounds__(256) void pool_bnstats_fwd_v2_kernel(const YT* __restrict__ y, PT* __restrict__ p,
                                                                 uint8_t* __restrict__ idx, double* __restrict__ partials,
                                                                 int N, int Hc, int Wc, int C, int bands) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [3][Wc][CS] floats (>= 16 KB for the reduction)
    const int tid = threadIdx.x, l8 = tid & 7, px = tid >> 3;
    const int ccn = C / CS;
    const int slot = blockIdx.x / ccn, c0 = (blockIdx.x % ccn) * CS + l8 * 4;
    const int Hp = Hc - 2, Wp = Wc - 2;
    double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    // work unit = (frame, band of pooled rows): with few frames (the reference's 10-frame sub-batches) the bands give
    // the grid its parallelism; a unit keeps three conv rows in LDS and rolls down its band
    for (int unit = slot; unit < N * bands; unit += (int)gridDim.x / ccn) {
        const int n = unit / bands, band = unit % bands;
        const int p0 = (int)((int64_t)Hp * band / bands), p1 = (int)((int64_t)Hp * (band + 1) / bands);
        const YT* yn = y + (int64_t)n * Hc * Wc * C + c0;
        PT* pn = p + (int64_t)n * Hp * Wp * C + c0;
        uint8_t* in = idx ? idx + ((int64_t)n * ccn + blockIdx.x % ccn) * Hp * Wp * CS + l8 * 4 : nullptr;     // slice-major, see idx_off
        // Software pipeline over conv rows, as in the backward kernel below: the loads of row r+1 are issued (unconditionally,
        // from clamped addresses, raw) before row r's windows are computed and are written to the LDS ring one iteration later.
        // Loading straight into LDS paid one global round trip per row between two barriers (2.7 TB/s with a bf16 y).
        decltype(load_dz4_raw(yn)) ry[NPW];
        auto fetch = [&](int rr) {
            const int rc = rr < Hc ? rr : Hc - 1;
#pragma unroll
            for (int ps = 0; ps < NPW; ++ps) {
                const int x = px + 32 * ps, xc = x < Wc ? x : Wc - 1;
                ry[ps] = load_dz4_raw(yn + ((int64_t)rc * Wc + xc) * C);
            }
        };
        auto stash = [&](int rr) {
#pragma unroll
            for (int ps = 0; ps < NPW; ++ps) {
                const int x = px + 32 * ps;
                if (x < Wc) *reinterpret_cast<float4*>(&smem[(((rr % 3) * Wc) + x) * CS + l8 * 4]) = dz4_of(ry[ps]);
            }
        };
        __syncthreads();
        fetch(p0); stash(p0);
        fetch(p0 + 1); stash(p0 + 1);
        fetch(p0 + 2);
        for (int ph = p0; ph < p1; ++ph) {
            stash(ph + 2);
            fetch(ph + 3);
            __syncthreads();
            for (int pw = px; pw < Wp; pw += 32) {
                float4 best;
                unsigned bi[4] = {0, 0, 0, 0};
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const float* row = &smem[((((ph + kh) % 3) * Wc) + pw) * CS + l8 * 4];
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const float4 v = *reinterpret_cast<const float4*>(row + kw * CS);
                        const unsigned k = kh * 3 + kw;
                        if (k == 0) { best = v; continue; }
                        if (v.x > best.x || v.x != v.x) { best.x = v.x; bi[0] = k; }
                        if (v.y > best.y || v.y != v.y) { best.y = v.y; bi[1] = k; }
                        if (v.z > best.z || v.z != v.z) { best.z = v.z; bi[2] = k; }
                        if (v.w > best.w || v.w != v.w) { best.w = v.w; bi[3] = k; }
                    }
                }
                const int64_t o = ((int64_t)ph * Wp + pw) * C;
                best = store_p4(pn + o, best);
                if (in) *reinterpret_cast<uint32_t*>(in + ((int64_t)ph * Wp + pw) * CS) = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
                acc[0][0] += (double)best.x; acc[1][0] += (double)best.x * (double)best.x;
                acc[0][1] += (double)best.y; acc[1][1] += (double)best.y * (double)best.y;
                acc[0][2] += (double)best.z; acc[1][2] += (double)best.z * (double)best.z;
                acc[0][3] += (double)best.w; acc[1][3] += (double)best.w * (double)best.w;
            }
            __syncthreads();
        }
    }
    double* row = partials + (int64_t)slot * 2 * C + (blockIdx.x % ccn) * CS;
    slice_reduce_store(acc, 2, tid, reinterpret_cast<double*>(smem), row, row + C);
}

// ---- 16-bit conv output -> 16-bit pooled activation (the 16-bit modes' blocks 2 and 3): integer keys ----------------------------
// v2 is bound by instruction issue, not by HBM (~38 vector instructions per channel-output: eight compare + select steps on
// (value, index) pairs). For 16-bit inputs the window maximum AND its first position come out of one unsigned maximum over
// 32-bit keys  (value bits << 16) | (8 - tap):  non-negative IEEE values (bf16 or fp16) order like their bit patterns, and
// among equal values the smaller tap has the larger key — ATen's first-maximum rule. v_max3_u32 folds nine keys in four
// instructions; building a key is one v_lshl_or / v_and_or. A key whose value field lies above +inf's pattern flags a NaN
// or a negative value (sign bit set, -0.0 included) in the window: that thread recomputes its four channels with the exact
// float comparisons of v2 (a ReLU output has neither, so this is the rare path). Same thread -> (pixel, channel) mapping,
// same statistics order, same argmax layout as v2: every result, the fp64 partial rows included, is bit-identical to v2's.
template <bool F16>
__device__ __forceinline__ float h16_to_f32(unsigned v16) {
    if (F16) return (float)__builtin_bit_cast(_Float16, (unsigned short)v16);
    return __uint_as_float(v16 << 16);
}

// the exact window of one lane (4 channels): ATen's comparisons on the float values, as v2. in = (p words, argmax word, lane
// flag): lanes whose windows were ordinary get their fast-path result back.
// the exact window of two channels (one 32-bit word of a lane): ATen's comparisons on the float values, as v2. Returns
// (value of channel 0 | value of channel 1 << 16, tap of channel 0 | tap of channel 1 << 8).
template <bool F16>
__device__ __attribute__((noinline)) uint2 pool_window_exact(const char* ring, unsigned o0, unsigned o1, unsigned o2, unsigned loff, unsigned pix_stride) {
    float best[2]; unsigned raw[2], bi[2] = {0, 0};
    for (int kh = 0; kh < 3; ++kh) {
        const char* row = ring + (kh == 0 ? o0 : kh == 1 ? o1 : o2) + loff;
        for (int kw = 0; kw < 3; ++kw) {
            const unsigned v = *reinterpret_cast<const unsigned*>(row + kw * pix_stride);
            const unsigned r2[2] = {v & 0xffffu, v >> 16};
            const unsigned k = kh * 3 + kw;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const float f = h16_to_f32<F16>(r2[c]);
                if (k == 0 || f > best[c] || f != f) { best[c] = f; raw[c] = r2[c]; bi[c] = k; }
            }
        }
    }
    return make_uint2(raw[0] | (raw[1] << 16), bi[0] | (bi[1] << 8));
}

// Addressing: v2 spends as many SCALAR instructions per row (~420: 64-bit index products per pass, `% 3` ring indices) as
// vector ones, and the CU's four SIMDs share one scalar unit. Here every address is a uniform row pointer advanced by a scalar
// add per row plus a loop-invariant 32-bit lane offset, and the three ring rows rotate as three scalar offsets.
// With both instruction streams cut to a third the kernel is what is left: latency-bound — five resident blocks x one conv row
// of loads in flight each. WPL = 32-bit words per lane: 2 (4 channels, 32-channel slices, 64 B per pixel) or 4 (8 channels,
// 64-channel slices, 128 B per pixel: twice the bytes in flight per thread at about the same occupancy). The pixel -> thread
// mapping, hence every per-channel summation order, is v2's in both.
template <bool F16, int NPW, int WPL>
__global__ __launch_bounds__(256) void pool_bnstats_fwd_k16_kernel(const unsigned short* __restrict__ y, unsigned short* __restrict__ p,
                                                                  uint8_t* __restrict__ idx, double* __restrict__ partials,
                                                                  int N, int Hc, int Wc, int C, int bands) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [3][Wc][CSL] 16-bit values; reused by the reduction
    constexpr int CPL = 2 * WPL, CSL = 8 * CPL;                       // channels per lane, per block slice
    typedef unsigned wvec __attribute__((ext_vector_type(WPL)));
    char* ring = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, l8 = tid & 7, px = tid >> 3;
    const int ccn = C / CSL;
    const int slot = blockIdx.x / ccn, c0 = (blockIdx.x % ccn) * CSL + l8 * CPL;
    const int Hp = Hc - 2, Wp = Wc - 2;
    constexpr unsigned INF16 = F16 ? 0x7C00u : 0x7F80u;
    constexpr unsigned PIX = CSL * 2;                                 // bytes per pixel in a ring row
    const unsigned rowbytes = (unsigned)Wc * PIX;
    double acc[2][CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[0][c] = acc[1][c] = 0.0;
    uint32_t yoff[NPW], poff[NPW], ioff[NPW], loff[NPW];              // loop-invariant lane offsets (bytes)
#pragma unroll
    for (int ps = 0; ps < NPW; ++ps) {
        const int x = px + 32 * ps, xc = x < Wc ? x : Wc - 1;
        yoff[ps] = (uint32_t)(((int64_t)xc * C + c0) * 2);
        poff[ps] = (uint32_t)(((int64_t)x * C + c0) * 2);
        ioff[ps] = (uint32_t)(x * CS + (c0 % CS));                    // argmax bytes: 32-channel slices (idx_off)
        loff[ps] = (uint32_t)(x * PIX + l8 * CPL * 2);
    }
    for (int unit = slot; unit < N * bands; unit += (int)gridDim.x / ccn) {
        const int n = unit / bands, band = unit % bands;
        const int p0 = (int)((int64_t)Hp * band / bands), p1 = (int)((int64_t)Hp * (band + 1) / bands);
        const char* yrow = reinterpret_cast<const char*>(y + ((int64_t)n * Hc + p0) * Wc * C);          // conv row being fetched
        char* prow = reinterpret_cast<char*>(p + ((int64_t)n * Hp + p0) * Wp * C);                      // pooled row being written
        char* irow = idx ? reinterpret_cast<char*>(idx + (((int64_t)n * (C / CS) + c0 / CS) * Hp + p0) * Wp * CS) : nullptr;
        const int64_t ystep = (int64_t)Wc * C * 2, pstep = (int64_t)Wp * C * 2, istep = (int64_t)Wp * CS;
        int fr = p0;                                                   // the conv row yrow points at
        wvec ry[NPW];
        auto fetch = [&]() {       // unconditional loads; past the frame's last row that row is fetched again (never consumed)
#pragma unroll
            for (int ps = 0; ps < NPW; ++ps) ry[ps] = *reinterpret_cast<const wvec*>(yrow + yoff[ps]);
            if (fr + 1 < Hc) { yrow += ystep; ++fr; }
        };
        auto stash = [&](unsigned ro) {
#pragma unroll
            for (int ps = 0; ps < NPW; ++ps)
                if (px + 32 * ps < Wc) *reinterpret_cast<wvec*>(ring + ro + loff[ps]) = ry[ps];
        };
        unsigned o0 = 0, o1 = rowbytes, o2 = 2 * rowbytes;            // ring offsets of conv rows ph, ph + 1, ph + 2
        __syncthreads();
        fetch(); stash(o0);
        fetch(); stash(o1);
        fetch();
        for (int ph = p0; ph < p1; ++ph) {
            stash(o2);
            fetch();
            __syncthreads();
#pragma unroll
            for (int ps = 0; ps < NPW; ++ps) {
                if (px + 32 * ps < Wp) {
                    unsigned key[CPL];                            // key[2 j] / key[2 j + 1]: low / high half of word j
#pragma unroll
                    for (int c = 0; c < CPL; ++c) key[c] = 0;
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        const char* row = ring + (kh == 0 ? o0 : kh == 1 ? o1 : o2) + loff[ps];
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const wvec v = *reinterpret_cast<const wvec*>(row + kw * PIX);
                            const unsigned code = 8u - (unsigned)(kh * 3 + kw);
#pragma unroll
                            for (int j = 0; j < WPL; ++j) {
                                key[2 * j] = max(key[2 * j], (v[j] << 16) | code);
                                key[2 * j + 1] = max(key[2 * j + 1], (v[j] & 0xffff0000u) | code);
                            }
                        }
                    }
                    unsigned top = 0;
#pragma unroll
                    for (int c = 0; c < CPL; ++c) top = max(top, key[c]);
                    wvec pw16;                                     // pooled values, two per word
                    unsigned taps[WPL];                            // their taps, one byte each (two per word)
#pragma unroll
                    for (int j = 0; j < WPL; ++j) {
                        pw16[j] = (key[2 * j] >> 16) | (key[2 * j + 1] & 0xffff0000u);
                        taps[j] = (8u - (key[2 * j] & 15u)) | ((8u - (key[2 * j + 1] & 15u)) << 8);
                    }
                    const bool special = (top >> 16) > INF16;
                    if (__builtin_amdgcn_ballot_w64(special) != 0) {          // rare and wave-uniform: real calls, never if-converted
#pragma unroll
                        for (int j = 0; j < WPL; ++j) {
                            const uint2 e = pool_window_exact<F16>(ring, o0, o1, o2, loff[ps] + 4 * j, PIX);
                            if (special) { pw16[j] = e.x; taps[j] = e.y; }
                        }
                    }
                    *reinterpret_cast<wvec*>(prow + poff[ps]) = pw16;
                    if (irow) {
                        if (WPL == 2) *reinterpret_cast<uint32_t*>(irow + ioff[ps]) = taps[0] | (taps[1] << 16);
                        else *reinterpret_cast<uint2*>(irow + ioff[ps]) = make_uint2(taps[0] | (taps[1] << 16), taps[WPL - 2] | (taps[WPL - 1] << 16));
                    }
#pragma unroll
                    for (int j = 0; j < WPL; ++j) {
                        const float f0 = h16_to_f32<F16>(pw16[j] & 0xffffu), f1 = h16_to_f32<F16>(pw16[j] >> 16);
                        acc[0][2 * j] += (double)f0; acc[1][2 * j] += (double)f0 * (double)f0;
                        acc[0][2 * j + 1] += (double)f1; acc[1][2 * j + 1] += (double)f1 * (double)f1;
                    }
                }
            }
            prow += pstep;
            if (irow) irow += istep;
            const unsigned t = o0; o0 = o1; o1 = o2; o2 = t;
            __syncthreads();
        }
    }
    // per-channel sums over the block in v2's order: lane l8's channels over the pixel lanes px = 0 .. 31
    __syncthreads();
    double* red = reinterpret_cast<double*>(smem);                    // [2][256][CPL]
    double* row = partials + (int64_t)slot * 2 * C + (blockIdx.x % ccn) * CSL;
#pragma unroll
    for (int h = 0; h < CPL / 4; ++h) {                               // 4 channels per round: 16 KB of scratch, as v2
        if (h) __syncthreads();
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) red[(a * 256 + tid) * 4 + c] = acc[a][4 * h + c];
        __syncthreads();
        if (tid < 8) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    double t = 0.0;
                    for (int q = tid; q < 256; q += 8) t += red[(a * 256 + q) * 4 + c];
                    row[a * C + tid * CPL + 4 * h + c] = t;
                }
        }
    }
}

// dy (fp32, nullable) and/or dy_pad (bf16 in the zero-padded [N][Hc+2][Wc+2][C] layout of gemm_bf16.hip, nullable)
//
// The gather is branch-free: the ring rows carry two guard pixels on either side and rows outside the frame are guard rows, all
// with an argmax word that matches no tap (0xFF bytes), so each of the nine taps is an unconditional pair of LDS reads, a byte
// compare and a select. (With `continue`s around the out-of-frame taps hipcc built nine basic blocks, each waiting for its own
// LDS reads and juggling exec masks: ~16 vector + 7 scalar instructions and one LDS round trip per tap; the counters show the
// kernel bound by vector-instruction issue, profiles/r02_pool_counters.md.) Skipped taps and taps that add 0.0f give the same
// sum bit for bit (the accumulator starts at +0 and can never be -0). Addresses: uniform row pointers + loop-invariant lane
// offsets, ring rows as three rotating scalar offsets.
// NP = passes of 32 pixels per pooled row, NC = per conv row (launcher: NC = NP or NP + 1).
template <typename DZ, typename PT, int NP, int NC, bool F16>
__global__ __launch_bounds__(256) void bnpool_bwd_v2_kernel(const DZ* __restrict__ dz, const PT* __restrict__ p,
                                                           const uint8_t* __restrict__ idx,
                                                           const float* __restrict__ coef3, float* __restrict__ dy,
                                                           __hip_bfloat16* __restrict__ dy_pad,
                                                           double* __restrict__ dbias_partials, int N, int Hc, int Wc, int C,
                                                           int bands) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // dp [3][Wp+4][CS] floats, then idx [3][Wp+4][CS] bytes
    const int tid = threadIdx.x, l8 = tid & 7, px = tid >> 3;
    const int ccn = C / CS;
    // Blocks b and b + 8 run on the same XCD (round-robin dispatch) at about the same time: give them the two 32-channel
    // slices that share every 128-B line of the bf16 dy (64 B each), so that the XCD's L2 sees both halves of a line.
    unsigned vb = blockIdx.x;
    if ((gridDim.x & 15u) == 0) { const unsigned q = vb >> 4, r = vb & 15u; vb = ((q << 3) + (r & 7u)) * 2 + (r >> 3); }
    const int slot = vb / ccn, sl = vb % ccn, cb = sl * CS, c0 = cb + l8 * 4;
    const int Hp = Hc - 2, Wp = Wc - 2, Wg = Wp + 4;                   // ring row = 2 guard pixels + Wp pooled pixels + 2 guard pixels
    char* ring = reinterpret_cast<char*>(smem);
    char* iring = ring + (size_t)3 * Wg * CS * 4;                     // argmax words [3][Wg][8]
    const unsigned drow = (unsigned)Wg * CS * 4, irow = (unsigned)Wg * CS;
    const float4 ca = *reinterpret_cast<const float4*>(coef3 + c0);
    const float4 cbv = *reinterpret_cast<const float4*>(coef3 + C + c0);
    const float4 cc = *reinterpret_cast<const float4*>(coef3 + 2 * C + c0);
    double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    // loop-invariant lane offsets (bytes): pooled pixel x = px + 32 ps in the input rows / in a ring row; conv pixel w = px + 32 ps
    uint32_t zoff[NP], qoff[NP], ioff[NP], ldp[NP], lix[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
        const int x = px + 32 * ps, xc = x < Wp ? x : Wp - 1;
        zoff[ps] = (uint32_t)(((int64_t)xc * C + c0) * (int64_t)sizeof(DZ));
        qoff[ps] = (uint32_t)(((int64_t)xc * C + c0) * (int64_t)sizeof(PT));
        ioff[ps] = (uint32_t)(xc * CS + l8 * 4);
        ldp[ps] = (uint32_t)(((x + 2) * CS + l8 * 4) * 4);
        lix[ps] = (uint32_t)((x + 2) * CS + l8 * 4);
    }
    uint32_t yoff[NC], poff[NC], gdp[NC], gix[NC];
#pragma unroll
    for (int ps = 0; ps < NC; ++ps) {
        const int w = px + 32 * ps, wc = w < Wc ? w : Wc - 1;          // lanes past the row gather a clamped pixel and store nothing
        yoff[ps] = (uint32_t)(((int64_t)wc * C + c0) * 4);
        poff[ps] = (uint32_t)(((int64_t)(wc + 1) * C + c0) * 2);
        gdp[ps] = (uint32_t)(((wc + 2) * CS + l8 * 4) * 4);           // tap dw reads the ring pixel (w + 2 - dw)
        gix[ps] = (uint32_t)((wc + 2) * CS + l8 * 4);
    }
    for (int unit = slot; unit < N * bands; unit += (int)gridDim.x / ccn) {       // (frame, band of dy rows), as in the forward
        const int n = unit / bands, band = unit % bands;
        const int h0 = (int)((int64_t)Hc * band / bands), h1 = (int)((int64_t)Hc * (band + 1) / bands);
        const int hfirst = h0 > 2 ? h0 - 2 : 0;                  // dy row h gathers from pooled rows h-2..h: two warm-up rows
        const char* zrow = reinterpret_cast<const char*>(dz + ((int64_t)n * Hp + hfirst) * Wp * C);
        const char* qrow = reinterpret_cast<const char*>(p + ((int64_t)n * Hp + hfirst) * Wp * C);
        const char* xrow = reinterpret_cast<const char*>(idx + (((int64_t)n * ccn + sl) * Hp + hfirst) * Wp * CS);   // slice-major, see idx_off
        const int64_t zstep = (int64_t)Wp * C * sizeof(DZ), qstep = (int64_t)Wp * C * sizeof(PT), xstep = (int64_t)Wp * CS;
        char* dyr = dy ? reinterpret_cast<char*>(dy + ((int64_t)n * Hc + h0) * Wc * C) : nullptr;
        char* dpr = dy_pad ? reinterpret_cast<char*>(dy_pad + (((int64_t)n * (Hc + 2) + h0 + 1) * (Wc + 2)) * C) : nullptr;
        const int64_t ystep = (int64_t)Wc * C * 4, pstep = (int64_t)(Wc + 2) * C * 2;
        __syncthreads();                                           // the previous unit's last gather
        for (int i = tid; i < 3 * Wg * 8; i += 256) reinterpret_cast<uint32_t*>(iring)[i] = 0xFFFFFFFFu;       // every ring pixel: no tap
        __syncthreads();
        // Software pipeline over rows: the loads of pooled row h+1 (dz, p, argmax) are issued into registers before row h is
        // gathered and land while it is computed; every load is unconditional, from a clamped address past the end (never
        // consumed), so that hipcc can count the loads in flight.
        decltype(load_dz4_raw(dz)) rd[NP];
        decltype(load_dz4_raw(p)) rq[NP];
        uint32_t ri[NP];
        int fr = hfirst;                                           // the pooled row the input pointers are at
        auto prefetch = [&]() {
#pragma unroll
            for (int ps = 0; ps < NP; ++ps) {
                rd[ps] = load_dz4_raw(reinterpret_cast<const DZ*>(zrow + zoff[ps]));
                rq[ps] = load_dz4_raw(reinterpret_cast<const PT*>(qrow + qoff[ps]));
                ri[ps] = *reinterpret_cast<const uint32_t*>(xrow + ioff[ps]);
            }
            if (fr + 1 < Hp) { zrow += zstep; qrow += qstep; xrow += xstep; ++fr; }
        };
        unsigned o0 = 0, o1 = 1, o2 = 2;                           // ring slots of pooled rows h, h - 1, h - 2
        prefetch();
        for (int h = hfirst; h < h1; ++h) {
            char* drow0 = ring + o0 * drow;
            char* irow0 = iring + o0 * irow;
            if (h < Hp) {       // pooled row h enters the rolling buffer as dp = a*dz + b*p + c
#pragma unroll
                for (int ps = 0; ps < NP; ++ps) {
                    if (px + 32 * ps < Wp) {
                        const float4 d = dz4_of(rd[ps]), q = dz4_of(rq[ps]);
                        float4 v;
                        // the ReLU mask rides on p: every window whose argmax is a given conv pixel has p equal to that pixel's y
                        v.x = q.x > 0.f ? fmaf(ca.x, d.x, fmaf(cbv.x, q.x, cc.x)) : 0.f;
                        v.y = q.y > 0.f ? fmaf(ca.y, d.y, fmaf(cbv.y, q.y, cc.y)) : 0.f;
                        v.z = q.z > 0.f ? fmaf(ca.z, d.z, fmaf(cbv.z, q.z, cc.z)) : 0.f;
                        v.w = q.w > 0.f ? fmaf(ca.w, d.w, fmaf(cbv.w, q.w, cc.w)) : 0.f;
                        *reinterpret_cast<float4*>(drow0 + ldp[ps]) = v;
                        *reinterpret_cast<uint32_t*>(irow0 + lix[ps]) = ri[ps];
                    }
                }
            } else {            // past the last pooled row: a guard row
#pragma unroll
                for (int ps = 0; ps < NP; ++ps)
                    if (px + 32 * ps < Wp) *reinterpret_cast<uint32_t*>(irow0 + lix[ps]) = 0xFFFFFFFFu;
            }
            if (h + 1 < h1) prefetch();
            __syncthreads();
            if (h >= h0) {
#pragma unroll
                for (int ps = 0; ps < NC; ++ps) {
                    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int dh = 0; dh < 3; ++dh) {
                        const unsigned o = dh == 0 ? o0 : dh == 1 ? o1 : o2;
                        const char* dr = ring + o * drow + gdp[ps];
                        const char* ir = iring + o * irow + gix[ps];
#pragma unroll
                        for (int dw = 0; dw < 3; ++dw) {
                            const float4 v = *reinterpret_cast<const float4*>(dr - dw * (CS * 4));
                            const uint32_t ii = *reinterpret_cast<const uint32_t*>(ir - dw * CS);
                            const unsigned k = dh * 3 + dw;       // (SDWA byte compares into scalar masks via inline asm: 107 instead of
                                                                  // 136 vector instructions per pass, and 5 % slower)
                            a4.x += (ii & 0xffu) == k ? v.x : 0.f;
                            a4.y += ((ii >> 8) & 0xffu) == k ? v.y : 0.f;
                            a4.z += ((ii >> 16) & 0xffu) == k ? v.z : 0.f;
                            a4.w += (ii >> 24) == k ? v.w : 0.f;
                        }
                    }
                    if (px + 32 * ps < Wc) {
                        if (dyr) *reinterpret_cast<float4*>(dyr + yoff[ps]) = a4;
                        if (dpr) *reinterpret_cast<uint2*>(dpr + poff[ps]) = make_uint2(pack_h16x2<F16>(a4.x, a4.y), pack_h16x2<F16>(a4.z, a4.w));
                        acc[0][0] += (double)a4.x; acc[0][1] += (double)a4.y; acc[0][2] += (double)a4.z; acc[0][3] += (double)a4.w;
                    }
                }
                if (dyr) dyr += ystep;
                if (dpr) dpr += pstep;
            }
            const unsigned t = o2; o2 = o1; o1 = o0; o0 = t;      // next row goes where row h - 2 was
            __syncthreads();
        }
    }
    double* row = dbias_partials + (int64_t)slot * C + cb;
    slice_reduce_store(acc, 1, tid, reinterpret_cast<double*>(smem), row, row);
}

__global__ __launch_bounds__(256) void partials_sum_kernel(const double* __restrict__ partials, int nparts, int64_t stride,
                                                          int C, float* __restrict__ out) {
    __shared__ double sm[256];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const double s = column_sum16(partials + c, nparts, stride, c < C, sm);
    if (c < C && threadIdx.x < 16) out[c] = (float)s;
}

// two partial-sum arrays in one launch (the conv3 / conv2 bias gradients of the 10-frame step): blocks [0, blocks_a) serve the first
__global__ __launch_bounds__(256) void partials_sum2_kernel(const double* __restrict__ pa, int na, int Ca, float* __restrict__ oa, int blocks_a,
                                                           const double* __restrict__ pb, int nb, int Cb, float* __restrict__ ob) {
    __shared__ double sm[256];
    const bool first = (int)blockIdx.x < blocks_a;
    const double* partials = first ? pa : pb;
    const int nparts = first ? na : nb, C = first ? Ca : Cb;
    float* out = first ? oa : ob;
    const int c = (first ? blockIdx.x : blockIdx.x - blocks_a) * 16 + (threadIdx.x & 15);
    const double s = column_sum16(partials + c, nparts, C, c < C, sm);
    if (c < C && threadIdx.x < 16) out[c] = (float)s;
}

__global__ __launch_bounds__(256) void partials_sum_f64_kernel(const double* __restrict__ partials, int nparts, int64_t stride,
                                                              int C, double* __restrict__ out) {
    __shared__ double sm[256];
    const int c = blockIdx.x * 16 + (threadIdx.x & 15);
    const double s = column_sum16(partials + c, nparts, stride, c < C, sm);
    if (c < C && threadIdx.x < 16) out[c] = s;
}

// partial rows beyond one per frame are spent on row bands inside each frame (>= 3 rows per band)
int row_bands(int nparts, int N, int rows) {
    int b = nparts / N;
    if (b > rows / 3) b = rows / 3;
    return b < 1 ? 1 : b;
}

template <typename YT, typename PT>
static void launch_pool_fwd_v2(int nparts, size_t lds, hipStream_t st, const YT* y, PT* p, uint8_t* idx, double* partials, int N, int Hc, int Wc, int C) {
    const dim3 grid(nparts * (C / CS)), block(256);
    const int bands = row_bands(nparts, N, Hc - 2);
    switch ((Wc + 31) / 32) {                                    // passes of 32 pixels per conv row; the LDS limit keeps Wc <= 170
    case 1: hipLaunchKernelGGL((pool_bnstats_fwd_v2_kernel<YT, PT, 1>), grid, block, lds, st, y, p, idx, partials, N, Hc, Wc, C, bands); break;
    case 2: hipLaunchKernelGGL((pool_bnstats_fwd_v2_kernel<YT, PT, 2>), grid, block, lds, st, y, p, idx, partials, N, Hc, Wc, C, bands); break;
    case 3: hipLaunchKernelGGL((pool_bnstats_fwd_v2_kernel<YT, PT, 3>), grid, block, lds, st, y, p, idx, partials, N, Hc, Wc, C, bands); break;
    case 4: hipLaunchKernelGGL((pool_bnstats_fwd_v2_kernel<YT, PT, 4>), grid, block, lds, st, y, p, idx, partials, N, Hc, Wc, C, bands); break;
    case 5: hipLaunchKernelGGL((pool_bnstats_fwd_v2_kernel<YT, PT, 5>), grid, block, lds, st, y, p, idx, partials, N, Hc, Wc, C, bands); break;
    default: hipLaunchKernelGGL((pool_bnstats_fwd_v2_kernel<YT, PT, 6>), grid, block, lds, st, y, p, idx, partials, N, Hc, Wc, C, bands); break;
    }
}

template <bool F16, int WPL>
static void launch_pool_fwd_k16_w(int nparts, hipStream_t st, const void* y, void* p, uint8_t* idx, double* partials, int N, int Hc, int Wc, int C) {
    constexpr int CSL = 16 * WPL;
    const dim3 grid(nparts * (C / CSL)), block(256);
    const int bands = row_bands(nparts, N, Hc - 2);
    size_t lds = (size_t)3 * Wc * CSL * 2;
    if (lds < 16384) lds = 16384;                               // the fp64 block reduction reuses the buffer (256 x 8 doubles)
    const unsigned short* yy = (const unsigned short*)y;
    unsigned short* pp = (unsigned short*)p;
    switch ((Wc + 31) / 32) {
    case 1: hipLaunchKernelGGL((pool_bnstats_fwd_k16_kernel<F16, 1, WPL>), grid, block, lds, st, yy, pp, idx, partials, N, Hc, Wc, C, bands); break;
    case 2: hipLaunchKernelGGL((pool_bnstats_fwd_k16_kernel<F16, 2, WPL>), grid, block, lds, st, yy, pp, idx, partials, N, Hc, Wc, C, bands); break;
    case 3: hipLaunchKernelGGL((pool_bnstats_fwd_k16_kernel<F16, 3, WPL>), grid, block, lds, st, yy, pp, idx, partials, N, Hc, Wc, C, bands); break;
    case 4: hipLaunchKernelGGL((pool_bnstats_fwd_k16_kernel<F16, 4, WPL>), grid, block, lds, st, yy, pp, idx, partials, N, Hc, Wc, C, bands); break;
    case 5: hipLaunchKernelGGL((pool_bnstats_fwd_k16_kernel<F16, 5, WPL>), grid, block, lds, st, yy, pp, idx, partials, N, Hc, Wc, C, bands); break;
    default: hipLaunchKernelGGL((pool_bnstats_fwd_k16_kernel<F16, 6, WPL>), grid, block, lds, st, yy, pp, idx, partials, N, Hc, Wc, C, bands); break;
    }
}

template <bool F16>
static void launch_pool_fwd_k16(int nparts, hipStream_t st, const void* y, void* p, uint8_t* idx, double* partials, int N, int Hc, int Wc, int C) {
    const char* w = getenv("GOALNET_POOL_K16_WPL");             // 2 | 4: A/B runs
    const bool wide = C % 64 == 0 && (size_t)3 * Wc * 64 * 2 <= 64 * 1024 && !(w && w[0] == '2');
    if (wide) launch_pool_fwd_k16_w<F16, 4>(nparts, st, y, p, idx, partials, N, Hc, Wc, C);
    else launch_pool_fwd_k16_w<F16, 2>(nparts, st, y, p, idx, partials, N, Hc, Wc, C);
}

template <typename DZ, typename PT, bool F16 = false>
static void launch_bnpool_bwd_v2(int nparts, size_t lds, hipStream_t st, const DZ* dz, const PT* p, const uint8_t* idx, const float* coef3,
                                 float* dy, __hip_bfloat16* dy_pad, double* dbias_partials, int N, int Hc, int Wc, int C) {
    const dim3 grid(nparts * (C / CS)), block(256);
    const int bands = row_bands(nparts, N, Hc);
    const int np = (Wc - 2 + 31) / 32, nc = (Wc + 31) / 32;    // passes of 32 pixels per pooled / conv row; the LDS limit keeps Wp <= 132
#define GN_BWD2(NPV, NCV) hipLaunchKernelGGL((bnpool_bwd_v2_kernel<DZ, PT, NPV, NCV, F16>), grid, block, lds, st, dz, p, idx, coef3, dy, dy_pad, dbias_partials, N, Hc, Wc, C, bands)
    switch (np * 2 + (nc - np)) {
    case 2: GN_BWD2(1, 1); break;  case 3: GN_BWD2(1, 2); break;
    case 4: GN_BWD2(2, 2); break;  case 5: GN_BWD2(2, 3); break;
    case 6: GN_BWD2(3, 3); break;  case 7: GN_BWD2(3, 4); break;
    case 8: GN_BWD2(4, 4); break;  case 9: GN_BWD2(4, 5); break;
    case 10: GN_BWD2(5, 5); break; default: GN_BWD2(5, 6); break;
    }
#undef GN_BWD2
}

bool chan_ok(int C) { return C >= 4 && C <= 1024 && (C & 3) == 0 && (256 % (C >> 2)) == 0; }

}  // namespace

extern "C" {

int goalnet_stat_parts(int64_t units) { return units < 1 ? 1 : units > PARTS ? PARTS : (int)units; }

#define GN_PARTS_OK(name) GN_REQUIRE(nparts >= 1 && nparts <= PARTS, GOALNET_E_SHAPE, name ": nparts=%d must be in [1, %d]", nparts, PARTS)

int goalnet_pool_bnstats_fwd(const float* y, float* p, uint8_t* idx, double* partials, int nparts,
                             int N, int Hc, int Wc, int C, void* stream) {
    GN_REQUIRE(y && p && partials, GOALNET_E_NULL, "pool_bnstats_fwd: null pointer");
    GN_PARTS_OK("pool_bnstats_fwd");
    GN_REQUIRE(N > 0 && Hc >= 3 && Wc >= 3, GOALNET_E_SHAPE, "pool_bnstats_fwd: need Hc, Wc >= 3");
    GN_REQUIRE(chan_ok(C), GOALNET_E_SHAPE, "pool_bnstats_fwd: C=%d must be 4*2^k, <= 1024", C);
    GN_REQUIRE(aligned16(y) && aligned16(p) && (reinterpret_cast<uintptr_t>(idx) & 3u) == 0, GOALNET_E_ALIGN,
               "pool_bnstats_fwd: pointers must be 16-byte aligned");
    const size_t lds = (size_t)3 * Wc * CS * sizeof(float);
    if (C % CS == 0 && lds <= 64 * 1024 && !getenv("GOALNET_POOL_V1")) {
        const size_t need = lds < 16384 ? 16384 : lds;      // the fp64 block reduction reuses the buffer (256 x 8 doubles)
        launch_pool_fwd_v2<float, float>(nparts, need, (hipStream_t)stream, y, p, idx, partials, N, Hc, Wc, C);
    } else {
        hipLaunchKernelGGL(pool_bnstats_fwd_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, y, p, idx, partials, N, Hc, Wc, C);
    }
    GN_LAUNCH_CHECK("pool_bnstats_fwd");
    return 0;
}

/* p stored as bf16; y fp32 (y_bf16 = 0) or bf16 as goalnet_conv3x3_fwd_bf16p_o16 writes it (rounding is monotonic: the
 * maximum of the rounded values is the rounded maximum, so p is the same either way; only ties in the argmax differ) */
int goalnet_pool_bnstats_fwd_p16(const void* y, int y_bf16, void* p_bf16, uint8_t* idx, double* partials, int nparts,
                                 int N, int Hc, int Wc, int C, int f16, void* stream) {
    GN_REQUIRE(y && p_bf16 && partials, GOALNET_E_NULL, "pool_bnstats_fwd_p16: null pointer");
    GN_PARTS_OK("pool_bnstats_fwd_p16");
    GN_REQUIRE(N > 0 && Hc >= 3 && Wc >= 3, GOALNET_E_SHAPE, "pool_bnstats_fwd_p16: need Hc, Wc >= 3");
    GN_REQUIRE(chan_ok(C) && C % CS == 0, GOALNET_E_SHAPE, "pool_bnstats_fwd_p16: C=%d must be 32*2^k, <= 1024", C);
    GN_REQUIRE(aligned16(y) && aligned16(p_bf16) && (reinterpret_cast<uintptr_t>(idx) & 3u) == 0, GOALNET_E_ALIGN,
               "pool_bnstats_fwd_p16: pointers must be 16-byte aligned");
    const size_t lds = (size_t)3 * Wc * CS * sizeof(float);
    GN_REQUIRE(lds <= 64 * 1024, GOALNET_E_SHAPE, "pool_bnstats_fwd_p16: image too wide for the rolling LDS rows");
    const size_t need = lds < 16384 ? 16384 : lds;
    typedef __hip_bfloat16 bf;
    typedef _Float16 hf;
    if (y_bf16 && !getenv("GOALNET_POOL_K16_OFF")) {          // 16-bit in, 16-bit out: the integer-key kernel (bit-identical to v2)
        if (f16) launch_pool_fwd_k16<true>(nparts, (hipStream_t)stream, y, p_bf16, idx, partials, N, Hc, Wc, C);
        else launch_pool_fwd_k16<false>(nparts, (hipStream_t)stream, y, p_bf16, idx, partials, N, Hc, Wc, C);
    } else if (f16) {
        if (y_bf16) launch_pool_fwd_v2<hf, hf>(nparts, need, (hipStream_t)stream, (const hf*)y, (hf*)p_bf16, idx, partials, N, Hc, Wc, C);
        else launch_pool_fwd_v2<float, hf>(nparts, need, (hipStream_t)stream, (const float*)y, (hf*)p_bf16, idx, partials, N, Hc, Wc, C);
    } else if (y_bf16) launch_pool_fwd_v2<bf, bf>(nparts, need, (hipStream_t)stream, (const bf*)y, (bf*)p_bf16, idx, partials, N, Hc, Wc, C);
    else launch_pool_fwd_v2<float, bf>(nparts, need, (hipStream_t)stream, (const float*)y, (bf*)p_bf16, idx, partials, N, Hc, Wc, C);
    GN_LAUNCH_CHECK("pool_bnstats_fwd_p16");
    return 0;
}

int goalnet_bn_finalize(const double* partials, int nparts, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps, int64_t count,
                        int C, float* mean, float* invstd, float* scale, float* shift, void* stream) {
    GN_REQUIRE(partials && gamma && beta && mean && invstd && scale && shift, GOALNET_E_NULL, "bn_finalize: null pointer");
    GN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), GOALNET_E_NULL, "bn_finalize: running stats must both be set or both NULL");
    GN_REQUIRE(C > 0 && count > 0, GOALNET_E_SHAPE, "bn_finalize: bad dims");
    GN_PARTS_OK("bn_finalize");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, partials, gamma, beta,
                       running_mean, running_var, momentum, eps, (double)count, nparts, C, mean, invstd, scale, shift);
    GN_LAUNCH_CHECK("bn_finalize");
    return 0;
}

int goalnet_bn_bwd_reduce(const float* dz, const float* p, const float* mean, const float* invstd,
                          double* partials, int nparts, int64_t npix, int C, void* stream) {
    GN_REQUIRE(dz && p && mean && invstd && partials, GOALNET_E_NULL, "bn_bwd_reduce: null pointer");
    GN_REQUIRE(npix > 0 && chan_ok(C), GOALNET_E_SHAPE, "bn_bwd_reduce: bad dims");
    GN_PARTS_OK("bn_bwd_reduce");
    GN_REQUIRE(aligned16(dz) && aligned16(p) && aligned16(mean) && aligned16(invstd), GOALNET_E_ALIGN, "bn_bwd_reduce: alignment");
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, float>), dim3(nparts), dim3(256), 0, (hipStream_t)stream, dz, p, mean, invstd, partials, npix, C);
    GN_LAUNCH_CHECK("bn_bwd_reduce");
    return 0;
}

/* dz and p each fp32 (flag 0) or bf16 (flag 1) */
int goalnet_bn_bwd_reduce_t(const void* dz, int dz_bf16, const void* p, int p_bf16, const float* mean, const float* invstd,
                            double* partials, int nparts, int64_t npix, int C, int f16, void* stream) {
    GN_REQUIRE(dz && p && mean && invstd && partials, GOALNET_E_NULL, "bn_bwd_reduce_t: null pointer");
    GN_REQUIRE(npix > 0 && chan_ok(C), GOALNET_E_SHAPE, "bn_bwd_reduce_t: bad dims");
    GN_PARTS_OK("bn_bwd_reduce_t");
    GN_REQUIRE(aligned16(dz) && aligned16(p) && aligned16(mean) && aligned16(invstd), GOALNET_E_ALIGN, "bn_bwd_reduce_t: alignment");
    const dim3 grid(nparts), block(256);
    hipStream_t st = (hipStream_t)stream;
    typedef __hip_bfloat16 bf;
    typedef _Float16 hf;
    if (f16 && dz_bf16 && p_bf16) hipLaunchKernelGGL((bn_bwd_reduce_kernel<hf, hf>), grid, block, 0, st, (const hf*)dz, (const hf*)p, mean, invstd, partials, npix, C);
    else if (f16 && dz_bf16)      hipLaunchKernelGGL((bn_bwd_reduce_kernel<hf, float>), grid, block, 0, st, (const hf*)dz, (const float*)p, mean, invstd, partials, npix, C);
    else if (f16 && p_bf16)       hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, hf>), grid, block, 0, st, (const float*)dz, (const hf*)p, mean, invstd, partials, npix, C);
    else if (dz_bf16 && p_bf16) hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf, bf>), grid, block, 0, st, (const bf*)dz, (const bf*)p, mean, invstd, partials, npix, C);
    else if (dz_bf16)      hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf, float>), grid, block, 0, st, (const bf*)dz, (const float*)p, mean, invstd, partials, npix, C);
    else if (p_bf16)       hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, bf>), grid, block, 0, st, (const float*)dz, (const bf*)p, mean, invstd, partials, npix, C);
    else                   hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, float>), grid, block, 0, st, (const float*)dz, (const float*)p, mean, invstd, partials, npix, C);
    GN_LAUNCH_CHECK("bn_bwd_reduce_t");
    return 0;
}

int goalnet_bn_bwd_finalize(const double* partials, int nparts, const float* gamma, const float* mean, const float* invstd,
                            int64_t count, int C, float* dgamma, float* dbeta, float* coef3, void* stream) {
    GN_REQUIRE(partials && gamma && mean && invstd && dgamma && dbeta && coef3, GOALNET_E_NULL, "bn_bwd_finalize: null pointer");
    GN_REQUIRE(C > 0 && count > 0, GOALNET_E_SHAPE, "bn_bwd_finalize: bad dims");
    GN_PARTS_OK("bn_bwd_finalize");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, partials, gamma, mean,
                       invstd, (double)count, nparts, C, dgamma, dbeta, coef3);
    GN_LAUNCH_CHECK("bn_bwd_finalize");
    return 0;
}

int goalnet_bnpool_bwd(const float* dz, const float* p, const uint8_t* idx, const float* coef3,
                       float* dy, double* dbias_partials, int nparts, int N, int Hc, int Wc, int C, void* stream) {
    GN_REQUIRE(dz && p && idx && coef3 && dy && dbias_partials, GOALNET_E_NULL, "bnpool_bwd: null pointer");
    GN_PARTS_OK("bnpool_bwd");
    GN_REQUIRE(N > 0 && Hc >= 3 && Wc >= 3 && chan_ok(C), GOALNET_E_SHAPE, "bnpool_bwd: bad dims");
    GN_REQUIRE(aligned16(dz) && aligned16(p) && aligned16(dy) && aligned16(coef3) &&
               (reinterpret_cast<uintptr_t>(idx) & 3u) == 0, GOALNET_E_ALIGN, "bnpool_bwd: alignment");
    const size_t lds = (size_t)3 * (Wc + 2) * CS * (sizeof(float) + 1);      // three ring rows of Wp + 4 pixels: value + argmax byte
    if (C % CS == 0 && lds <= 64 * 1024 && !getenv("GOALNET_POOL_V1")) {
        const size_t need = lds < 8192 ? 8192 : lds;
        launch_bnpool_bwd_v2<float, float>(nparts, need, (hipStream_t)stream, dz, p, idx, coef3, dy, (__hip_bfloat16*)nullptr, dbias_partials, N, Hc, Wc, C);
    } else {
        hipLaunchKernelGGL(bnpool_bwd_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, dz, p, idx, coef3, dy,
                           dbias_partials, N, Hc, Wc, C);
    }
    GN_LAUNCH_CHECK("bnpool_bwd");
    return 0;
}

int goalnet_bnpool_bwd_bf16p(const float* dz, const float* p, const uint8_t* idx, const float* coef3,
                             float* dy, void* dy_pad_bf16, double* dbias_partials, int nparts, int N, int Hc, int Wc, int C,
                             int f16, void* stream) {
    GN_REQUIRE(dz && p && idx && coef3 && dy_pad_bf16 && dbias_partials, GOALNET_E_NULL, "bnpool_bwd_bf16p: null pointer");
    GN_PARTS_OK("bnpool_bwd_bf16p");
    GN_REQUIRE(N > 0 && Hc >= 3 && Wc >= 3 && chan_ok(C) && C % CS == 0, GOALNET_E_SHAPE, "bnpool_bwd_bf16p: bad dims (C %% 32)");
    GN_REQUIRE(aligned16(dz) && aligned16(p) && aligned16(dy) && aligned16(coef3) && aligned16(dy_pad_bf16) &&
               (reinterpret_cast<uintptr_t>(idx) & 3u) == 0, GOALNET_E_ALIGN, "bnpool_bwd_bf16p: alignment");
    const size_t lds = (size_t)3 * (Wc + 2) * CS * (sizeof(float) + 1);      // three ring rows of Wp + 4 pixels: value + argmax byte
    GN_REQUIRE(lds <= 64 * 1024, GOALNET_E_SHAPE, "bnpool_bwd_bf16p: image too wide for the rolling LDS rows");
    const size_t need = lds < 8192 ? 8192 : lds;
    if (f16) launch_bnpool_bwd_v2<float, float, true>(nparts, need, (hipStream_t)stream, dz, p, idx, coef3, dy, (__hip_bfloat16*)dy_pad_bf16, dbias_partials, N, Hc, Wc, C);
    else launch_bnpool_bwd_v2<float, float>(nparts, need, (hipStream_t)stream, dz, p, idx, coef3, dy, (__hip_bfloat16*)dy_pad_bf16, dbias_partials, N, Hc, Wc, C);
    GN_LAUNCH_CHECK("bnpool_bwd_bf16p");
    return 0;
}

/* dz and p each fp32 (flag 0) or bf16 (flag 1) */
int goalnet_bnpool_bwd_bf16p_t(const void* dz, int dz_bf16, const void* p, int p_bf16, const uint8_t* idx, const float* coef3,
                               float* dy, void* dy_pad_bf16, double* dbias_partials, int nparts, int N, int Hc, int Wc, int C,
                               int f16, void* stream) {
    GN_REQUIRE(dz && p && idx && coef3 && (dy || dy_pad_bf16) && dbias_partials, GOALNET_E_NULL, "bnpool_bwd_bf16p_t: null pointer");
    GN_PARTS_OK("bnpool_bwd_bf16p_t");
    GN_REQUIRE(N > 0 && Hc >= 3 && Wc >= 3 && chan_ok(C) && C % CS == 0, GOALNET_E_SHAPE, "bnpool_bwd_bf16p_t: bad dims (C %% 32)");
    GN_REQUIRE(aligned16(dz) && aligned16(p) && aligned16(dy) && aligned16(coef3) && aligned16(dy_pad_bf16) &&
               (reinterpret_cast<uintptr_t>(idx) & 3u) == 0, GOALNET_E_ALIGN, "bnpool_bwd_bf16p_t: alignment");
    const size_t lds = (size_t)3 * (Wc + 2) * CS * (sizeof(float) + 1);      // three ring rows of Wp + 4 pixels: value + argmax byte
    GN_REQUIRE(lds <= 64 * 1024, GOALNET_E_SHAPE, "bnpool_bwd_bf16p_t: image too wide for the rolling LDS rows");
    const size_t need = lds < 8192 ? 8192 : lds;
    hipStream_t st = (hipStream_t)stream;
    typedef __hip_bfloat16 bf;
    bf* dp = (bf*)dy_pad_bf16;
    typedef _Float16 hf;
    if (f16 && dz_bf16 && p_bf16) launch_bnpool_bwd_v2<hf, hf, true>(nparts, need, st, (const hf*)dz, (const hf*)p, idx, coef3, dy, dp, dbias_partials, N, Hc, Wc, C);
    else if (f16 && dz_bf16)      launch_bnpool_bwd_v2<hf, float, true>(nparts, need, st, (const hf*)dz, (const float*)p, idx, coef3, dy, dp, dbias_partials, N, Hc, Wc, C);
    else if (f16 && p_bf16)       launch_bnpool_bwd_v2<float, hf, true>(nparts, need, st, (const float*)dz, (const hf*)p, idx, coef3, dy, dp, dbias_partials, N, Hc, Wc, C);
    else if (f16)                 launch_bnpool_bwd_v2<float, float, true>(nparts, need, st, (const float*)dz, (const float*)p, idx, coef3, dy, dp, dbias_partials, N, Hc, Wc, C);
    else if (dz_bf16 && p_bf16) launch_bnpool_bwd_v2<bf, bf>(nparts, need, st, (const bf*)dz, (const bf*)p, idx, coef3, dy, dp, dbias_partials, N, Hc, Wc, C);
    else if (dz_bf16)      launch_bnpool_bwd_v2<bf, float>(nparts, need, st, (const bf*)dz, (const float*)p, idx, coef3, dy, dp, dbias_partials, N, Hc, Wc, C);
    else if (p_bf16)       launch_bnpool_bwd_v2<float, bf>(nparts, need, st, (const float*)dz, (const bf*)p, idx, coef3, dy, dp, dbias_partials, N, Hc, Wc, C);
    else                   launch_bnpool_bwd_v2<float, float>(nparts, need, st, (const float*)dz, (const float*)p, idx, coef3, dy, dp, dbias_partials, N, Hc, Wc, C);
    GN_LAUNCH_CHECK("bnpool_bwd_bf16p_t");
    return 0;
}

/* ---- small shapes (the reference's 10-frame sub-batches at 40 x 40, main.py:177-196): see the kernels' comment. */
static int small_blocks(int64_t pixels, int C) {
    const int ppi = SMALL_T / (C >> 2);
    int64_t nb = (pixels + 3 * ppi - 1) / (3 * ppi);          // about three pixels per thread
    return nb < 1 ? 1 : nb > SMALL_MAXB ? SMALL_MAXB : (int)nb;
}
static bool small_ok(int N, int Hc, int Wc, int C) {
    return chan_ok(C) && C <= 512 && (int64_t)N * Hc * Wc * C <= (1ll << 20);
}

size_t goalnet_bn_small_ws_bytes(int C) { return (size_t)SMALL_MAXB * 2 * C * sizeof(double); }

int goalnet_pool_bn_fwd_small(const float* y, float* p, uint8_t* idx, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps,
                              float* mean, float* invstd, float* scale, float* shift,
                              void* ws, size_t ws_bytes, int* ctr, int N, int Hc, int Wc, int C, void* stream) {
    GN_REQUIRE(y && p && gamma && beta && mean && invstd && scale && shift && ws && ctr, GOALNET_E_NULL, "pool_bn_fwd_small: null pointer");
    GN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), GOALNET_E_NULL, "pool_bn_fwd_small: running stats must both be set or both NULL");
    GN_REQUIRE(N > 0 && Hc >= 3 && Wc >= 3 && small_ok(N, Hc, Wc, C), GOALNET_E_SHAPE, "pool_bn_fwd_small: need Hc, Wc >= 3, C = 4*2^k <= 512, <= 2^20 elements");
    GN_REQUIRE(aligned16(y) && aligned16(p) && (reinterpret_cast<uintptr_t>(idx) & 3u) == 0 && (reinterpret_cast<uintptr_t>(ws) & 7u) == 0, GOALNET_E_ALIGN,
               "pool_bn_fwd_small: pointers must be 16-byte aligned");
    GN_REQUIRE(ws_bytes >= goalnet_bn_small_ws_bytes(C), GOALNET_E_WORKSPACE, "pool_bn_fwd_small: workspace too small");
    const int nb = small_blocks((int64_t)N * (Hc - 2) * (Wc - 2), C);
    hipLaunchKernelGGL(pool_bn_fwd_small_kernel, dim3(nb), dim3(SMALL_T), SMALL_T * 4 * sizeof(double), (hipStream_t)stream, y, p, idx,
                       (double*)ws, ctr, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift, N, Hc, Wc, C);
    GN_LAUNCH_CHECK("pool_bn_fwd_small");
    return 0;
}

int goalnet_bn_bwd_reduce_small(const float* dz, const float* p, const float* mean, const float* invstd, const float* gamma,
                                float* dgamma, float* dbeta, float* coef3, void* ws, size_t ws_bytes, int* ctr,
                                int N, int Hc, int Wc, int C, void* stream) {
    GN_REQUIRE(dz && p && mean && invstd && gamma && dgamma && dbeta && coef3 && ws && ctr, GOALNET_E_NULL, "bn_bwd_reduce_small: null pointer");
    GN_REQUIRE(N > 0 && Hc >= 3 && Wc >= 3 && small_ok(N, Hc, Wc, C), GOALNET_E_SHAPE, "bn_bwd_reduce_small: bad dims");
    GN_REQUIRE(aligned16(dz) && aligned16(p) && aligned16(coef3) && aligned16(mean) && aligned16(invstd) && (reinterpret_cast<uintptr_t>(ws) & 7u) == 0,
               GOALNET_E_ALIGN, "bn_bwd_reduce_small: alignment");
    GN_REQUIRE(ws_bytes >= goalnet_bn_small_ws_bytes(C), GOALNET_E_WORKSPACE, "bn_bwd_reduce_small: workspace too small");
    const int npool = N * (Hc - 2) * (Wc - 2);
    hipLaunchKernelGGL(bn_bwd_reduce_small_kernel, dim3(small_blocks(npool, C)), dim3(SMALL_T), SMALL_T * 4 * sizeof(double), (hipStream_t)stream,
                       dz, p, mean, invstd, (double*)ws, ctr, gamma, dgamma, dbeta, coef3, npool, C);
    GN_LAUNCH_CHECK("bn_bwd_reduce_small");
    return 0;
}

int goalnet_bnpool_bwd_small(const float* dz, const float* p, const uint8_t* idx, const float* coef3, float* dy, float* dbias,
                             void* ws, size_t ws_bytes, int* ctr, int N, int Hc, int Wc, int C, void* stream) {
    GN_REQUIRE(dz && p && idx && coef3 && dy && dbias && ws && ctr, GOALNET_E_NULL, "bnpool_bwd_small: null pointer");
    GN_REQUIRE(N > 0 && Hc >= 3 && Wc >= 3 && small_ok(N, Hc, Wc, C), GOALNET_E_SHAPE, "bnpool_bwd_small: bad dims");
    GN_REQUIRE(aligned16(dz) && aligned16(p) && aligned16(dy) && aligned16(coef3) &&
               (reinterpret_cast<uintptr_t>(idx) & 3u) == 0 && (reinterpret_cast<uintptr_t>(ws) & 7u) == 0, GOALNET_E_ALIGN, "bnpool_bwd_small: alignment");
    GN_REQUIRE(ws_bytes >= goalnet_bn_small_ws_bytes(C), GOALNET_E_WORKSPACE, "bnpool_bwd_small: workspace too small");
    hipLaunchKernelGGL(bnpool_bwd_small_kernel, dim3(small_blocks((int64_t)N * Hc * Wc, C)), dim3(SMALL_T), SMALL_T * 4 * sizeof(double),
                       (hipStream_t)stream, dz, p, idx, coef3, dy, (double*)ws, ctr, dbias, N, Hc, Wc, C);
    GN_LAUNCH_CHECK("bnpool_bwd_small");
    return 0;
}

int goalnet_partials_sum(const double* partials, int nparts, int64_t stride, int C, float* out, void* stream) {
    GN_REQUIRE(partials && out, GOALNET_E_NULL, "partials_sum: null pointer");
    GN_REQUIRE(nparts > 0 && C > 0 && stride >= C, GOALNET_E_SHAPE, "partials_sum: bad dims");
    hipLaunchKernelGGL(partials_sum_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, partials, nparts, stride, C, out);
    GN_LAUNCH_CHECK("partials_sum");
    return 0;
}

int goalnet_partials_sum2(const double* pa, int na, int Ca, float* oa, const double* pb, int nb, int Cb, float* ob, void* stream) {
    GN_REQUIRE(pa && oa && pb && ob, GOALNET_E_NULL, "partials_sum2: null pointer");
    GN_REQUIRE(na > 0 && Ca > 0 && nb > 0 && Cb > 0, GOALNET_E_SHAPE, "partials_sum2: bad dims");
    const int ba = (Ca + 15) / 16, bb = (Cb + 15) / 16;
    hipLaunchKernelGGL(partials_sum2_kernel, dim3(ba + bb), dim3(256), 0, (hipStream_t)stream, pa, na, Ca, oa, ba, pb, nb, Cb, ob);
    GN_LAUNCH_CHECK("partials_sum2");
    return 0;
}

int goalnet_partials_sum_f64(const double* partials, int nparts, int64_t stride, int C, double* out, void* stream) {
    GN_REQUIRE(partials && out, GOALNET_E_NULL, "partials_sum_f64: null pointer");
    GN_REQUIRE(nparts > 0 && C > 0 && stride >= C, GOALNET_E_SHAPE, "partials_sum_f64: bad dims");
    hipLaunchKernelGGL(partials_sum_f64_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, partials, nparts, stride, C, out);
    GN_LAUNCH_CHECK("partials_sum_f64");
    return 0;
}

}  // extern "C"
