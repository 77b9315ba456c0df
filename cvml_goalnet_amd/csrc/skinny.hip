// Linear layers at the reference's operating point: M <= 16 rows (sub-batches of 10 frames, /root/reference/main.py:44,
// 177-184) against weights of up to 512 x 2.5 M (visbl.linear5, utils.py:168). With so few rows a 128 x 128 MFMA tile is
// > 87 % padding and the contraction is a pure weight stream: these kernels read (forward, dX) or write (dW) every
// weight exactly once with 16-byte lanes and keep the M activation rows in LDS / registers. HBM-bound by design:
// algorithmic bytes = 4 J K per call (85 MB for linear5 at 40x40, 5.1 GB at 224x224); fp32 FMA throughput needed is
// M/2 flop per weight byte, far below the VALU roof. All reductions are in a fixed order (deterministic).
//
// Called from goalnet_linear_fwd / _bwd_dx / _bwd_dw (gemm_f32.hip) when M <= SKINNY_MAX_M.
#include "gemm_common.h"
#include "skinny.h"

using namespace goalnet;

namespace {

constexpr int KT = 512;      // forward: K-slab per block (x slab in LDS: MR x KT floats = 24..32 KB -> 5..6 blocks per CU)
// forward: a block computes 4 waves x CW columns x RN rounds. Long reductions (linear5: 81 K-slabs): CW = 4, RN = 1 (16 columns:
// 2 592 blocks of ~100 VGPRs; RN = 2 measured 38 us against 35.5; round 3: four consecutive K-slabs per block with the next slab's
// weights prefetched and ONE set of wave reductions per block — 672 blocks — measured 69 us: at this size more blocks with one
// memory round trip each beat fewer blocks with several). The small layers (fusion MLP, AudBl linear: <= 2 K-slabs) had grids of 4 - 32
// blocks that way and ran ~10 us each on a sixteenth of the chip: CW = 1, RN = 1 (4 columns per block, 32 - 128 blocks).

__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// ---- forward: y[m][j] = epi(sum_k xa[m][k] w[j][k]); grid (ceil(J / 16), KS) --------------------------------------
template <int MR, int CW, int RN>
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int bnC, const float* __restrict__ w,
                                                        EpiP ep, int M, int64_t K, int J, int KS) {
    __shared__ float xs[MR * KT];
    constexpr int JB = 4 * CW * RN;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t k0 = (int64_t)blockIdx.y * KT;
    // all weight loads of the block (two rounds of 4 columns x KT) are issued first: they do not depend on x, and the HBM
    // round trip then overlaps the staging of the x slab instead of following it
    float4 w4[RN][KT / 256][CW];
#pragma unroll
    for (int rnd = 0; rnd < RN; ++rnd)
#pragma unroll
        for (int kk = 0; kk < KT / 256; ++kk)
#pragma unroll
            for (int jj = 0; jj < CW; ++jj) {
                const int j = blockIdx.x * JB + rnd * 4 * CW + wv * CW + jj;
                const int64_t k = k0 + kk * 256 + lane * 4;
                w4[rnd][kk][jj] = (j < J && k < K) ? *reinterpret_cast<const float4*>(w + (int64_t)j * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
    // stage the x slab (BatchNorm affine folded in, zero beyond M / K)
    for (int i = tid; i < MR * (KT / 4); i += 256) {
        const int m = i / (KT / 4), kq = i % (KT / 4);
        const int64_t k = k0 + kq * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < M && k < K) {
            v = *reinterpret_cast<const float4*>(x + (int64_t)m * ldx + k);
            if (scale) {
                const int c = (int)(k % bnC);
                const float4 s = *reinterpret_cast<const float4*>(scale + c), t = *reinterpret_cast<const float4*>(shift + c);
                v = make_float4(v.x * s.x + t.x, v.y * s.y + t.y, v.z * s.z + t.z, v.w * s.w + t.w);
            }
        }
        *reinterpret_cast<float4*>(&xs[m * KT + kq * 4]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int rnd = 0; rnd < RN; ++rnd) {
        const int j0 = blockIdx.x * JB + rnd * 4 * CW + wv * CW;
        float acc[CW][MR];
#pragma unroll
        for (int jj = 0; jj < CW; ++jj)
#pragma unroll
            for (int m = 0; m < MR; ++m) acc[jj][m] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KT / 256; ++kk) {
            const int kl = kk * 256 + lane * 4;
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const float4 x4 = *reinterpret_cast<const float4*>(&xs[m * KT + kl]);
#pragma unroll
                for (int jj = 0; jj < CW; ++jj) acc[jj][m] += dot4(w4[rnd][kk][jj], x4);
            }
        }
        // lane jj * MR + m keeps the total of (column j0 + jj, row m): one parallel epilogue instead of 4 * MR serial ones
        float mine = 0.f;
#pragma unroll
        for (int jj = 0; jj < CW; ++jj)
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const float v = wave_sum_dpp(acc[jj][m]);
                if (lane == jj * MR + m) mine = v;
            }
        const int m = lane % MR, j = j0 + lane / MR;
        if (lane < CW * MR && m < M && j < J) {
            if (KS > 1) ep.out[(int64_t)blockIdx.y * ep.slab_stride + (int64_t)m * J + j] = mine;
            else ep.out[(int64_t)m * ep.ld + j] = epi_apply(ep, mine, m, j);
        }
    }
}

// ---- dX: dx[m][k] = (sum_j dy[m][j] w[j][k]) * mult[m][k]; grid ceil(K / (4 KL)); (256 / KL) j-groups x KL k-float4 per block.
// KL = 16 for linear5 (64 k per block, 648 blocks; KL = 8 reads 128-byte pieces of the weight rows: 35 us against 24.5); KL = 4 for the small layers (16 k per block: 32 - 64 blocks instead of 8 - 16,
// and 64 j-groups: 8 instead of 32 dependent steps per thread) ----
template <int MR, int KL>
__global__ __launch_bounds__(256) void skinny_dx_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ w,
                                                       const float* __restrict__ mult, int64_t ldmult, float* __restrict__ dx,
                                                       int64_t lddx, int M, int64_t K, int J) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // max(JC * MR, 256 * MR * 4) floats
    constexpr int JC = 512;                                       // dy columns staged per round
    constexpr int JG = 256 / KL;
    const int tid = threadIdx.x, kl = tid % KL, jg = tid / KL;
    const int64_t k = (int64_t)blockIdx.x * (KL * 4) + kl * 4;
    const bool kok = k < K;
    float4 acc[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int jc = 0; jc < J; jc += JC) {
        const int jn = J - jc < JC ? J - jc : JC;
        __syncthreads();
        for (int i = tid; i < jn * MR; i += 256) {                // dys[j][m] (transposed: one b128 read gives 4 rows)
            const int j = i / MR, m = i % MR;
            sm[i] = m < M ? dy[(int64_t)m * lddy + jc + j] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int j = jg; j < jn; j += JG) {
            const float4 w4 = kok ? *reinterpret_cast<const float4*>(w + (int64_t)(jc + j) * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int mq = 0; mq < MR / 4; ++mq) {
                const float4 d = *reinterpret_cast<const float4*>(&sm[j * MR + mq * 4]);
                acc[mq * 4 + 0].x += d.x * w4.x; acc[mq * 4 + 0].y += d.x * w4.y; acc[mq * 4 + 0].z += d.x * w4.z; acc[mq * 4 + 0].w += d.x * w4.w;
                acc[mq * 4 + 1].x += d.y * w4.x; acc[mq * 4 + 1].y += d.y * w4.y; acc[mq * 4 + 1].z += d.y * w4.z; acc[mq * 4 + 1].w += d.y * w4.w;
                acc[mq * 4 + 2].x += d.z * w4.x; acc[mq * 4 + 2].y += d.z * w4.y; acc[mq * 4 + 2].z += d.z * w4.z; acc[mq * 4 + 2].w += d.z * w4.w;
                acc[mq * 4 + 3].x += d.w * w4.x; acc[mq * 4 + 3].y += d.w * w4.y; acc[mq * 4 + 3].z += d.w * w4.z; acc[mq * 4 + 3].w += d.w * w4.w;
            }
        }
    }
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(sm);                  // red[jg][m][kl]
#pragma unroll
    for (int m = 0; m < MR; ++m) red[(jg * MR + m) * KL + kl] = acc[m];
    __syncthreads();
    const int m = tid / KL;
    if (m < MR && m < M && kok) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 16
        for (int g = 0; g < JG; ++g) {
            const float4 v = red[(g * MR + m) * KL + kl];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (mult) {
            const float4 mv = *reinterpret_cast<const float4*>(mult + (int64_t)m * ldmult + k);
            s.x *= mv.x; s.y *= mv.y; s.z *= mv.z; s.w *= mv.w;
        }
        *reinterpret_cast<float4*>(dx + (int64_t)m * lddx + k) = s;
    }
}

// ---- dW: dw[j][k] = sum_m dy[m][j] xa[m][k]; grid (ceil(K / 1024), ceil(J / 32)); the thread keeps its M x-values -----
// JR = output rows (j) per block: 16 for linear5 (1 312 blocks; 32 rows / 656 blocks measured 27.7 us against 23.0), 4 for the
// small layers, whose grids were 4 - 16 blocks with 32
template <int MR, int JR>
__global__ __launch_bounds__(256) void skinny_dw_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x,
                                                       int64_t ldx, const float* __restrict__ scale, const float* __restrict__ shift,
                                                       int bnC, float* __restrict__ dw, float* __restrict__ db, int M, int64_t K,
                                                       int J) {
    __shared__ __attribute__((aligned(16))) float dys[JR * MR];
    const int tid = threadIdx.x;
    const int j0 = blockIdx.y * JR;
    for (int i = tid; i < JR * MR; i += 256) {
        const int j = i / MR, m = i % MR;
        dys[i] = (m < M && j0 + j < J) ? dy[(int64_t)m * lddy + j0 + j] : 0.f;
    }
    const int64_t k = (int64_t)blockIdx.x * 1024 + tid * 4;
    const bool kok = k < K;
    float4 x4[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < M && kok) {
            v = *reinterpret_cast<const float4*>(x + (int64_t)m * ldx + k);
            if (scale) {
                const int c = (int)(k % bnC);
                const float4 s = *reinterpret_cast<const float4*>(scale + c), t = *reinterpret_cast<const float4*>(shift + c);
                v = make_float4(v.x * s.x + t.x, v.y * s.y + t.y, v.z * s.z + t.z, v.w * s.w + t.w);
            }
        }
        x4[m] = v;
    }
    __syncthreads();
    const int jn = J - j0 < JR ? J - j0 : JR;
    if (db && blockIdx.x == 0 && tid < jn) {            // bias gradient = column sums of dy, rows added in index order
        float t = 0.f;
#pragma unroll
        for (int m = 0; m < MR; ++m) t += dys[tid * MR + m];
        db[j0 + tid] = t;
    }
    for (int j = 0; j < jn; ++j) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int mq = 0; mq < MR / 4; ++mq) {
            const float4 d = *reinterpret_cast<const float4*>(&dys[j * MR + mq * 4]);
            s.x += d.x * x4[mq * 4].x + d.y * x4[mq * 4 + 1].x + d.z * x4[mq * 4 + 2].x + d.w * x4[mq * 4 + 3].x;
            s.y += d.x * x4[mq * 4].y + d.y * x4[mq * 4 + 1].y + d.z * x4[mq * 4 + 2].y + d.w * x4[mq * 4 + 3].y;
            s.z += d.x * x4[mq * 4].z + d.y * x4[mq * 4 + 1].z + d.z * x4[mq * 4 + 2].z + d.w * x4[mq * 4 + 3].z;
            s.w += d.x * x4[mq * 4].w + d.y * x4[mq * 4 + 1].w + d.z * x4[mq * 4 + 2].w + d.w * x4[mq * 4 + 3].w;
        }
        if (kok) *reinterpret_cast<float4*>(dw + (int64_t)(j0 + j) * K + k) = s;
    }
}

int rows_class(int M) { return M <= 4 ? 4 : M <= 8 ? 8 : M <= 12 ? 12 : 16; }

#define SKINNY_DISPATCH(M, ...)                  \
    switch (rows_class(M)) {                     \
        case 4: { constexpr int MR = 4; __VA_ARGS__; } break;   \
        case 8: { constexpr int MR = 8; __VA_ARGS__; } break;   \
        case 12: { constexpr int MR = 12; __VA_ARGS__; } break; \
        default: { constexpr int MR = 16; __VA_ARGS__; } break; \
    }

}  // namespace

namespace goalnet {

int skinny_fwd_splits(int64_t K) { return (int)((K + KT - 1) / KT); }

size_t skinny_fwd_ws_bytes(int M, int64_t K, int J) {
    const int s = skinny_fwd_splits(K);
    return s > 1 ? (size_t)s * (size_t)M * (size_t)J * sizeof(float) : 0;
}

int skinny_linear_fwd(const float* x, int64_t ldx, const float* scale, const float* shift, int bnC, const float* w,
                      const EpiP& efinal, int M, int64_t K, int J, void* ws, hipStream_t st) {
    const int KS = skinny_fwd_splits(K);
    EpiP ep = efinal;
    if (KS > 1) ep = EpiP{EPI_RAW, (float*)ws, J, M, J, nullptr, 0, nullptr, 0, nullptr, 0, (int64_t)M * J};
    if (KS >= 8) {
        const dim3 grid((unsigned)((J + 15) / 16), (unsigned)KS);
        SKINNY_DISPATCH(M, hipLaunchKernelGGL((skinny_fwd_kernel<MR, 4, 1>), grid, dim3(256), 0, st, x, ldx, scale, shift, bnC, w, ep, M, K, J, KS));
    } else {
        const dim3 grid((unsigned)((J + 3) / 4), (unsigned)KS);
        SKINNY_DISPATCH(M, hipLaunchKernelGGL((skinny_fwd_kernel<MR, 1, 1>), grid, dim3(256), 0, st, x, ldx, scale, shift, bnC, w, ep, M, K, J, KS));
    }
    GN_LAUNCH_CHECK("linear_fwd(skinny)");
    if (KS == 1) return 0;
    return launch_splitk_reduce("linear_fwd(skinny).reduce", (const float*)ws, KS, (int64_t)M * J, efinal, st);
}

int skinny_linear_dx(const float* dy, int64_t lddy, const float* w, const float* mult, int64_t ldmult, float* dx, int64_t lddx,
                     int M, int64_t K, int J, hipStream_t st) {
    if (K > 4096) {
        const dim3 grid((unsigned)((K + 63) / 64));
        SKINNY_DISPATCH(M, {
            const size_t a = (size_t)512 * MR * sizeof(float), b = (size_t)256 * MR * sizeof(float4);
            hipLaunchKernelGGL((skinny_dx_kernel<MR, 16>), grid, dim3(256), a > b ? a : b, st, dy, lddy, w, mult, ldmult, dx, lddx, M, K, J);
        });
    } else {
        const dim3 grid((unsigned)((K + 15) / 16));
        SKINNY_DISPATCH(M, {
            const size_t a = (size_t)512 * MR * sizeof(float), b = (size_t)256 * MR * sizeof(float4);
            hipLaunchKernelGGL((skinny_dx_kernel<MR, 4>), grid, dim3(256), a > b ? a : b, st, dy, lddy, w, mult, ldmult, dx, lddx, M, K, J);
        });
    }
    GN_LAUNCH_CHECK("linear_bwd_dx(skinny)");
    return 0;
}

int skinny_linear_dw(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* scale, const float* shift, int bnC,
                     float* dw, float* db, int M, int64_t K, int J, hipStream_t st) {
    if (K > 4096) {
        const dim3 grid((unsigned)((K + 1023) / 1024), (unsigned)((J + 15) / 16));
        SKINNY_DISPATCH(M, hipLaunchKernelGGL((skinny_dw_kernel<MR, 16>), grid, dim3(256), 0, st, dy, lddy, x, ldx, scale, shift, bnC, dw, db, M, K, J));
    } else {
        const dim3 grid((unsigned)((K + 1023) / 1024), (unsigned)((J + 3) / 4));
        SKINNY_DISPATCH(M, hipLaunchKernelGGL((skinny_dw_kernel<MR, 4>), grid, dim3(256), 0, st, dy, lddy, x, ldx, scale, shift, bnC, dw, db, M, K, J));
    }
    GN_LAUNCH_CHECK("linear_bwd_dw(skinny)");
    return 0;
}

}  // namespace goalnet
