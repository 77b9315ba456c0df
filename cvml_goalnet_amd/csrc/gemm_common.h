// Shared pieces of the GEMM engines (gemm_f32.hip: fp32 MFMA, gemm_bf16.hip: bf16 MFMA): epilogue parameters,
// the accumulator -> memory epilogue for the 32x32 MFMA C/D layout (identical for every dtype on gfx950), the
// XCD-aware tile order and the split-K slab reduction.
#pragma once
#include <stdlib.h>

#include "common.h"

namespace goalnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int GEMM_BM = 128, GEMM_BN = 128;

// ------------------------------------------------------------------------------------------------
// epilogue
// ------------------------------------------------------------------------------------------------
enum { EPI_RAW = 0, EPI_BIAS_RELU = 1, EPI_MUL = 2, EPI_FULL = 3 };

struct EpiP {
    int mode;
    float* out; int64_t ld; int rows; int cols;
    const float* bias; int relu;
    const float* mul; int64_t ldmul;      // elementwise multiplier (dropout mask forward, saved mult backward)
    float* mult_out; int64_t ldmo;        // (pre-activation > 0) * mul, saved for backward
    int64_t slab_stride;                  // > 0: raw partial sums to out + split * slab_stride (ld = cols)
    void* out16;                          // non-null (256^2 bf16 kernel, raw epilogue only): the result as bf16 [rows][ld] instead of `out`
    // splitk_reduce only (fp32 conv weight gradient): out[row][col] -= corr_sh[col % corr_C] * corr_u[row * 9 + col / corr_C]
    const float* corr_u; const float* corr_sh; int corr_C;
    // fused split-K reduction (fp32 engine, few output tiles: the reference's 10-frame sub-batches): the raw partial sums go to
    // slabs + split * slab_stride (ld = cols), every block then takes a ticket on tile_ctr[tile] and the LAST block of a tile
    // sums the tile's slabs in split order (the order of splitk_reduce_kernel: same bits) and applies this epilogue (out, ld,
    // bias, relu, mul, mult_out) — no second launch. tile_ctr[] must be zero on entry and is zero again on exit.
    float* slabs; int* tile_ctr;
    // split operands in fp16 (csrc/split3.hip, "fp16x3"): the GEMM ran on operands scaled by powers of two 2^ka, 2^kb; *oscale = -(ka + kb)
    // (device memory, written by goalnet_split_scales): the accumulated value becomes ldexpf(v, *oscale) — one instruction, exact, and no
    // intermediate product of scales that could overflow — BEFORE bias / ReLU. Applied by the kernels that serve those modes ONLY
    // (gemm_bf16_256_kernel, the 128 x 64 tile's store, the split-K reduction kernels) — not by epi_apply: one more conditional load inside
    // the fp32 GEMM's unrolled store loop kept it from unrolling, its accumulators went to scratch (352 B per lane) and the fp32 conv
    // forward dropped from 0.85 to 0.54 of peak with 357 GB of "HBM" traffic per launch.
    const int* oscale = nullptr;
};

__device__ __forceinline__ float epi_apply(const EpiP& e, float v, int64_t row, int col) {
    if (e.bias) v += e.bias[col];
    float g = 1.f;
    if (e.relu) { g = v > 0.f ? 1.f : 0.f; v = v > 0.f ? v : 0.f; }
    if (e.mul) { const float m = e.mul[row * e.ldmul + col]; v *= m; g *= m; }
    if (e.mult_out) e.mult_out[row * e.ldmo + col] = g;
    return v;
}


// epilogue: D[row][col], col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// Each mode is a branch-free body under a wave-uniform switch: a branch around a load inside the
// unrolled store loop would serialise 64 dependent memory round trips.
// AIL / BIL: the A / B operand was read row-contiguous, so fragment f holds the strip's rows (cols) 2i + f.
template <bool AIL, bool BIL>
__device__ __forceinline__ void store_acc(const EpiP& ep, const f32x16 (&acc)[2][2], int tm, int tn, int split,
                                          int wm, int wn, int r, int h) {
    float* outp = ep.slab_stride > 0 ? (ep.slabs ? ep.slabs : ep.out) + (int64_t)split * ep.slab_stride : ep.out;
    const int mode = ep.slab_stride > 0 ? EPI_RAW : ep.mode;
    const int64_t ld = (ep.slab_stride > 0 && ep.slabs) ? (int64_t)ep.cols : ep.ld;
#pragma unroll
    for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int fn = 0; fn < 2; ++fn) {
            const int col = tn * GEMM_BN + wn * 64 + (BIL ? 2 * r + fn : fn * 32 + r);
            const bool colok = col < ep.cols;
            const int colc = colok ? col : 0;
            constexpr int RS = AIL ? 2 : 1;                                   // row step per MFMA row index
            const int64_t row0 = (int64_t)tm * GEMM_BM + wm * 64 + (AIL ? fm : fm * 32) + RS * 4 * h;
            if (mode == EPI_RAW) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = row0 + RS * ((e & 3) + 8 * (e >> 2));
                    if (colok && row < ep.rows) outp[row * ld + col] = acc[fm][fn][e];
                }
            } else if (mode == EPI_BIAS_RELU) {
                const float bv = ep.bias ? ep.bias[colc] : 0.f;
                const float lo = ep.relu ? 0.f : -INFINITY;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = row0 + RS * ((e & 3) + 8 * (e >> 2));
                    if (colok && row < ep.rows) outp[row * ld + col] = fmaxf(acc[fm][fn][e] + bv, lo);
                }
            } else if (mode == EPI_MUL) {
                float mv[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = row0 + RS * ((e & 3) + 8 * (e >> 2));
                    mv[e] = ep.mul[(row < ep.rows ? row : ep.rows - 1) * ep.ldmul + colc];
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = row0 + RS * ((e & 3) + 8 * (e >> 2));
                    if (colok && row < ep.rows) outp[row * ld + col] = acc[fm][fn][e] * mv[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = row0 + RS * ((e & 3) + 8 * (e >> 2));
                    if (colok && row < ep.rows) outp[row * ld + col] = epi_apply(ep, acc[fm][fn][e], row, col);
                }
            }
        }
}

// The fused split-K reduction (EpiP::tile_ctr): called by every block of a split-K GEMM after its raw slab store. Release
// (the block's slab stores reach device scope: L2 write-back on the multi-XCD part), ticket, and for the last block of the
// tile acquire + sum + epilogue. TN = tile width (128 or 64), tiles are 128 rows.
template <int TN>
__device__ __forceinline__ void fused_splitk_reduce(const EpiP& ep, int tm, int tn, int tile_id, int nsplit, int tid) {
    __shared__ int s_last;
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const int old = atomicAdd(&ep.tile_ctr[tile_id], 1);
        s_last = old == nsplit - 1;
        if (s_last) ep.tile_ctr[tile_id] = 0;              // nobody else touches this counter any more in this launch
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    constexpr int C4 = TN / 4, RP = 256 / C4;
    const int col = tn * TN + (tid % C4) * 4;
    if (col >= ep.cols) return;
    for (int rl = tid / C4; rl < GEMM_BM; rl += RP) {
        const int64_t row = (int64_t)tm * GEMM_BM + rl;
        if (row >= ep.rows) break;
        const float* s = ep.slabs + row * ep.cols + col;
        float4 a = *reinterpret_cast<const float4*>(s);
        int k = 1;
        for (; k + 3 < nsplit; k += 4) {                     // four loads in flight, added in split order
            const float4 b0 = *reinterpret_cast<const float4*>(s + (int64_t)k * ep.slab_stride);
            const float4 b1 = *reinterpret_cast<const float4*>(s + (int64_t)(k + 1) * ep.slab_stride);
            const float4 b2 = *reinterpret_cast<const float4*>(s + (int64_t)(k + 2) * ep.slab_stride);
            const float4 b3 = *reinterpret_cast<const float4*>(s + (int64_t)(k + 3) * ep.slab_stride);
            a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
            a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
            a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
            a.x += b3.x; a.y += b3.y; a.z += b3.z; a.w += b3.w;
        }
        for (; k < nsplit; ++k) {
            const float4 b = *reinterpret_cast<const float4*>(s + (int64_t)k * ep.slab_stride);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        float* o = ep.out + row * ep.ld + col;
        o[0] = epi_apply(ep, a.x, row, col);
        o[1] = epi_apply(ep, a.y, row, col + 1);
        o[2] = epi_apply(ep, a.z, row, col + 2);
        o[3] = epi_apply(ep, a.w, row, col + 3);
    }
}

// 4 x 4 transpose inside each quad of lanes: on entry register k of lane l holds element (row k, column l) of the quad's
// block, on exit register j of lane l holds (row l, column j). Two stages (swap bit 0, then bit 1, of register index against
// lane index). The cross-lane moves are volatile asm so that they run with all lanes active, ahead of any predicate: as
// builtins inside `?:` the compiler turned the selects into exec-masked regions and the moves read disabled lanes. The s_nop
// covers the gfx9 hazard "VALU write of a DPP source needs 2 wait states before the DPP reads it".
__device__ __forceinline__ void quad_transpose4(float& m0, float& m1, float& m2, float& m3, int lane) {
    const bool l0 = lane & 1, l1 = lane & 2;
    float x0, x1, x2, x3;
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %1, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %2, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %3, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                 : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(m0), "v"(m1), "v"(m2), "v"(m3));
    const float a0 = l0 ? x1 : m0, a1 = l0 ? m1 : x0, a2 = l0 ? x3 : m2, a3 = l0 ? m3 : x2;      // A[k](l1,l0) = M[2 k1 + l0](l1, k0)
    float y0, y1, y2, y3;
    asm volatile("s_nop 1\n\t"
                 "v_mov_b32_dpp %0, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %1, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %2, %6 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %3, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
                 : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    m0 = l1 ? y2 : a0; m1 = l1 ? y3 : a1; m2 = l1 ? a2 : y0; m3 = l1 ? a3 : y1;                    // N[j](l1,l0) = A[2 l1 + j0](j1, l0)
}

__device__ __forceinline__ void tile_of_block(int tiles_m, int tiles_n, int m_fast, int& tm, int& tn) {
    const unsigned v = xcd_remap(blockIdx.x, (unsigned)(tiles_m * tiles_n));
    if (m_fast) { tm = (int)(v % (unsigned)tiles_m); tn = (int)(v / (unsigned)tiles_m); }
    else        { tn = (int)(v % (unsigned)tiles_n); tm = (int)(v / (unsigned)tiles_n); }
}

// implemented in gemm_f32.hip
int launch_splitk_reduce(const char* name, const float* slabs, int nsplit, int64_t slab_stride, const EpiP& ep, hipStream_t st);
int pick_splits(int64_t tiles, int ktiles);

// Convolutions over few pixels (the reference's 10-frame sub-batches at 40x40: 10..40 output tiles) cannot fill 256 CUs
// with output tiles alone: split K into slabs (>= 4 K-tiles each, ~512 blocks), reduced in a fixed order.
inline int conv_fwd_splits(int64_t tiles, int ktiles) {
    if (tiles >= 256 || ktiles < 8) return 1;
    static const int target = getenv("GOALNET_SPLIT_TARGET") ? atoi(getenv("GOALNET_SPLIT_TARGET")) : 512;      // blocks to aim for (A/B runs)
    static const int minkt = getenv("GOALNET_SPLIT_MINKT") ? atoi(getenv("GOALNET_SPLIT_MINKT")) : 4;          // K-tiles per split, at least
    int64_t s = (target + tiles - 1) / tiles;
    const int64_t smax = ktiles / minkt;
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    const int kps = (int)((ktiles + s - 1) / s);
    return (ktiles + kps - 1) / kps;   // no empty splits
}

}  // namespace goalnet
