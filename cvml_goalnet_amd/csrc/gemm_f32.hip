// fp32 GEMM engine on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-exact
// fmaf chain, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak) and the hot-path ops built on it:
//
//   conv 3x3 s1 p1 forward / data-gradient  (implicit GEMM, im2col on the A load, BatchNorm folded into the load)
//   conv 3x3 weight-gradient                (implicit GEMM reduced over N*H*W, deterministic split + slab sum)
//   linear forward / dX / dW                (linear5 with BatchNorm folded into the load and split-K)
//
// Replaces ATen's CPU conv2d / linear (oneDNN, MKL) under /root/reference/utils.py:156-170, 243-253 and
// their autograd counterparts (main.py:192).
//
// Geometry: 256 threads = 4 waves (one per SIMD), block tile 128 x 128 x 32, wave tile 64 x 64 =
// 2 x 2 MFMA tiles of 32 x 32 (64 accumulator VGPRs). Operands are staged global -> registers -> LDS
// (the A loader applies an affine and zero padding on the way, so LDS-DMA is not usable), double-buffered
// in LDS with the next K-tile's global loads in flight under the current tile's 64 MFMAs per wave.
// An fp32 MFMA occupies its SIMD for 64 cycles, so the 4096 MFMA cycles per K-tile per wave cover the
// 8 global loads + 8 LDS writes + 16-32 LDS reads a thread issues per K-tile with room to spare.
//
// The K order inside a K-tile is permuted identically for A and B (lane half h of MFMA step j of group s
// holds k = 8s + 4h + j) so that a K-contiguous operand is read from LDS with one ds_read_b128 per four
// MFMA steps; rows are padded to 36 floats, which makes those reads bank-conflict free.
#include <stdlib.h>

#include "common.h"

using namespace goalnet;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDK = BK + 4;           // K-contiguous operand: LDS row stride in floats (pad = one b128)
constexpr int LDR = 128;              // row-contiguous operand: LDS stride between k rows
constexpr int OP_FLOATS = BM * LDK;   // 4608 floats (>= BK * LDR = 4096)

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4fma(float4 v, float4 s, float4 t) {
    return make_float4(fmaf(v.x, s.x, t.x), fmaf(v.y, s.y, t.y), fmaf(v.z, s.z, t.z), fmaf(v.w, s.w, t.w));
}

// ------------------------------------------------------------------------------------------------
// operand loaders: global -> registers (4 x float4 per thread per K-tile)
// ------------------------------------------------------------------------------------------------

// Every loader is split in two so that the global loads stay in flight under the MFMAs of the current
// K-tile: issue() only issues unconditional loads (out-of-range rows / padding taps read a clamped, valid
// address), finish() applies the zero fill and the affine right before the registers go to LDS. A branch
// around a load would make hipcc wait for it at once (cdna_hip_programming.md §5, trap (c)).
__device__ __forceinline__ float4 f4sel(bool v, float4 a) { return v ? a : f4zero(); }

// K-contiguous matrix X[rows][K] (leading dim ld). Optional affine x*scale[k % C] + shift[k % C].
template <bool AFFINE>
struct KCLoader {
    struct P {
        const float* x; int64_t ld; int rows;
        const float* scale; const float* shift; int bnC;
    };
    static constexpr bool KC = true;
    const float* ptr[4];
    unsigned okmask;
    const float* scale; const float* shift;
    int bnC, c4;
    float4 sc, sh;
    __device__ KCLoader(const P& p, int row0, int tid) {
        c4 = (tid & 7) * 4;
        scale = p.scale; shift = p.shift; bnC = p.bnC;
        const int rr = tid >> 3;
        okmask = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + rr + 32 * i;
            const bool ok = row < p.rows;
            okmask |= (ok ? 1u : 0u) << i;
            ptr[i] = p.x + (int64_t)(ok ? row : 0) * p.ld + c4;
        }
    }
    __device__ __forceinline__ void issue(int kt, float4 (&r)[4]) {
        const int k = kt * BK;
        if (AFFINE) {
            const int ch = (k + c4) % bnC;
            sc = *reinterpret_cast<const float4*>(scale + ch);
            sh = *reinterpret_cast<const float4*>(shift + ch);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = *reinterpret_cast<const float4*>(ptr[i] + k);
    }
    __device__ __forceinline__ void finish(float4 (&r)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = f4sel((okmask >> i) & 1u, AFFINE ? f4fma(r[i], sc, sh) : r[i]);
    }
};

// Row-contiguous matrix X[kred][cols] (leading dim ld): for a fixed reduction index the tile's columns
// are contiguous. Optional affine per column (channel = col % C), fixed per thread.
template <bool AFFINE>
struct MCLoader {
    struct P {
        const float* x; int64_t ld; int cols; int kred;
        const float* scale; const float* shift; int bnC;
    };
    static constexpr bool KC = false;
    const float* base;
    int64_t ld;
    int kred, k0;
    bool colok;
    unsigned vmask;
    float4 sc, sh;
    __device__ MCLoader(const P& p, int col0, int tid) {
        const int col = col0 + (tid & 31) * 4;
        colok = col < p.cols;
        base = p.x + (colok ? col : 0);
        ld = p.ld; kred = p.kred; k0 = tid >> 5;
        if (AFFINE) {
            const int ch = (colok ? col : 0) % p.bnC;
            sc = *reinterpret_cast<const float4*>(p.scale + ch);
            sh = *reinterpret_cast<const float4*>(p.shift + ch);
        }
    }
    __device__ __forceinline__ void issue(int kt, float4 (&r)[4]) {
        vmask = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = kt * BK + k0 + 8 * i;
            const bool v = colok && kk < kred;
            vmask |= (v ? 1u : 0u) << i;
            r[i] = *reinterpret_cast<const float4*>(base + (int64_t)(kk < kred ? kk : kred - 1) * ld);
        }
    }
    __device__ __forceinline__ void finish(float4 (&r)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = f4sel((vmask >> i) & 1u, AFFINE ? f4fma(r[i], sc, sh) : r[i]);
    }
};

// im2col of an NHWC tensor for a 3x3 / stride 1 / pad 1 convolution: row m = (n,h,w), k = (kh,kw,ci).
// The affine (BatchNorm of the producing block) is applied to in-bounds elements only: the reference pads
// the BatchNorm OUTPUT with zeros (/root/reference/utils.py:154-156).
template <bool AFFINE>
struct ConvALoader {
    struct P {
        const float* x; int H, W, C; int64_t M;
        const float* scale; const float* shift;
    };
    static constexpr bool KC = true;
    const float* ptr[4];
    unsigned mask[4];
    unsigned vmask;
    const float* scale; const float* shift;
    int W, C, c4;
    float4 sc, sh;
    __device__ ConvALoader(const P& p, int row0, int tid) {
        c4 = (tid & 7) * 4;
        W = p.W; C = p.C; scale = p.scale; shift = p.shift;
        const int rr = tid >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = (int64_t)row0 + rr + 32 * i;
            unsigned mk = 0;
            int64_t mm = 0;
            if (m < p.M) {
                mm = m;
                const int w = (int)((unsigned)m % (unsigned)p.W);
                const int h = (int)(((unsigned)m / (unsigned)p.W) % (unsigned)p.H);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const bool v = (unsigned)(h + kh - 1) < (unsigned)p.H && (unsigned)(w + kw - 1) < (unsigned)p.W;
                        mk |= (v ? 1u : 0u) << (kh * 3 + kw);
                    }
            }
            mask[i] = mk;
            ptr[i] = p.x + mm * p.C + c4;
        }
    }
    __device__ __forceinline__ void issue(int kt, float4 (&r)[4]) {
        const int k = kt * BK;
        const int tap = k / C;
        const int ci = k - tap * C;
        const int kh = tap / 3, kw = tap - 3 * kh;
        const int64_t off = (int64_t)((kh - 1) * W + (kw - 1)) * C;
        if (AFFINE) {
            sc = *reinterpret_cast<const float4*>(scale + ci + c4);
            sh = *reinterpret_cast<const float4*>(shift + ci + c4);
        }
        vmask = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool v = (mask[i] >> tap) & 1u;
            vmask |= (v ? 1u : 0u) << i;
            r[i] = *reinterpret_cast<const float4*>(ptr[i] + ci + (v ? off : 0));   // centre tap is always in bounds
        }
    }
    __device__ __forceinline__ void finish(float4 (&r)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = f4sel((vmask >> i) & 1u, AFFINE ? f4fma(r[i], sc, sh) : r[i]);
    }
};

// B operand of the weight gradient: B(col = (tap, ci), red = m) = bnapply(x)[pixel m shifted by tap][ci].
template <bool AFFINE>
struct ConvWgradBLoader {
    struct P {
        const float* x; int H, W, C; int M;
        const float* scale; const float* shift;
    };
    static constexpr bool KC = false;
    const float* base;
    int H, W, C, M, k0, dh, dw, tapoff;
    bool colok;
    unsigned vmask;
    float4 sc, sh;
    __device__ ConvWgradBLoader(const P& p, int col0, int tid) {
        const int col = col0 + (tid & 31) * 4;
        colok = col < 9 * p.C;
        const int cc = colok ? col : 0;
        const int tap = cc / p.C, ci = cc - tap * p.C;
        const int kh = tap / 3, kw = tap - 3 * kh;
        dh = kh - 1; dw = kw - 1;
        H = p.H; W = p.W; C = p.C; M = p.M; k0 = tid >> 5;
        tapoff = (dh * p.W + dw) * p.C;
        base = p.x + ci;
        if (AFFINE) {
            sc = *reinterpret_cast<const float4*>(p.scale + ci);
            sh = *reinterpret_cast<const float4*>(p.shift + ci);
        }
    }
    __device__ __forceinline__ void issue(int kt, float4 (&r)[4]) {
        vmask = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m = kt * BK + k0 + 8 * i;
            bool v = colok && m < M;
            m = m < M ? m : M - 1;
            const unsigned t = (unsigned)m / (unsigned)W;
            const int w = m - (int)t * W;
            const int h = (int)(t % (unsigned)H);
            v = v && (unsigned)(h + dh) < (unsigned)H && (unsigned)(w + dw) < (unsigned)W;
            vmask |= (v ? 1u : 0u) << i;
            r[i] = *reinterpret_cast<const float4*>(base + (int64_t)m * C + (v ? tapoff : 0));
        }
    }
    __device__ __forceinline__ void finish(float4 (&r)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = f4sel((vmask >> i) & 1u, AFFINE ? f4fma(r[i], sc, sh) : r[i]);
    }
};

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
};

__device__ __forceinline__ float epi_apply(const EpiP& e, float v, int64_t row, int col) {
    if (e.bias) v += e.bias[col];
    float g = 1.f;
    if (e.relu) { g = v > 0.f ? 1.f : 0.f; v = v > 0.f ? v : 0.f; }
    if (e.mul) { const float m = e.mul[row * e.ldmul + col]; v *= m; g *= m; }
    if (e.mult_out) e.mult_out[row * e.ldmo + col] = g;
    return v;
}

// ------------------------------------------------------------------------------------------------
// LDS staging and the MFMA tile
// ------------------------------------------------------------------------------------------------
template <bool KC>
__device__ __forceinline__ void store_tile(float* l, const float4 (&r)[4], int tid) {
    if (KC) {
        const int c4 = (tid & 7) * 4, row = tid >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&l[(row + 32 * i) * LDK + c4]) = r[i];
    } else {
        const int r4 = (tid & 31) * 4, k = tid >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&l[(k + 8 * i) * LDR + r4]) = r[i];
    }
}

template <bool KC>
__device__ __forceinline__ void read_frag(const float* l, int rowbase, int s, int r, int h, float (&f)[4]) {
    if (KC) {
        const float4 t = *reinterpret_cast<const float4*>(&l[(rowbase + r) * LDK + 8 * s + 4 * h]);
        f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = l[(8 * s + 4 * h + j) * LDR + rowbase + r];
    }
}

template <bool AKC, bool BKC>
__device__ __forceinline__ void read_group(const float* la, const float* lb, int wm, int wn, int s, int r, int h,
                                           float (&a)[2][4], float (&b)[2][4]) {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        read_frag<AKC>(la, wm * 64 + f * 32, s, r, h, a[f]);
        read_frag<BKC>(lb, wn * 64 + f * 32, s, r, h, b[f]);
    }
}

__device__ __forceinline__ void mfma_group(const float (&a)[2][4], const float (&b)[2][4], f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int fm = 0; fm < 2; ++fm)
#pragma unroll
            for (int fn = 0; fn < 2; ++fn)
                acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[fm][j], b[fn][j], acc[fm][fn], 0, 0, 0);
    }
}

// One K-tile = 4 groups of 16 MFMAs. VAR 0: fragments are read right before their group (hipcc sinks the reads to
// the end of the previous group, exposing ~100 cycles of LDS latency per group). VAR 1: software-pipelined — the
// fragments of group s+1 are in flight while group s runs on the matrix pipe (two fragment register sets).
template <bool AKC, bool BKC, int VAR>
__device__ __forceinline__ void compute_tile(const float* la, const float* lb, f32x16 (&acc)[2][2],
                                             int wm, int wn, int r, int h) {
    if (VAR == 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float a[2][4], b[2][4];
            read_group<AKC, BKC>(la, lb, wm, wn, s, r, h, a, b);
            mfma_group(a, b, acc);
        }
    } else {
        float a0[2][4], b0[2][4], a1[2][4], b1[2][4];
        read_group<AKC, BKC>(la, lb, wm, wn, 0, r, h, a0, b0);
        read_group<AKC, BKC>(la, lb, wm, wn, 1, r, h, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(a0, b0, acc);
        __builtin_amdgcn_sched_barrier(0);
        read_group<AKC, BKC>(la, lb, wm, wn, 2, r, h, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(a1, b1, acc);
        __builtin_amdgcn_sched_barrier(0);
        read_group<AKC, BKC>(la, lb, wm, wn, 3, r, h, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group(a0, b0, acc);
        mfma_group(a1, b1, acc);
    }
}

template <class AL, class BL, int VAR>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(typename AL::P ap, typename BL::P bp, EpiP ep,
                                                          int tiles_m, int tiles_n, int m_fast,
                                                          int ktiles, int ktiles_per_split) {
    __shared__ __attribute__((aligned(16))) float lds[VAR == 8 ? 4 : 2][2][OP_FLOATS];   // VAR 8: 1 block/CU experiment
    const int tid = threadIdx.x;
    if (VAR == 8 && ktiles < 0) lds[3][1][tid] = 0.f;
    const unsigned v = xcd_remap(blockIdx.x, (unsigned)(tiles_m * tiles_n));
    int tm, tn;
    if (m_fast) { tm = (int)(v % (unsigned)tiles_m); tn = (int)(v / (unsigned)tiles_m); }
    else        { tn = (int)(v % (unsigned)tiles_n); tm = (int)(v / (unsigned)tiles_n); }
    const int split = blockIdx.y;
    const int kt0 = split * ktiles_per_split;
    const int kt1 = min(ktiles, kt0 + ktiles_per_split);

    AL al(ap, VAR == 2 ? 0 : tm * BM, tid);      // VAR 2: timing experiment, every block reads M-tile 0 (L2-resident)
    BL bl(bp, tn * BN, tid);

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[4], rb[4];
    if (kt0 < kt1) {
        al.issue(kt0, ra);
        bl.issue(kt0, rb);
        al.finish(ra);
        bl.finish(rb);
        store_tile<AL::KC>(lds[0][0], ra, tid);
        store_tile<BL::KC>(lds[0][1], rb, tid);
    }
    __syncthreads();
    if (VAR == 7) {
        // One basic block per K-tile (staging is unconditional: the last iteration re-stages the final tile into the
        // spare buffer), so that the scheduler can follow the interleave below: every fp32 MFMA occupies the matrix
        // pipe for 64 cycles, during which the SAME wave can issue the address arithmetic, global loads, LDS reads,
        // affine and LDS writes of the staging — instead of running them as a separate phase in front of the MFMAs.
        for (int kt = kt0; kt < kt1; ++kt) {
            const int cur = (kt - kt0) & 1;
            const int nxt = kt + 1 < kt1 ? kt + 1 : kt;
            al.issue(nxt, ra);
            bl.issue(nxt, rb);
            compute_tile<AL::KC, BL::KC, 0>(lds[cur][0], lds[cur][1], acc, wm, wn, r, h);
            al.finish(ra);
            bl.finish(rb);
            store_tile<AL::KC>(lds[cur ^ 1][0], ra, tid);
            store_tile<BL::KC>(lds[cur ^ 1][1], rb, tid);
            // ---- interleave: masks 0x8 MFMA, 0x2 VALU, 0x20 VMEM read, 0x100 DS read, 0x200 DS write
            __builtin_amdgcn_sched_group_barrier(0x100, AL::KC && BL::KC ? 4 : 8, 0);
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < 40) {
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                    if (i % 4 == 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    if (i % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                if (i % 16 == 8 && i < 48) __builtin_amdgcn_sched_group_barrier(0x100, AL::KC && BL::KC ? 4 : 8, 0);
            }
            __syncthreads();
        }
    } else
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        const bool more = (VAR == 3 || VAR == 4) ? false : kt + 1 < kt1;   // VAR 3/4: timing experiments (no staging)
        if (more) {
            if (VAR != 9) al.issue(kt + 1, ra);       // VAR 9: B loads only; VAR 10: A loads only (timing experiments)
            if (VAR != 10) bl.issue(kt + 1, rb);
        }
        compute_tile<AL::KC, BL::KC, (VAR == 1 ? 1 : 0)>(lds[cur][0], lds[cur][1], acc, wm, wn, r, h);
        if (more) {
            if (VAR == 5 || VAR == 9 || VAR == 10) {          // timing experiment: loads only, no finish / LDS write
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    asm volatile("" ::"v"(ra[i].x), "v"(ra[i].y), "v"(ra[i].z), "v"(ra[i].w));
                    asm volatile("" ::"v"(rb[i].x), "v"(rb[i].y), "v"(rb[i].z), "v"(rb[i].w));
                }
            } else if (VAR == 6) {   // timing experiment: loads + LDS write, no finish (no affine / selects)
                store_tile<AL::KC>(lds[cur ^ 1][0], ra, tid);
                store_tile<BL::KC>(lds[cur ^ 1][1], rb, tid);
            } else {
                al.finish(ra);
                bl.finish(rb);
                store_tile<AL::KC>(lds[cur ^ 1][0], ra, tid);
                store_tile<BL::KC>(lds[cur ^ 1][1], rb, tid);
            }
        }
        if (VAR != 4) __syncthreads();
    }

    // epilogue: D[row][col], col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    // Each mode is a branch-free body under a wave-uniform switch: a branch around a load inside the
    // unrolled store loop would serialise 64 dependent memory round trips.
    float* outp = ep.out + (ep.slab_stride > 0 ? (int64_t)split * ep.slab_stride : 0);
    const int mode = ep.slab_stride > 0 ? EPI_RAW : ep.mode;
#pragma unroll
    for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int fn = 0; fn < 2; ++fn) {
            const int col = tn * BN + wn * 64 + fn * 32 + r;
            const bool colok = col < ep.cols;
            const int colc = colok ? col : 0;
            const int64_t row0 = (int64_t)tm * BM + wm * 64 + fm * 32 + 4 * h;
            if (mode == EPI_RAW) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
                    if (colok && row < ep.rows) outp[row * ep.ld + col] = acc[fm][fn][e];
                }
            } else if (mode == EPI_BIAS_RELU) {
                const float bv = ep.bias ? ep.bias[colc] : 0.f;
                const float lo = ep.relu ? 0.f : -INFINITY;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
                    if (colok && row < ep.rows) outp[row * ep.ld + col] = fmaxf(acc[fm][fn][e] + bv, lo);
                }
            } else if (mode == EPI_MUL) {
                float mv[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
                    mv[e] = ep.mul[(row < ep.rows ? row : ep.rows - 1) * ep.ldmul + colc];
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
                    if (colok && row < ep.rows) outp[row * ep.ld + col] = acc[fm][fn][e] * mv[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = row0 + (e & 3) + 8 * (e >> 2);
                    if (colok && row < ep.rows) outp[row * ep.ld + col] = epi_apply(ep, acc[fm][fn][e], row, col);
                }
            }
        }
}

// sums `nsplit` raw slabs (deterministic order) and applies the epilogue. cols % 4 == 0.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* slabs, int nsplit, int64_t slab_stride, EpiP ep) {
    const int c4n = ep.cols >> 2;
    const int64_t total = (int64_t)ep.rows * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / c4n;
        const int col = (int)(i - row * c4n) * 4;
        const float* s = slabs + row * ep.cols + col;
        float4 a = *reinterpret_cast<const float4*>(s);
        for (int k = 1; k < nsplit; ++k) {
            const float4 b = *reinterpret_cast<const float4*>(s + (int64_t)k * slab_stride);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        float* o = ep.out + row * ep.ld + col;
        o[0] = epi_apply(ep, a.x, row, col);
        o[1] = epi_apply(ep, a.y, row, col + 1);
        o[2] = epi_apply(ep, a.z, row, col + 2);
        o[3] = epi_apply(ep, a.w, row, col + 3);
    }
}

template <class AL, class BL>
int launch_gemm(const char* name, const typename AL::P& ap, const typename BL::P& bp, const EpiP& ep,
                int64_t M, int64_t N, int ktiles, int nsplit, int m_fast, hipStream_t st) {
    const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    GN_REQUIRE(tiles_m * tiles_n < (1ll << 31), GOALNET_E_SHAPE, "%s: too many tiles", name);
    GN_REQUIRE(nsplit >= 1 && nsplit <= 65535, GOALNET_E_SHAPE, "%s: bad split count %d", name, nsplit);
    const int kps = (ktiles + nsplit - 1) / nsplit;
    dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)nsplit, 1);
    static const int variant = getenv("GOALNET_GEMM_VARIANT") ? atoi(getenv("GOALNET_GEMM_VARIANT")) : 0;
    if (variant == 9)
        hipLaunchKernelGGL((gemm_f32_kernel<AL, BL, 9>), grid, dim3(256), 0, st, ap, bp, ep, (int)tiles_m, (int)tiles_n,
                           m_fast, ktiles, kps);
    else if (variant == 10)
        hipLaunchKernelGGL((gemm_f32_kernel<AL, BL, 10>), grid, dim3(256), 0, st, ap, bp, ep, (int)tiles_m, (int)tiles_n,
                           m_fast, ktiles, kps);
    else if (variant == 8)
        hipLaunchKernelGGL((gemm_f32_kernel<AL, BL, 8>), grid, dim3(256), 0, st, ap, bp, ep, (int)tiles_m, (int)tiles_n,
                           m_fast, ktiles, kps);
    else if (variant == 7)
        hipLaunchKernelGGL((gemm_f32_kernel<AL, BL, 7>), grid, dim3(256), 0, st, ap, bp, ep, (int)tiles_m, (int)tiles_n,
                           m_fast, ktiles, kps);
    else if (variant == 5)
        hipLaunchKernelGGL((gemm_f32_kernel<AL, BL, 5>), grid, dim3(256), 0, st, ap, bp, ep, (int)tiles_m, (int)tiles_n,
                           m_fast, ktiles, kps);
    else if (variant == 6)
        hipLaunchKernelGGL((gemm_f32_kernel<AL, BL, 6>), grid, dim3(256), 0, st, ap, bp, ep, (int)tiles_m, (int)tiles_n,
                           m_fast, ktiles, kps);
    else if (variant == 3)
        hipLaunchKernelGGL((gemm_f32_kernel<AL, BL, 3>), grid, dim3(256), 0, st, ap, bp, ep, (int)tiles_m, (int)tiles_n,
                           m_fast, ktiles, kps);
    else if (variant == 4)
        hipLaunchKernelGGL((gemm_f32_kernel<AL, BL, 4>), grid, dim3(256), 0, st, ap, bp, ep, (int)tiles_m, (int)tiles_n,
                           m_fast, ktiles, kps);
    else if (variant == 2)
        hipLaunchKernelGGL((gemm_f32_kernel<AL, BL, 2>), grid, dim3(256), 0, st, ap, bp, ep, (int)tiles_m, (int)tiles_n,
                           m_fast, ktiles, kps);
    else if (variant == 1)
        hipLaunchKernelGGL((gemm_f32_kernel<AL, BL, 1>), grid, dim3(256), 0, st, ap, bp, ep, (int)tiles_m, (int)tiles_n,
                           m_fast, ktiles, kps);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<AL, BL, 0>), grid, dim3(256), 0, st, ap, bp, ep, (int)tiles_m, (int)tiles_n,
                           m_fast, ktiles, kps);
    GN_LAUNCH_CHECK(name);
    return 0;
}

int launch_reduce(const char* name, const float* slabs, int nsplit, int64_t slab_stride, const EpiP& ep, hipStream_t st) {
    const int64_t total = (int64_t)ep.rows * (ep.cols >> 2);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, slabs, nsplit, slab_stride, ep);
    GN_LAUNCH_CHECK(name);
    return 0;
}

// split count so that the grid has about 1024 blocks but every split keeps >= 8 K-tiles
int pick_splits(int64_t tiles, int ktiles) {
    if (ktiles < 64) return 1;
    int64_t s = (1024 + tiles - 1) / tiles;
    const int64_t smax = ktiles / 8 > 1 ? ktiles / 8 : 1;
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    const int kps = (int)((ktiles + s - 1) / s);
    return (ktiles + kps - 1) / kps;   // no empty splits
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int goalnet_conv3x3_fwd(const float* x, const float* scale, const float* shift, const float* w,
                        const float* bias, int relu, float* y, int N, int H, int W, int Cin, int Cout, void* stream) {
    GN_REQUIRE(x && w && y, GOALNET_E_NULL, "conv3x3_fwd: null pointer");
    GN_REQUIRE((scale == nullptr) == (shift == nullptr), GOALNET_E_NULL, "conv3x3_fwd: scale/shift must both be set or both NULL");
    GN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, GOALNET_E_SHAPE, "conv3x3_fwd: non-positive dim");
    GN_REQUIRE(Cin % BK == 0, GOALNET_E_SHAPE, "conv3x3_fwd: Cin=%d must be a multiple of %d", Cin, BK);
    GN_REQUIRE(Cout % 4 == 0, GOALNET_E_SHAPE, "conv3x3_fwd: Cout=%d must be a multiple of 4", Cout);
    GN_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y) && aligned16(scale) && aligned16(shift),
               GOALNET_E_ALIGN, "conv3x3_fwd: pointers must be 16-byte aligned");
    const int64_t M = (int64_t)N * H * W;
    GN_REQUIRE(M < (1ll << 31) - 256, GOALNET_E_SHAPE, "conv3x3_fwd: N*H*W too large");
    hipStream_t st = (hipStream_t)stream;
    KCLoader<false>::P bp{w, (int64_t)9 * Cin, Cout, nullptr, nullptr, 1};
    EpiP ep{EPI_BIAS_RELU, y, Cout, (int)M, Cout, bias, relu, nullptr, 0, nullptr, 0, 0};
    const int ktiles = 9 * Cin / BK;
    if (scale) {
        ConvALoader<true>::P ap{x, H, W, Cin, M, scale, shift};
        return launch_gemm<ConvALoader<true>, KCLoader<false>>("conv3x3_fwd", ap, bp, ep, M, Cout, ktiles, 1, 0, st);
    }
    ConvALoader<false>::P ap{x, H, W, Cin, M, nullptr, nullptr};
    return launch_gemm<ConvALoader<false>, KCLoader<false>>("conv3x3_fwd", ap, bp, ep, M, Cout, ktiles, 1, 0, st);
}

static int wgrad_splits(int64_t M, int Cin, int Cout) {
    const int64_t tiles = (int64_t)((Cout + BM - 1) / BM) * ((9 * Cin + BN - 1) / BN);
    const int ktiles = (int)((M + BK - 1) / BK);
    int64_t s = (2048 + tiles - 1) / tiles;
    const int64_t smax = ktiles / 16 > 1 ? ktiles / 16 : 1;
    if (s > smax) s = smax;
    if (s > 512) s = 512;
    const int kps = (int)((ktiles + s - 1) / s);
    return (ktiles + kps - 1) / kps;
}

size_t goalnet_conv3x3_wgrad_ws_bytes(int N, int H, int W, int Cin, int Cout) {
    const int64_t M = (int64_t)N * H * W;
    return (size_t)wgrad_splits(M, Cin, Cout) * (size_t)Cout * 9 * Cin * sizeof(float);
}

int goalnet_conv3x3_wgrad(const float* x, const float* scale, const float* shift, const float* dy, float* dw,
                          void* ws, size_t ws_bytes, int N, int H, int W, int Cin, int Cout, void* stream) {
    GN_REQUIRE(x && dy && dw && ws, GOALNET_E_NULL, "conv3x3_wgrad: null pointer");
    GN_REQUIRE((scale == nullptr) == (shift == nullptr), GOALNET_E_NULL, "conv3x3_wgrad: scale/shift must both be set or both NULL");
    GN_REQUIRE(N > 0 && H > 0 && W > 0, GOALNET_E_SHAPE, "conv3x3_wgrad: non-positive dim");
    GN_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0 && Cin > 0 && Cout > 0, GOALNET_E_SHAPE, "conv3x3_wgrad: channels must be multiples of 4");
    GN_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dw) && aligned16(ws) && aligned16(scale) && aligned16(shift),
               GOALNET_E_ALIGN, "conv3x3_wgrad: pointers must be 16-byte aligned");
    const int64_t M = (int64_t)N * H * W;
    GN_REQUIRE(M < (1ll << 31) - 256, GOALNET_E_SHAPE, "conv3x3_wgrad: N*H*W too large");
    GN_REQUIRE(ws_bytes >= goalnet_conv3x3_wgrad_ws_bytes(N, H, W, Cin, Cout), GOALNET_E_WORKSPACE, "conv3x3_wgrad: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nsplit = wgrad_splits(M, Cin, Cout);
    const int ktiles = (int)((M + BK - 1) / BK);
    const int64_t slab = (int64_t)Cout * 9 * Cin;
    MCLoader<false>::P ap{dy, Cout, Cout, (int)M, nullptr, nullptr, 1};
    EpiP ep{EPI_RAW, (float*)ws, (int64_t)9 * Cin, Cout, 9 * Cin, nullptr, 0, nullptr, 0, nullptr, 0, slab};
    int rc;
    if (scale) {
        ConvWgradBLoader<true>::P bp{x, H, W, Cin, (int)M, scale, shift};
        rc = launch_gemm<MCLoader<false>, ConvWgradBLoader<true>>("conv3x3_wgrad", ap, bp, ep, Cout, 9 * Cin, ktiles, nsplit, 1, st);
    } else {
        ConvWgradBLoader<false>::P bp{x, H, W, Cin, (int)M, nullptr, nullptr};
        rc = launch_gemm<MCLoader<false>, ConvWgradBLoader<false>>("conv3x3_wgrad", ap, bp, ep, Cout, 9 * Cin, ktiles, nsplit, 1, st);
    }
    if (rc) return rc;
    EpiP er{EPI_RAW, dw, (int64_t)9 * Cin, Cout, 9 * Cin, nullptr, 0, nullptr, 0, nullptr, 0, 0};
    return launch_reduce("conv3x3_wgrad.reduce", (const float*)ws, nsplit, slab, er, st);
}

static int linear_splits(int M, int64_t K, int J) {
    const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((J + BN - 1) / BN);
    return pick_splits(tiles, (int)(K / BK));
}

size_t goalnet_linear_fwd_ws_bytes(int M, int64_t K, int J) {
    if (M <= 0 || K <= 0 || J <= 0) return 0;
    const int s = linear_splits(M, K, J);
    return s > 1 ? (size_t)s * (size_t)M * (size_t)J * sizeof(float) : 0;
}

int goalnet_linear_fwd(const float* x, int64_t ldx, const float* scale, const float* shift, int bnC,
                       const float* w, const float* bias, int relu, const float* dropmask, int64_t ldmask,
                       float* y, int64_t ldy, float* mult_out, int64_t ldmult,
                       int M, int64_t K, int J, void* ws, size_t ws_bytes, void* stream) {
    GN_REQUIRE(x && w && y, GOALNET_E_NULL, "linear_fwd: null pointer");
    GN_REQUIRE((scale == nullptr) == (shift == nullptr), GOALNET_E_NULL, "linear_fwd: scale/shift must both be set or both NULL");
    GN_REQUIRE(M > 0 && J > 0 && K > 0 && K < (1ll << 31) - 64, GOALNET_E_SHAPE, "linear_fwd: bad dims");
    GN_REQUIRE(K % BK == 0, GOALNET_E_SHAPE, "linear_fwd: K=%lld must be a multiple of %d", (long long)K, BK);
    GN_REQUIRE(J % 4 == 0, GOALNET_E_SHAPE, "linear_fwd: J=%d must be a multiple of 4", J);
    GN_REQUIRE(!scale || (bnC > 0 && bnC % 4 == 0), GOALNET_E_SHAPE, "linear_fwd: bnC must be a positive multiple of 4");
    GN_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y) && aligned16(scale) && aligned16(shift) && ldx % 4 == 0 && ldy % 4 == 0,
               GOALNET_E_ALIGN, "linear_fwd: pointers / leading dims must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int nsplit = linear_splits(M, K, J);
    const int ktiles = (int)(K / BK);
    KCLoader<false>::P bp{w, K, J, nullptr, nullptr, 1};
    EpiP efinal{(dropmask || mult_out) ? EPI_FULL : EPI_BIAS_RELU, y, ldy, M, J, bias, relu, dropmask, ldmask, mult_out, ldmult, 0};
    EpiP ep = efinal;
    if (nsplit > 1) {
        GN_REQUIRE(ws && aligned16(ws), GOALNET_E_WORKSPACE, "linear_fwd: split-K needs a 16-byte aligned workspace");
        GN_REQUIRE(ws_bytes >= goalnet_linear_fwd_ws_bytes(M, K, J), GOALNET_E_WORKSPACE, "linear_fwd: workspace too small");
        ep = EpiP{EPI_RAW, (float*)ws, J, M, J, nullptr, 0, nullptr, 0, nullptr, 0, (int64_t)M * J};
    }
    int rc;
    if (scale) {
        KCLoader<true>::P ap{x, ldx, M, scale, shift, bnC};
        rc = launch_gemm<KCLoader<true>, KCLoader<false>>("linear_fwd", ap, bp, ep, M, J, ktiles, nsplit, 0, st);
    } else {
        KCLoader<false>::P ap{x, ldx, M, nullptr, nullptr, 1};
        rc = launch_gemm<KCLoader<false>, KCLoader<false>>("linear_fwd", ap, bp, ep, M, J, ktiles, nsplit, 0, st);
    }
    if (rc || nsplit == 1) return rc;
    return launch_reduce("linear_fwd.reduce", (const float*)ws, nsplit, (int64_t)M * J, efinal, st);
}

int goalnet_linear_bwd_dx(const float* dy, int64_t lddy, const float* w, const float* mult, int64_t ldmult,
                          float* dx, int64_t lddx, int M, int64_t K, int J, void* stream) {
    GN_REQUIRE(dy && w && dx, GOALNET_E_NULL, "linear_bwd_dx: null pointer");
    GN_REQUIRE(M > 0 && J > 0 && K > 0 && K < (1ll << 31) - 256, GOALNET_E_SHAPE, "linear_bwd_dx: bad dims");
    GN_REQUIRE(J % BK == 0, GOALNET_E_SHAPE, "linear_bwd_dx: J=%d must be a multiple of %d", J, BK);
    GN_REQUIRE(K % 4 == 0, GOALNET_E_SHAPE, "linear_bwd_dx: K must be a multiple of 4");
    GN_REQUIRE(aligned16(dy) && aligned16(w) && aligned16(dx) && lddy % 4 == 0 && lddx % 4 == 0,
               GOALNET_E_ALIGN, "linear_bwd_dx: pointers / leading dims must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    KCLoader<false>::P ap{dy, lddy, M, nullptr, nullptr, 1};
    MCLoader<false>::P bp{w, K, (int)K, J, nullptr, nullptr, 1};
    EpiP ep{mult ? EPI_MUL : EPI_RAW, dx, lddx, M, (int)K, nullptr, 0, mult, ldmult, nullptr, 0, 0};
    return launch_gemm<KCLoader<false>, MCLoader<false>>("linear_bwd_dx", ap, bp, ep, M, K, J / BK, 1, 1, st);
}

int goalnet_linear_bwd_dw(const float* dy, int64_t lddy, const float* x, int64_t ldx,
                          const float* scale, const float* shift, int bnC,
                          float* dw, int M, int64_t K, int J, void* stream) {
    GN_REQUIRE(dy && x && dw, GOALNET_E_NULL, "linear_bwd_dw: null pointer");
    GN_REQUIRE((scale == nullptr) == (shift == nullptr), GOALNET_E_NULL, "linear_bwd_dw: scale/shift must both be set or both NULL");
    GN_REQUIRE(M > 0 && J > 0 && K > 0 && K < (1ll << 31) - 256, GOALNET_E_SHAPE, "linear_bwd_dw: bad dims");
    GN_REQUIRE(J % 4 == 0 && K % 4 == 0, GOALNET_E_SHAPE, "linear_bwd_dw: J and K must be multiples of 4");
    GN_REQUIRE(!scale || (bnC > 0 && bnC % 4 == 0), GOALNET_E_SHAPE, "linear_bwd_dw: bnC must be a positive multiple of 4");
    GN_REQUIRE(aligned16(dy) && aligned16(x) && aligned16(dw) && aligned16(scale) && aligned16(shift) && lddy % 4 == 0 && ldx % 4 == 0,
               GOALNET_E_ALIGN, "linear_bwd_dw: pointers / leading dims must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int ktiles = (M + BK - 1) / BK;
    MCLoader<false>::P ap{dy, lddy, J, M, nullptr, nullptr, 1};
    EpiP ep{EPI_RAW, dw, K, J, (int)K, nullptr, 0, nullptr, 0, nullptr, 0, 0};
    if (scale) {
        MCLoader<true>::P bp{x, ldx, (int)K, M, scale, shift, bnC};
        return launch_gemm<MCLoader<false>, MCLoader<true>>("linear_bwd_dw", ap, bp, ep, J, K, ktiles, 1, 1, st);
    }
    MCLoader<false>::P bp{x, ldx, (int)K, M, nullptr, nullptr, 1};
    return launch_gemm<MCLoader<false>, MCLoader<false>>("linear_bwd_dw", ap, bp, ep, J, K, ktiles, 1, 1, st);
}

}  // extern "C"
