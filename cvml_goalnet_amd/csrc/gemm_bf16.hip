// bf16 GEMM engine on the gfx950 matrix cores (v_mfma_f32_32x32x16_bf16: bf16 in, fp32 accumulate; 1024 FLOP/clk/SIMD
// = 16x the fp32 matrix rate, ~2.5 PFLOP/s dense chip peak) for the `precision="bf16"` mode of the AVM hot path:
//
//   conv 3x3 s1 p1 forward / data-gradient   implicit GEMM; A = im2col of a bf16 NHWC tensor, B = bf16 OHWI weights
//   linear forward                           A = bf16 activations [M][K], B = bf16 weights [J][K], split-K
//
// Operands are bf16 copies made by small HBM-bound passes (goalnet_bn_apply_bf16: BatchNorm affine + cast of the
// pooled block output; goalnet_cast_bf16: weights, gradients); accumulation, bias, ReLU, BatchNorm statistics,
// master weights, gradients and Adam stay fp32. The reference is fp32 (utils.py:37-47); this mode is the north-star's
// "bf16 MFMA, logits within 1e-3" configuration and is reported separately from the fp32 path.
//
// Geometry: block tile 128 x 128 x 64 (bf16), 256 threads = 4 waves, wave tile 64 x 64 = 2 x 2 MFMA tiles, 16 MFMAs
// (512 matrix cycles) per K-tile per wave. A K-tile row is 64 bf16 = 128 B = the same LDS image as the fp32 engine
// (8 chunks of 16 B per row, chunk index XOR (row >> 1) & 7, conflict-free ds_read_b128): chunk 2*ks + h is exactly
// the 8 k-values lane-half h needs for MFMA step ks. Both operands are staged by LDS-DMA
// (`buffer_load_dwordx4 ... lds`, 1 KB per wave-instruction, swizzle applied on the source address; zero padding and
// rows past the end come from the buffer range check) into a 2-deep ring: at bf16 rates there is no VALU / VGPR
// budget for a register round trip.
#include <hip/hip_bf16.h>

#include "gemm_common.h"

using namespace goalnet;

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int BM = GEMM_BM, BN = GEMM_BN;
constexpr int BKH = 64;                 // bf16 elements per K-tile (128 B per row)
constexpr int ROWB = 128;               // bytes per LDS row
constexpr int OP_BYTES = BM * ROWB;     // 16 KB per operand per stage
constexpr unsigned OOB = 0xFFFFFF00u;

__device__ __forceinline__ int kc_boff(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t clamp_u32(int64_t v) {
    return v <= 0 ? 0u : (v > 0xFFFFFF00ll ? 0xFFFFFF00u : (uint32_t)v);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds_dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds_dst, 16, (int)voff, (int)soff, 0, 0);
}

// ---- loaders (LDS-DMA). One wave-instruction fills 8 rows x 128 B; wave w issues pieces 4w .. 4w+3 of a tile. -----
// K-contiguous bf16 matrix X[rows][K] (leading dim ld elements).
struct KCLoaderH {
    struct P { const __hip_bfloat16* x; int64_t ld; int rows; };
    __amdgpu_buffer_rsrc_t rx;
    unsigned voff[4];
    int wave;
    __device__ KCLoaderH(const P& p, int row0, int tid) {
        const int nrows = p.rows - row0 < BM ? p.rows - row0 : BM;
        rx = make_rsrc(p.x + (int64_t)row0 * p.ld, clamp_u32((int64_t)nrows * p.ld * 2));
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = (wave * 4 + i) * 8 + (lane >> 3);
            const int lc = (lane & 7) ^ ((rl >> 1) & 7);
            voff[i] = rl < nrows ? (unsigned)(((int64_t)rl * p.ld + lc * 8) * 2) : OOB;
        }
    }
    __device__ __forceinline__ void issue(int kt, char* l) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(rx, l + (wave * 4 + i) * 8 * ROWB, voff[i], (unsigned)kt * ROWB);
    }
};

// im2col of a bf16 NHWC tensor for a 3x3 / stride 1 / pad 1 convolution: row m = (n,h,w), k = (kh,kw,ci), C % 64 == 0.
// The resource starts W+1 pixels in front of the tile; padding taps are out-of-range voffsets -> zeros land in LDS.
struct ConvALoaderH {
    struct P { const __hip_bfloat16* x; int H, W, C; int64_t M; };
    __amdgpu_buffer_rsrc_t rx;
    unsigned voff[4], mask[4];
    int W, C, wave;
    __device__ ConvALoaderH(const P& p, int row0, int tid) {
        W = p.W; C = p.C;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63;
        rx = make_rsrc(p.x + ((int64_t)row0 - (p.W + 1)) * p.C, (uint32_t)((BM + 2 * (p.W + 1)) * p.C * 2));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = (wave * 4 + i) * 8 + (lane >> 3);
            const int lc = (lane & 7) ^ ((rl >> 1) & 7);
            voff[i] = (unsigned)((rl * p.C + lc * 8) * 2);
            const int64_t m = (int64_t)row0 + rl;
            unsigned mk = 0;
            if (m < p.M) {
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
        }
    }
    __device__ __forceinline__ void issue(int kt, char* l) const {
        const int k = kt * BKH;
        const int tap = k / C;
        const int ci = k - tap * C;
        const int kh = tap / 3, kw = tap - 3 * kh;
        const unsigned s0 = (unsigned)(((kh * W + kw) * C + ci) * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            dma16(rx, l + (wave * 4 + i) * 8 * ROWB, ((mask[i] >> tap) & 1u) ? voff[i] : OOB, s0);
    }
};

// ---- the MFMA tile --------------------------------------------------------------------------------------------------
__device__ __forceinline__ void compute_tile_h(const char* la, const char* lb, f32x16 (&acc)[2][2], int wm, int wn, int r, int h) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        bf16x8 a[2], b[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            a[f] = *reinterpret_cast<const bf16x8*>(la + kc_boff(wm * 64 + f * 32 + r, 2 * ks + h));
            b[f] = *reinterpret_cast<const bf16x8*>(lb + kc_boff(wn * 64 + f * 32 + r, 2 * ks + h));
        }
#pragma unroll
        for (int fm = 0; fm < 2; ++fm)
#pragma unroll
            for (int fn = 0; fn < 2; ++fn)
                acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[fm], b[fn], acc[fm][fn], 0, 0, 0);
    }
}

template <class AL, class BL>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(typename AL::P ap, typename BL::P bp, EpiP ep,
                                                           int tiles_m, int tiles_n, int m_fast,
                                                           int ktiles, int ktiles_per_split) {
    __shared__ __attribute__((aligned(16))) char lds[2][2][OP_BYTES];
    const int tid = threadIdx.x;
    int tm, tn;
    tile_of_block(tiles_m, tiles_n, m_fast, tm, tn);
    const int split = blockIdx.y;
    const int kt0 = split * ktiles_per_split;
    const int kt1 = min(ktiles, kt0 + ktiles_per_split);

    const AL al(ap, tm * BM, tid);
    const BL bl(bp, tn * BN, tid);

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

    if (kt0 < kt1) {
        al.issue(kt0, lds[0][0]);
        bl.issue(kt0, lds[0][1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        if (kt + 1 < kt1) {     // the buffer being filled was last read in iteration kt-1, left through the barrier
            al.issue(kt + 1, lds[cur ^ 1][0]);
            bl.issue(kt + 1, lds[cur ^ 1][1]);
        }
        compute_tile_h(lds[cur][0], lds[cur][1], acc, wm, wn, r, h);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the DMA data has landed before anyone passes the barrier
        __syncthreads();
    }
    store_acc<false, false>(ep, acc, tm, tn, split, wm, wn, r, h);
}

template <class AL, class BL>
int launch_gemm_h(const char* name, const typename AL::P& ap, const typename BL::P& bp, const EpiP& ep,
                  int64_t M, int64_t N, int ktiles, int nsplit, int m_fast, hipStream_t st) {
    const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    GN_REQUIRE(tiles_m * tiles_n < (1ll << 31), GOALNET_E_SHAPE, "%s: too many tiles", name);
    GN_REQUIRE(nsplit >= 1 && nsplit <= 65535, GOALNET_E_SHAPE, "%s: bad split count %d", name, nsplit);
    const int kps = (ktiles + nsplit - 1) / nsplit;
    dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)nsplit, 1);
    hipLaunchKernelGGL((gemm_bf16_kernel<AL, BL>), grid, dim3(256), 0, st, ap, bp, ep, (int)tiles_m, (int)tiles_n, m_fast,
                       ktiles, kps);
    GN_LAUNCH_CHECK(name);
    return 0;
}

// ---- fp32 -> bf16 passes (HBM-bound; 8 elements = 32 B in, 16 B out per lane) ---------------------------------------
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const __hip_bfloat16 x = __float2bfloat16(a), y = __float2bfloat16(b);
    return (unsigned)(*reinterpret_cast<const unsigned short*>(&x)) | ((unsigned)(*reinterpret_cast<const unsigned short*>(&y)) << 16);
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, __hip_bfloat16* __restrict__ y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
        reinterpret_cast<u32x4*>(y)[i] = u32x4{pack2(a.x, a.y), pack2(a.z, a.w), pack2(b.x, b.y), pack2(b.z, b.w)};
    }
}

// y = bf16(x * scale[c] + shift[c]), c = element index mod C (NHWC), C % 8 == 0
__global__ __launch_bounds__(256) void bn_apply_bf16_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, __hip_bfloat16* __restrict__ y,
                                                           int64_t n8, int C) {
    const int c8n = C >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c8n) * 8;
        const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
        const float4 s0 = *reinterpret_cast<const float4*>(scale + c), s1 = *reinterpret_cast<const float4*>(scale + c + 4);
        const float4 t0 = *reinterpret_cast<const float4*>(shift + c), t1 = *reinterpret_cast<const float4*>(shift + c + 4);
        reinterpret_cast<u32x4*>(y)[i] = u32x4{pack2(fmaf(a.x, s0.x, t0.x), fmaf(a.y, s0.y, t0.y)), pack2(fmaf(a.z, s0.z, t0.z), fmaf(a.w, s0.w, t0.w)),
                                               pack2(fmaf(b.x, s1.x, t1.x), fmaf(b.y, s1.y, t1.y)), pack2(fmaf(b.z, s1.z, t1.z), fmaf(b.w, s1.w, t1.w))};
    }
}

unsigned grid1d(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" {

int goalnet_cast_bf16(const float* x, void* y_bf16, int64_t n, void* stream) {
    GN_REQUIRE(x && y_bf16, GOALNET_E_NULL, "cast_bf16: null pointer");
    GN_REQUIRE(n > 0 && n % 8 == 0, GOALNET_E_SHAPE, "cast_bf16: element count must be a positive multiple of 8");
    GN_REQUIRE(aligned16(x) && aligned16(y_bf16), GOALNET_E_ALIGN, "cast_bf16: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid1d(n / 8)), dim3(256), 0, (hipStream_t)stream, x, (__hip_bfloat16*)y_bf16, n / 8);
    GN_LAUNCH_CHECK("cast_bf16");
    return 0;
}

int goalnet_bn_apply_bf16(const float* x, const float* scale, const float* shift, void* y_bf16, int64_t n, int C, void* stream) {
    GN_REQUIRE(x && scale && shift && y_bf16, GOALNET_E_NULL, "bn_apply_bf16: null pointer");
    GN_REQUIRE(n > 0 && C > 0 && C % 8 == 0 && n % C == 0, GOALNET_E_SHAPE, "bn_apply_bf16: n must be a multiple of C, C of 8");
    GN_REQUIRE(aligned16(x) && aligned16(y_bf16) && aligned16(scale) && aligned16(shift), GOALNET_E_ALIGN, "bn_apply_bf16: alignment");
    hipLaunchKernelGGL(bn_apply_bf16_kernel, dim3(grid1d(n / 8)), dim3(256), 0, (hipStream_t)stream, x, scale, shift,
                       (__hip_bfloat16*)y_bf16, n / 8, C);
    GN_LAUNCH_CHECK("bn_apply_bf16");
    return 0;
}

int goalnet_conv3x3_fwd_bf16(const void* x_bf16, const void* w_bf16, const float* bias, int relu, float* y,
                             int N, int H, int W, int Cin, int Cout, void* stream) {
    GN_REQUIRE(x_bf16 && w_bf16 && y, GOALNET_E_NULL, "conv3x3_fwd_bf16: null pointer");
    GN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, GOALNET_E_SHAPE, "conv3x3_fwd_bf16: non-positive dim");
    GN_REQUIRE(Cin % BKH == 0, GOALNET_E_SHAPE, "conv3x3_fwd_bf16: Cin=%d must be a multiple of %d", Cin, BKH);
    GN_REQUIRE(Cout % 4 == 0, GOALNET_E_SHAPE, "conv3x3_fwd_bf16: Cout=%d must be a multiple of 4", Cout);
    GN_REQUIRE(aligned16(x_bf16) && aligned16(w_bf16) && aligned16(y), GOALNET_E_ALIGN, "conv3x3_fwd_bf16: pointers must be 16-byte aligned");
    const int64_t M = (int64_t)N * H * W;
    GN_REQUIRE(M < (1ll << 31) - 256, GOALNET_E_SHAPE, "conv3x3_fwd_bf16: N*H*W too large");
    ConvALoaderH::P ap{(const __hip_bfloat16*)x_bf16, H, W, Cin, M};
    KCLoaderH::P bp{(const __hip_bfloat16*)w_bf16, (int64_t)9 * Cin, Cout};
    EpiP ep{EPI_BIAS_RELU, y, Cout, (int)M, Cout, bias, relu, nullptr, 0, nullptr, 0, 0};
    return launch_gemm_h<ConvALoaderH, KCLoaderH>("conv3x3_fwd_bf16", ap, bp, ep, M, Cout, 9 * Cin / BKH, 1, 0, (hipStream_t)stream);
}

static int linear_splits_h(int M, int64_t K, int J) {
    const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((J + BN - 1) / BN);
    return pick_splits(tiles, (int)(K / BKH));
}

size_t goalnet_linear_fwd_bf16_ws_bytes(int M, int64_t K, int J) {
    if (M <= 0 || K <= 0 || J <= 0) return 0;
    const int s = linear_splits_h(M, K, J);
    return s > 1 ? (size_t)s * (size_t)M * (size_t)J * sizeof(float) : 0;
}

int goalnet_linear_fwd_bf16(const void* x_bf16, int64_t ldx, const void* w_bf16, const float* bias, int relu,
                            const float* dropmask, int64_t ldmask, float* y, int64_t ldy, float* mult_out, int64_t ldmult,
                            int M, int64_t K, int J, void* ws, size_t ws_bytes, void* stream) {
    GN_REQUIRE(x_bf16 && w_bf16 && y, GOALNET_E_NULL, "linear_fwd_bf16: null pointer");
    GN_REQUIRE(M > 0 && J > 0 && K > 0 && K < (1ll << 31) - 64, GOALNET_E_SHAPE, "linear_fwd_bf16: bad dims");
    GN_REQUIRE(K % BKH == 0 && J % 4 == 0 && ldx % 8 == 0 && ldy % 4 == 0, GOALNET_E_SHAPE, "linear_fwd_bf16: K %% 64, J %% 4, ldx %% 8");
    GN_REQUIRE(aligned16(x_bf16) && aligned16(w_bf16) && aligned16(y), GOALNET_E_ALIGN, "linear_fwd_bf16: alignment");
    hipStream_t st = (hipStream_t)stream;
    const int nsplit = linear_splits_h(M, K, J);
    KCLoaderH::P ap{(const __hip_bfloat16*)x_bf16, ldx, M};
    KCLoaderH::P bp{(const __hip_bfloat16*)w_bf16, K, J};
    EpiP efinal{(dropmask || mult_out) ? EPI_FULL : EPI_BIAS_RELU, y, ldy, M, J, bias, relu, dropmask, ldmask, mult_out, ldmult, 0};
    EpiP ep = efinal;
    if (nsplit > 1) {
        GN_REQUIRE(ws && aligned16(ws) && ws_bytes >= goalnet_linear_fwd_bf16_ws_bytes(M, K, J), GOALNET_E_WORKSPACE,
                   "linear_fwd_bf16: split-K needs a 16-byte aligned workspace of goalnet_linear_fwd_bf16_ws_bytes()");
        ep = EpiP{EPI_RAW, (float*)ws, J, M, J, nullptr, 0, nullptr, 0, nullptr, 0, (int64_t)M * J};
    }
    const int rc = launch_gemm_h<KCLoaderH, KCLoaderH>("linear_fwd_bf16", ap, bp, ep, M, J, (int)(K / BKH), nsplit, 0, st);
    if (rc || nsplit == 1) return rc;
    return launch_splitk_reduce("linear_fwd_bf16.reduce", (const float*)ws, nsplit, (int64_t)M * J, efinal, st);
}

}  // extern "C"
