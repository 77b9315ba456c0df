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
#include "gemm_bf16_common.h"

using namespace goalnet;

namespace {

constexpr int BM = GEMM_BM, BN = GEMM_BN;

// ---- loaders (LDS-DMA). One wave-instruction fills 8 rows x 128 B; wave w issues pieces 4w .. 4w+3 of a tile. -----
// K-contiguous bf16 matrix X[rows][K] (leading dim ld elements).
struct KCLoaderH {
    struct P { const __hip_bfloat16* x; int64_t ld; int rows; };
    static constexpr bool TR = false;
    __amdgpu_buffer_rsrc_t rx;
    unsigned voff[4];
    int wave, npieces;         // pieces (8 rows each) that hold rows of the matrix: the others are not staged at all
    __device__ KCLoaderH(const P& p, int row0, int tid, int tile_rows = BM) {
        const int nrows = p.rows - row0 < tile_rows ? p.rows - row0 : tile_rows;
        npieces = tile_rows / 8;
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
        for (int i = 0; i < 4; ++i)
            if (wave * 4 + i < npieces) dma16(rx, l + (wave * 4 + i) * 8 * ROWB, voff[i], (unsigned)kt * ROWB);
    }
};

// im2col of a bf16 NHWC tensor for a 3x3 / stride 1 / pad 1 convolution: row m = (n,h,w), k = (kh,kw,ci), C % 64 == 0.
// The resource starts W+1 pixels in front of the tile; padding taps are out-of-range voffsets -> zeros land in LDS.
struct ConvALoaderH {
    struct P { const __hip_bfloat16* x; int H, W, C; int64_t M; };
    static constexpr bool TR = false;
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

// ---- row-contiguous ("transposed") operands: the reduction index is the slow index in memory -------------------------
// LDS image [64 k-rows][128 cols] bf16 = 256-B rows, filled by DMA (1 KB = 4 k-rows per wave-instruction) and read with
// ds_read_b64_tr_b16 (each 16-lane group fetches a 4 x 16 block and receives it column-major = 4 consecutive k of one
// column per lane; two reads = the 8 k-values of an MFMA operand). The sixteen 16-B chunks of a row are XOR-swizzled
// by 2 * (krow & 7) (on the DMA source address), and a 32-row MFMA fragment takes the 16-column units {u, u + 4}: the two
// groups of a 32-lane half then read disjoint 128-B halves of the bank row -> conflict-free (layout derived from the
// bank rule of MI355X_MICROARCH.md §LDS; semantics of the transposed read probed in scripts/probe/tr_probe.hip).
constexpr int TROWB = 256;
__device__ __forceinline__ int tr_chunk(int krow, int chunk) { return chunk ^ (2 * (krow & 7)); }

// plain matrix X[kred][cols] (leading dim ld elements); re-based every K-tile, rows past kred / cols past `cols` read 0
struct MCLoaderH {
    struct P { const __hip_bfloat16* x; int64_t ld; int cols; int kred; };
    static constexpr bool TR = true;
    const __hip_bfloat16* x;
    int64_t ld;
    int kred, wave;
    unsigned voff[4];
    __device__ MCLoaderH(const P& p, int col0, int tid) {
        x = p.x; ld = p.ld; kred = p.kred;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kr = (wave * 4 + i) * 4 + (lane >> 4);
            const int col = col0 + tr_chunk(kr, lane & 15) * 8;
            voff[i] = col < p.cols ? (unsigned)(((int64_t)kr * p.ld + col) * 2) : OOB;
        }
    }
    __device__ __forceinline__ void issue(int kt, char* l) const {
        const int kbase = kt * BKH;
        const int nk = kred - kbase < BKH ? kred - kbase : BKH;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (int64_t)kbase * ld, clamp_u32((int64_t)nk * ld * 2));
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(rx, l + (wave * 4 + i) * 4 * TROWB, voff[i], 0);
    }
};

// B operand of the conv weight gradient on the zero-PADDED pixel grid: B(col = (tap, ci), k = pm) = xpad[pm + shift(tap)][ci].
// With a zero border around every frame the tap shift is a constant pixel offset for the whole reduction (no per-pixel
// validity), and pad pixels contribute nothing because dy_pad is zero there. `x` points at padded pixel 0 of a buffer
// that has G = W+3 zero guard pixels (padded width W+2) in front and behind.
struct ConvWgradBLoaderH {
    struct P { const __hip_bfloat16* x; int Wp2, C; int64_t Mp; };     // Wp2 = W + 2, Mp = N*(H+2)*(W+2)
    static constexpr bool TR = true;
    const __hip_bfloat16* x;
    int C, G, wave;
    unsigned voff[4];
    __device__ ConvWgradBLoaderH(const P& p, int col0, int tid) {
        x = p.x; C = p.C; G = p.Wp2 + 1;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kr = (wave * 4 + i) * 4 + (lane >> 4);
            const int col = col0 + tr_chunk(kr, lane & 15) * 8;
            if (col < 9 * p.C) {
                const int tap = col / p.C, ci = col - tap * p.C;
                const int kh = tap / 3, kw = tap - 3 * kh;
                voff[i] = (unsigned)(((kr + kh * p.Wp2 + kw) * p.C + ci) * 2);       // relative to pixel (kbase - G)
            } else {
                voff[i] = OOB;
            }
        }
    }
    __device__ __forceinline__ void issue(int kt, char* l) const {
        const int64_t kbase = (int64_t)kt * BKH;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(x + (kbase - G) * C, (uint32_t)((BKH + 2 * G) * C * 2));
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(rx, l + (wave * 4 + i) * 4 * TROWB, voff[i], 0);
    }
};

// im2col of a zero-padded bf16 NHWC tensor (same buffer convention as above): row m = (n,h,w) of the OUTPUT grid,
// k = (kh,kw,ci); no validity masks at all.
struct ConvAPadLoaderH {
    struct P { const __hip_bfloat16* x; int H, W, C; int64_t M; };
    static constexpr bool TR = false;
    __amdgpu_buffer_rsrc_t rx;
    unsigned voff[4];
    int Wp2, C, wave;
    __device__ ConvAPadLoaderH(const P& p, int row0, int tid) {
        Wp2 = p.W + 2; C = p.C;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63;
        const int hw = p.H * p.W;
        const double inv_hw = 1.0 / (double)hw, inv_w = 1.0 / (double)p.W;       // rows and pixels stay below 2^31 (launcher)
        const int n0 = fast_div(row0, inv_hw), r0 = row0 - n0 * hw, h0 = fast_div(r0, inv_w), w0 = r0 - h0 * p.W;
        const int64_t pm0 = ((int64_t)n0 * (p.H + 2) + h0 + 1) * Wp2 + w0 + 1;      // padded index of the tile's first pixel
        const int G = Wp2 + 1;
        // a tile of 128 consecutive output pixels spans < 128 + 2 * (rows crossed + frames crossed * (W+2)) padded pixels
        rx = make_rsrc(p.x + (pm0 - G) * p.C, (uint32_t)((BM + 2 * (BM / p.W + 2) + 4 * Wp2 + 2 * G) * p.C * 2));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = (wave * 4 + i) * 8 + (lane >> 3);
            const int lc = (lane & 7) ^ ((rl >> 1) & 7);
            const int64_t m = (int64_t)row0 + rl;
            if (m < p.M) {
                const int n = fast_div((int)m, inv_hw), rr = (int)m - n * hw, h = fast_div(rr, inv_w), w = rr - h * p.W;
                const int64_t pm = ((int64_t)n * (p.H + 2) + h + 1) * Wp2 + w + 1;
                voff[i] = (unsigned)(((pm - pm0) * p.C + lc * 8) * 2);
            } else {
                voff[i] = OOB;
            }
        }
    }
    __device__ __forceinline__ void issue(int kt, char* l) const {
        const int k = kt * BKH;
        const int tap = k / C;
        const int ci = k - tap * C;
        const int kh = tap / 3, kw = tap - 3 * kh;
        const unsigned s0 = (unsigned)(((kh * Wp2 + kw) * C + ci) * 2);      // (kh-1, kw-1) shift + the G-pixel lead
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(rx, l + (wave * 4 + i) * 8 * ROWB, voff[i], s0);
    }
};

// ---- the MFMA tile --------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// tile row (or column) of MFMA index i (0..31) of fragment f of strip s for the two operand forms
template <bool TR>
__device__ __forceinline__ int frag_index(int s, int f, int i) {
    return TR ? 16 * (2 * s + f + 4 * (i >> 4)) + (i & 15) : s * 64 + f * 32 + i;
}

template <bool TR>
__device__ __forceinline__ bf16x8 read_frag_h(const char* l, int s, int f, int ks, int lane) {
    if (!TR) {
        const int r = lane & 31, h = lane >> 5;
        return *reinterpret_cast<const bf16x8*>(l + kc_boff(s * 64 + f * 32 + r, 2 * ks + h));
    } else {
        const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const int unit = 2 * s + f + 4 * (g & 1);                    // 16-column unit this 16-lane group transposes
        const int chunk = 2 * unit + (p >> 1);
        s16x4 v[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int kr = 16 * ks + 8 * (g >> 1) + 4 * t + q;
            const char* a = l + kr * TROWB + tr_chunk(kr, chunk) * 16 + (p & 1) * 8;
            v[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a);
        }
        return bf16x8{v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w};
    }
}

template <bool ATR, bool BTR, bool F16>
__device__ __forceinline__ void compute_tile_h(const char* la, const char* lb, f32x16 (&acc)[2][2], int wm, int wn, int lane) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        bf16x8 a[2], b[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            a[f] = read_frag_h<ATR>(la, wm, f, ks, lane);
            b[f] = read_frag_h<BTR>(lb, wn, f, ks, lane);
        }
#pragma unroll
        for (int fm = 0; fm < 2; ++fm)
#pragma unroll
            for (int fn = 0; fn < 2; ++fn)
                if constexpr (F16)
                    acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[fm]), __builtin_bit_cast(f16x8, b[fn]),
                                                                        acc[fm][fn], 0, 0, 0);
                else
                    acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[fm], b[fn], acc[fm][fn], 0, 0, 0);
    }
}

// epilogue for either operand form (gemm_common.h's store_acc covers the K-contiguous/K-contiguous case)
template <bool ATR, bool BTR>
__device__ __forceinline__ void store_acc_h(const EpiP& ep, const f32x16 (&acc)[2][2], int tm, int tn, int split,
                                            int wm, int wn, int lane) {
    float* outp = ep.out + (ep.slab_stride > 0 ? (int64_t)split * ep.slab_stride : 0);
    const int mode = ep.slab_stride > 0 ? EPI_RAW : ep.mode;
    const int r = lane & 31, h = lane >> 5;
    // 16-byte stores after an in-quad transpose (gemm_common.h: quad_transpose4) for the modes whose epilogue is a function of
    // (value, column) [+ one multiplier per element]; needs 16-byte aligned rows. EPI_FULL keeps the element-wise form.
    const bool wide = mode != EPI_FULL && (ep.ld & 3) == 0 && (reinterpret_cast<uintptr_t>(outp) & 15u) == 0 &&
                      (mode != EPI_MUL || ((ep.ldmul & 3) == 0 && (reinterpret_cast<uintptr_t>(ep.mul) & 15u) == 0));
    if (wide) {
        const int r4 = r & ~3;
        const float lo = (mode == EPI_BIAS_RELU && ep.relu) ? 0.f : -INFINITY;
#pragma unroll
        for (int fm = 0; fm < 2; ++fm)
#pragma unroll
            for (int fn = 0; fn < 2; ++fn) {
                const int col = tn * BN + frag_index<BTR>(wn, fn, r4);               // first of 4 consecutive columns
                const bool colok = col < ep.cols;                                    // cols % 4 == 0 at every call site
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (mode == EPI_BIAS_RELU && ep.bias && colok) bv = *reinterpret_cast<const float4*>(ep.bias + col);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float n0 = acc[fm][fn][4 * g], n1 = acc[fm][fn][4 * g + 1], n2 = acc[fm][fn][4 * g + 2], n3 = acc[fm][fn][4 * g + 3];
                    quad_transpose4(n0, n1, n2, n3, lane);
                    const int64_t row = (int64_t)tm * BM + frag_index<ATR>(wm, fm, (lane & 3) + 8 * g + 4 * h);
                    if (!(colok && row < ep.rows)) continue;
                    float4 v = make_float4(n0, n1, n2, n3);
                    if (mode == EPI_BIAS_RELU) v = make_float4(fmaxf(n0 + bv.x, lo), fmaxf(n1 + bv.y, lo), fmaxf(n2 + bv.z, lo), fmaxf(n3 + bv.w, lo));
                    else if (mode == EPI_MUL) {
                        const float4 mv = *reinterpret_cast<const float4*>(ep.mul + row * ep.ldmul + col);
                        v = make_float4(n0 * mv.x, n1 * mv.y, n2 * mv.z, n3 * mv.w);
                    }
                    *reinterpret_cast<float4*>(outp + row * ep.ld + col) = v;
                }
            }
        return;
    }
#pragma unroll
    for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int fn = 0; fn < 2; ++fn) {
            const int col = tn * BN + frag_index<BTR>(wn, fn, r);
            const bool colok = col < ep.cols;
            const int colc = colok ? col : 0;
            float mv[16];
            if (mode == EPI_MUL) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t row = (int64_t)tm * BM + frag_index<ATR>(wm, fm, (e & 3) + 8 * (e >> 2) + 4 * h);
                    mv[e] = ep.mul[(row < ep.rows ? row : ep.rows - 1) * ep.ldmul + colc];
                }
            }
            const float bv = (mode == EPI_BIAS_RELU && ep.bias) ? ep.bias[colc] : 0.f;
            const float lo = (mode == EPI_BIAS_RELU && ep.relu) ? 0.f : -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t row = (int64_t)tm * BM + frag_index<ATR>(wm, fm, (e & 3) + 8 * (e >> 2) + 4 * h);
                if (!(colok && row < ep.rows)) continue;
                const float v = acc[fm][fn][e];
                if (mode == EPI_RAW) outp[row * ep.ld + col] = v;
                else if (mode == EPI_BIAS_RELU) outp[row * ep.ld + col] = fmaxf(v + bv, lo);
                else if (mode == EPI_MUL) outp[row * ep.ld + col] = v * mv[e];
                else outp[row * ep.ld + col] = epi_apply(ep, v, row, col);
            }
        }
}

// ---- narrow outputs (N <= 64: conv2's data gradient): block tile 128 x 64, the four waves stacked along M (32 rows x 64
// columns each = 1 x 2 MFMA tiles); LDS images and loaders unchanged (rows 64..127 of the B tile are out of range -> zeros);
// K-contiguous operands only. On the 128-wide tile half of every MFMA multiplied zero columns.
template <bool F16>
__device__ __forceinline__ void compute_tile_h_n64(const char* la, const char* lb, f32x16 (&acc)[2], int wave, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(la + kc_boff(wave * 32 + r, 2 * ks + h));
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(lb + kc_boff(r, 2 * ks + h));
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(lb + kc_boff(32 + r, 2 * ks + h));
        if constexpr (F16) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b0), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b1), acc[1], 0, 0, 0);
        } else {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc[1], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ void store_acc_h_n64(const EpiP& ep, const f32x16 (&acc)[2], int tm, int tn, int split, int wave, int lane) {
    float* outp = ep.out + (ep.slab_stride > 0 ? (int64_t)split * ep.slab_stride : 0);
    const int mode = ep.slab_stride > 0 ? EPI_RAW : ep.mode;
    const int r = lane & 31, h = lane >> 5;
    const int osc = ep.oscale ? *ep.oscale : 0;                    // split operands in scaled fp16 (csrc/split3.hip): exponent to add; 0 otherwise
#pragma unroll
    for (int fn = 0; fn < 2; ++fn) {
        const int col = tn * 64 + fn * 32 + r;
        const bool colok = col < ep.cols;
        const float bv = (mode == EPI_BIAS_RELU && ep.bias) ? ep.bias[colok ? col : 0] : 0.f;
        const float lo = (mode == EPI_BIAS_RELU && ep.relu) ? 0.f : -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int64_t row = (int64_t)tm * BM + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (!(colok && row < ep.rows)) continue;
            const float v = acc[fn][e];
            if (mode == EPI_RAW) outp[row * ep.ld + col] = v;
            else if (mode == EPI_BIAS_RELU) outp[row * ep.ld + col] = fmaxf(ldexpf(v, osc) + bv, lo);
            else outp[row * ep.ld + col] = epi_apply(ep, v, row, col);
        }
    }
}

// N64: the B tile is 64 rows (8 KB; waves 2, 3 stage nothing of it): 24 KB per stage, three blocks per CU
template <class AL, class BL, bool F16, bool N64 = false>
__global__ __launch_bounds__(256, N64 ? 3 : 2) void gemm_bf16_kernel(typename AL::P ap, typename BL::P bp, EpiP ep,
                                                           int tiles_m, int tiles_n, int m_fast,
                                                           int ktiles, int ktiles_per_split) {
    __shared__ __attribute__((aligned(16))) char lds[2][N64 ? 3 * OP_BYTES / 2 : 2 * OP_BYTES];
    const int tid = threadIdx.x;
    int tm, tn;
    tile_of_block(tiles_m, tiles_n, m_fast, tm, tn);
    const int split = blockIdx.y;
    const int kt0 = split * ktiles_per_split;
    const int kt1 = min(ktiles, kt0 + ktiles_per_split);

    static_assert(!N64 || (!AL::TR && !BL::TR), "the 128 x 64 tile reads K-contiguous LDS images");
    const AL al(ap, tm * BM, tid);
    const BL bl = [&] { if constexpr (N64) return BL(bp, tn * 64, tid, 64); else return BL(bp, tn * BN, tid); }();

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (kt0 < kt1) {
        al.issue(kt0, lds[0]);
        bl.issue(kt0, lds[0] + OP_BYTES);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        if (kt + 1 < kt1) {     // the buffer being filled was last read in iteration kt-1, left through the barrier
            al.issue(kt + 1, lds[cur ^ 1]);
            bl.issue(kt + 1, lds[cur ^ 1] + OP_BYTES);
        }
        if constexpr (N64) compute_tile_h_n64<F16>(lds[cur], lds[cur] + OP_BYTES, acc[0], wave, lane);
        else compute_tile_h<AL::TR, BL::TR, F16>(lds[cur], lds[cur] + OP_BYTES, acc, wm, wn, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the DMA data has landed before anyone passes the barrier
        __syncthreads();
    }
    if constexpr (N64) store_acc_h_n64(ep, acc[0], tm, tn, split, wave, lane);
    else store_acc_h<AL::TR, BL::TR>(ep, acc, tm, tn, split, wm, wn, lane);
}

// A short reduction with a huge output (linear5 dW: 16 K-tiles, 5 GB written): the time goes into per-block latency
// (prologue wait, a DMA round trip per K-tile, the tile store), not into MFMA or LDS. One LDS stage (32 KB) instead of two
// lets 4 blocks share a CU, which hides those latencies behind each other.
template <class AL, class BL, bool F16>
__global__ __launch_bounds__(256, 4) void gemm_bf16_1stage_kernel(typename AL::P ap, typename BL::P bp, EpiP ep,
                                                                  int tiles_m, int tiles_n, int m_fast, int ktiles) {
    __shared__ __attribute__((aligned(16))) char lds[2][OP_BYTES];
    const int tid = threadIdx.x;
    int tm, tn;
    tile_of_block(tiles_m, tiles_n, m_fast, tm, tn);
    const AL al(ap, tm * BM, tid);
    const BL bl(bp, tn * BN, tid);
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int kt = 0; kt < ktiles; ++kt) {
        al.issue(kt, lds[0]);
        bl.issue(kt, lds[1]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        compute_tile_h<AL::TR, BL::TR, F16>(lds[0], lds[1], acc, wm, wn, lane);
        __syncthreads();                                        // everyone has read the tile before it is overwritten
    }
    store_acc_h<AL::TR, BL::TR>(ep, acc, tm, tn, 0, wm, wn, lane);
}

template <class AL, class BL>
int launch_gemm_h_1stage(const char* name, const typename AL::P& ap, const typename BL::P& bp, const EpiP& ep,
                         int64_t M, int64_t N, int ktiles, int m_fast, bool f16, hipStream_t st) {
    const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    GN_REQUIRE(tiles_m * tiles_n < (1ll << 31), GOALNET_E_SHAPE, "%s: too many tiles", name);
    if (f16) hipLaunchKernelGGL((gemm_bf16_1stage_kernel<AL, BL, true>), dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 0, st, ap, bp, ep,
                                (int)tiles_m, (int)tiles_n, m_fast, ktiles);
    else hipLaunchKernelGGL((gemm_bf16_1stage_kernel<AL, BL, false>), dim3((unsigned)(tiles_m * tiles_n)), dim3(256), 0, st, ap, bp, ep,
                            (int)tiles_m, (int)tiles_n, m_fast, ktiles);
    GN_LAUNCH_CHECK(name);
    return 0;
}

template <class AL, class BL, bool N64 = false>
int launch_gemm_h(const char* name, const typename AL::P& ap, const typename BL::P& bp, const EpiP& ep,
                  int64_t M, int64_t N, int ktiles, int nsplit, int m_fast, bool f16, hipStream_t st) {
    const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + (N64 ? 64 : BN) - 1) / (N64 ? 64 : BN);
    GN_REQUIRE(tiles_m * tiles_n < (1ll << 31), GOALNET_E_SHAPE, "%s: too many tiles", name);
    GN_REQUIRE(nsplit >= 1 && nsplit <= 65535, GOALNET_E_SHAPE, "%s: bad split count %d", name, nsplit);
    const int kps = (ktiles + nsplit - 1) / nsplit;
    dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)nsplit, 1);
    if (f16) hipLaunchKernelGGL((gemm_bf16_kernel<AL, BL, true, N64>), grid, dim3(256), 0, st, ap, bp, ep, (int)tiles_m, (int)tiles_n, m_fast,
                                ktiles, kps);
    else hipLaunchKernelGGL((gemm_bf16_kernel<AL, BL, false, N64>), grid, dim3(256), 0, st, ap, bp, ep, (int)tiles_m, (int)tiles_n, m_fast,
                            ktiles, kps);
    GN_LAUNCH_CHECK(name);
    return 0;
}

// ---- fp32 <-> 16-bit passes (HBM-bound; 8 elements = 32 B in, 16 B out per lane); F16: IEEE fp16 instead of bf16 ----------
template <bool F16>
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, __hip_bfloat16* __restrict__ y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
        reinterpret_cast<u32x4*>(y)[i] = u32x4{pack2_h16<F16>(a.x, a.y), pack2_h16<F16>(a.z, a.w), pack2_h16<F16>(b.x, b.y), pack2_h16<F16>(b.z, b.w)};
    }
}

// 8 consecutive elements of an fp32 or 16-bit tensor as two float4 (H16: the tensor is 16-bit, of format F16)
template <bool H16, bool F16>
__device__ __forceinline__ void load8(const void* x, int64_t i, float4& a, float4& b) {
    if (!H16) {
        a = reinterpret_cast<const float4*>(x)[2 * i]; b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    } else {
        const u32x4 v = reinterpret_cast<const u32x4*>(x)[i];
        a = make_float4(unpack_lo<F16>(v.x), unpack_hi<F16>(v.x), unpack_lo<F16>(v.y), unpack_hi<F16>(v.y));
        b = make_float4(unpack_lo<F16>(v.z), unpack_hi<F16>(v.z), unpack_lo<F16>(v.w), unpack_hi<F16>(v.w));
    }
}

template <bool F16>
__global__ __launch_bounds__(256) void cast_f32_kernel(const __hip_bfloat16* __restrict__ x, float* __restrict__ y, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        float4 a, b;
        load8<true, F16>(x, i, a, b);
        reinterpret_cast<float4*>(y)[2 * i] = a;
        reinterpret_cast<float4*>(y)[2 * i + 1] = b;
    }
}

// y = h16(x * scale[c] + shift[c]), c = element index mod C (NHWC), C % 8 == 0
template <bool H16, bool F16>
__global__ __launch_bounds__(256) void bn_apply_bf16_kernel(const void* __restrict__ x, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, __hip_bfloat16* __restrict__ y,
                                                           int64_t n8, int C) {
    const int c8n = C >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c8n) * 8;
        float4 a, b;
        load8<H16, F16>(x, i, a, b);
        const float4 s0 = *reinterpret_cast<const float4*>(scale + c), s1 = *reinterpret_cast<const float4*>(scale + c + 4);
        const float4 t0 = *reinterpret_cast<const float4*>(shift + c), t1 = *reinterpret_cast<const float4*>(shift + c + 4);
        reinterpret_cast<u32x4*>(y)[i] = u32x4{pack2_h16<F16>(fmaf(a.x, s0.x, t0.x), fmaf(a.y, s0.y, t0.y)), pack2_h16<F16>(fmaf(a.z, s0.z, t0.z), fmaf(a.w, s0.w, t0.w)),
                                               pack2_h16<F16>(fmaf(b.x, s1.x, t1.x), fmaf(b.y, s1.y, t1.y)), pack2_h16<F16>(fmaf(b.z, s1.z, t1.z), fmaf(b.w, s1.w, t1.w))};
    }
}

// x fp32 / 16-bit [N][H][W][C] -> 16-bit zero-padded [N][H+2][W+2][C] (interior only; the caller zeroed the buffer once),
// optional per-channel affine. One thread = 8 channels of one pixel.
template <bool H16, bool F16>
__global__ __launch_bounds__(256) void to_bf16_padded_kernel(const void* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, __hip_bfloat16* __restrict__ y,
                                                            int64_t n8, int H, int W, int C) {
    const int c8n = C >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c8n) * 8;
        const int64_t pix = i / c8n;
        const int w = (int)(pix % W);
        const int64_t t = pix / W;
        const int h = (int)(t % H);
        const int64_t n = t / H;
        const int64_t pm = (n * (H + 2) + h + 1) * (W + 2) + w + 1;
        float4 a, b;
        load8<H16, F16>(x, i, a, b);
        if (scale) {
            const float4 s0 = *reinterpret_cast<const float4*>(scale + c), s1 = *reinterpret_cast<const float4*>(scale + c + 4);
            const float4 t0 = *reinterpret_cast<const float4*>(shift + c), t1 = *reinterpret_cast<const float4*>(shift + c + 4);
            a = make_float4(fmaf(a.x, s0.x, t0.x), fmaf(a.y, s0.y, t0.y), fmaf(a.z, s0.z, t0.z), fmaf(a.w, s0.w, t0.w));
            b = make_float4(fmaf(b.x, s1.x, t1.x), fmaf(b.y, s1.y, t1.y), fmaf(b.z, s1.z, t1.z), fmaf(b.w, s1.w, t1.w));
        }
        *reinterpret_cast<u32x4*>(y + pm * C + c) = u32x4{pack2_h16<F16>(a.x, a.y), pack2_h16<F16>(a.z, a.w), pack2_h16<F16>(b.x, b.y), pack2_h16<F16>(b.z, b.w)};
    }
}

unsigned grid1d(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ---- split operands (csrc/split3.hip) on the 128 x 64 tile: conv2's data gradient (256 -> 64 channels). Tap-major K order over
// NSEG x segC virtual channels per tap; virtual 64-channel chunk c reads part (segmap >> 4 (c % NSEG)) & 15 of chunk c / NSEG.
template <int NSEG>
__device__ __forceinline__ void seg_tap_channel(int kt, int segC, unsigned segmap, int& tap, int& ci) {
    const int cpt = NSEG * segC / BKH;                       // virtual chunks per tap
    tap = kt / cpt;
    const int chunk = kt - tap * cpt, cc = chunk / NSEG, seg = chunk - NSEG * cc;
    ci = (int)((segmap >> (4 * seg)) & 15u) * segC + cc * BKH;
}
template <int NSEG>
struct KCLoaderHS : KCLoaderH {
    struct P { const __hip_bfloat16* x; int64_t ld; int rows; int convC; int segC; unsigned segmap; };   // rows of 9 x convC stored values
    int convC, segC;
    unsigned segmap;
    __device__ KCLoaderHS(const P& p, int row0, int tid, int tile_rows = BM)
        : KCLoaderH(KCLoaderH::P{p.x, p.ld, p.rows}, row0, tid, tile_rows), convC(p.convC), segC(p.segC), segmap(p.segmap) {}
    __device__ __forceinline__ void issue(int kt, char* l) const {
        int tap, ci;
        seg_tap_channel<NSEG>(kt, segC, segmap, tap, ci);
        const unsigned koff = (unsigned)((tap * convC + ci) * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (wave * 4 + i < npieces) dma16(rx, l + (wave * 4 + i) * 8 * ROWB, voff[i], koff);
    }
};
template <int NSEG>
struct ConvAPadLoaderHS : ConvAPadLoaderH {
    struct P { const __hip_bfloat16* x; int H, W, C; int64_t M; int segC; unsigned segmap; };            // C = stored channels per pixel
    int segC;
    unsigned segmap;
    __device__ ConvAPadLoaderHS(const P& p, int row0, int tid)
        : ConvAPadLoaderH(ConvAPadLoaderH::P{p.x, p.H, p.W, p.C, p.M}, row0, tid), segC(p.segC), segmap(p.segmap) {}
    __device__ __forceinline__ void issue(int kt, char* l) const {
        int tap, ci;
        seg_tap_channel<NSEG>(kt, segC, segmap, tap, ci);
        const int kh = tap / 3, kw = tap - 3 * kh;
        const unsigned s0 = (unsigned)(((kh * Wp2 + kw) * C + ci) * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(rx, l + (wave * 4 + i) * 8 * ROWB, voff[i], s0);
    }
};

}  // namespace

namespace goalnet {
// conv 3x3 (forward form; the data gradient of a layer with <= 64 input channels) on split operands, 128 x 64 tile, fp32 result.
// parts = 3: bf16 triples, six segments; parts = 2: scaled fp16 pairs, three segments (ep.oscale)
int launch_conv_split_h64(const char* name, int parts, const __hip_bfloat16* x_pads, int H, int W, int Cin, int64_t M,
                          const __hip_bfloat16* ws, int Cout, const EpiP& ep, hipStream_t st) {
    if (parts == 3) {
        typedef ConvAPadLoaderHS<6> AL;
        typedef KCLoaderHS<6> BL;
        AL::P ap{x_pads, H, W, 3 * Cin, M, Cin, 0x010201u};
        BL::P bp{ws, (int64_t)9 * 3 * Cin, Cout, 3 * Cin, Cin, 0x001021u};
        return launch_gemm_h<AL, BL, true>(name, ap, bp, ep, M, Cout, 9 * 6 * Cin / BKH, 1, 0, false, st);
    }
    typedef ConvAPadLoaderHS<3> AL;
    typedef KCLoaderHS<3> BL;
    AL::P ap{x_pads, H, W, 2 * Cin, M, Cin, 0x010u};
    BL::P bp{ws, (int64_t)9 * 2 * Cin, Cout, 2 * Cin, Cin, 0x001u};
    return launch_gemm_h<AL, BL, true>(name, ap, bp, ep, M, Cout, 9 * 3 * Cin / BKH, 1, 0, true, st);
}
}  // namespace goalnet

extern "C" {

int goalnet_cast_bf16(const float* x, void* y_bf16, int64_t n, int f16, void* stream) {
    GN_REQUIRE(x && y_bf16, GOALNET_E_NULL, "cast_bf16: null pointer");
    GN_REQUIRE(n > 0 && n % 8 == 0, GOALNET_E_SHAPE, "cast_bf16: element count must be a positive multiple of 8");
    GN_REQUIRE(aligned16(x) && aligned16(y_bf16), GOALNET_E_ALIGN, "cast_bf16: pointers must be 16-byte aligned");
    if (f16) hipLaunchKernelGGL(cast_bf16_kernel<true>, dim3(grid1d(n / 8)), dim3(256), 0, (hipStream_t)stream, x, (__hip_bfloat16*)y_bf16, n / 8);
    else hipLaunchKernelGGL(cast_bf16_kernel<false>, dim3(grid1d(n / 8)), dim3(256), 0, (hipStream_t)stream, x, (__hip_bfloat16*)y_bf16, n / 8);
    GN_LAUNCH_CHECK("cast_bf16");
    return 0;
}

int goalnet_cast_f32(const void* x_bf16, float* y, int64_t n, int f16, void* stream) {
    GN_REQUIRE(x_bf16 && y, GOALNET_E_NULL, "cast_f32: null pointer");
    GN_REQUIRE(n > 0 && n % 8 == 0, GOALNET_E_SHAPE, "cast_f32: element count must be a positive multiple of 8");
    GN_REQUIRE(aligned16(x_bf16) && aligned16(y), GOALNET_E_ALIGN, "cast_f32: pointers must be 16-byte aligned");
    if (f16) hipLaunchKernelGGL(cast_f32_kernel<true>, dim3(grid1d(n / 8)), dim3(256), 0, (hipStream_t)stream, (const __hip_bfloat16*)x_bf16, y, n / 8);
    else hipLaunchKernelGGL(cast_f32_kernel<false>, dim3(grid1d(n / 8)), dim3(256), 0, (hipStream_t)stream, (const __hip_bfloat16*)x_bf16, y, n / 8);
    GN_LAUNCH_CHECK("cast_f32");
    return 0;
}

int goalnet_bn_apply_bf16(const float* x, const float* scale, const float* shift, void* y_bf16, int64_t n, int C, int f16, void* stream) {
    GN_REQUIRE(x && scale && shift && y_bf16, GOALNET_E_NULL, "bn_apply_bf16: null pointer");
    GN_REQUIRE(n > 0 && C > 0 && C % 8 == 0 && n % C == 0, GOALNET_E_SHAPE, "bn_apply_bf16: n must be a multiple of C, C of 8");
    GN_REQUIRE(aligned16(x) && aligned16(y_bf16) && aligned16(scale) && aligned16(shift), GOALNET_E_ALIGN, "bn_apply_bf16: alignment");
    if (f16) hipLaunchKernelGGL((bn_apply_bf16_kernel<false, true>), dim3(grid1d(n / 8)), dim3(256), 0, (hipStream_t)stream, (const void*)x, scale, shift,
                                (__hip_bfloat16*)y_bf16, n / 8, C);
    else hipLaunchKernelGGL((bn_apply_bf16_kernel<false, false>), dim3(grid1d(n / 8)), dim3(256), 0, (hipStream_t)stream, (const void*)x, scale, shift,
                            (__hip_bfloat16*)y_bf16, n / 8, C);
    GN_LAUNCH_CHECK("bn_apply_bf16");
    return 0;
}

int goalnet_bn_apply_bf16_p16(const void* x_bf16, const float* scale, const float* shift, void* y_bf16, int64_t n, int C, int f16, void* stream) {
    GN_REQUIRE(x_bf16 && scale && shift && y_bf16, GOALNET_E_NULL, "bn_apply_bf16_p16: null pointer");
    GN_REQUIRE(n > 0 && C > 0 && C % 8 == 0 && n % C == 0, GOALNET_E_SHAPE, "bn_apply_bf16_p16: n must be a multiple of C, C of 8");
    GN_REQUIRE(aligned16(x_bf16) && aligned16(y_bf16) && aligned16(scale) && aligned16(shift), GOALNET_E_ALIGN, "bn_apply_bf16_p16: alignment");
    if (f16) hipLaunchKernelGGL((bn_apply_bf16_kernel<true, true>), dim3(grid1d(n / 8)), dim3(256), 0, (hipStream_t)stream,
                                x_bf16, scale, shift, (__hip_bfloat16*)y_bf16, n / 8, C);
    else hipLaunchKernelGGL((bn_apply_bf16_kernel<true, false>), dim3(grid1d(n / 8)), dim3(256), 0, (hipStream_t)stream,
                            x_bf16, scale, shift, (__hip_bfloat16*)y_bf16, n / 8, C);
    GN_LAUNCH_CHECK("bn_apply_bf16_p16");
    return 0;
}

int goalnet_conv3x3_fwd_bf16(const void* x_bf16, const void* w_bf16, const float* bias, int relu, float* y,
                             int N, int H, int W, int Cin, int Cout, int f16, void* stream) {
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
    return launch_gemm_h<ConvALoaderH, KCLoaderH>("conv3x3_fwd_bf16", ap, bp, ep, M, Cout, 9 * Cin / BKH, 1, 0, f16 != 0, (hipStream_t)stream);
}

static int linear_splits_h(int M, int64_t K, int J) {
    const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((J + BN - 1) / BN);
    return pick_splits(tiles, (int)(K / BKH));
}

// many rows and a long reduction (linear5 at the bench shape): the 256 x 256 phased kernel with XCD-local K-splits
static bool linear_use_256(int M, int64_t K, int J) {
    const char* forced = getenv("GOALNET_BF16_TILE");
    return forced ? forced[0] == '2' : (M >= 512 && J >= 256 && K >= (1ll << 18));
}

size_t goalnet_linear_fwd_bf16_ws_bytes(int M, int64_t K, int J) {
    if (M <= 0 || K <= 0 || J <= 0) return 0;
    const int a = linear_splits_h(M, K, J), b = linear_fwd_splits_256(M, K, J);     // enough for either kernel
    const int s = a > b ? a : b;
    return s > 1 ? (size_t)s * (size_t)M * (size_t)J * sizeof(float) : 0;
}

int goalnet_linear_fwd_bf16(const void* x_bf16, int64_t ldx, const void* w_bf16, const float* bias, int relu,
                            const float* dropmask, int64_t ldmask, float* y, int64_t ldy, float* mult_out, int64_t ldmult,
                            int M, int64_t K, int J, void* ws, size_t ws_bytes, int f16, void* stream) {
    GN_REQUIRE(x_bf16 && w_bf16 && y, GOALNET_E_NULL, "linear_fwd_bf16: null pointer");
    GN_REQUIRE(M > 0 && J > 0 && K > 0 && K < (1ll << 31) - 64, GOALNET_E_SHAPE, "linear_fwd_bf16: bad dims");
    GN_REQUIRE(K % BKH == 0 && J % 4 == 0 && ldx % 8 == 0 && ldy % 4 == 0, GOALNET_E_SHAPE, "linear_fwd_bf16: K %% 64, J %% 4, ldx %% 8");
    GN_REQUIRE(aligned16(x_bf16) && aligned16(w_bf16) && aligned16(y), GOALNET_E_ALIGN, "linear_fwd_bf16: alignment");
    hipStream_t st = (hipStream_t)stream;
    if (linear_use_256(M, K, J)) {
        const int ns = linear_fwd_splits_256(M, K, J);
        if (ns > 1) {
            GN_REQUIRE(ws && aligned16(ws) && ws_bytes >= (size_t)ns * (size_t)M * (size_t)J * sizeof(float), GOALNET_E_WORKSPACE,
                       "linear_fwd_bf16: workspace too small");
            const EpiP efin{(dropmask || mult_out) ? EPI_FULL : EPI_BIAS_RELU, y, ldy, M, J, bias, relu, dropmask, ldmask, mult_out, ldmult, 0};
            const int rc2 = launch_linear_fwd_bf16_256("linear_fwd_bf16(256)", (const __hip_bfloat16*)x_bf16, ldx, (const __hip_bfloat16*)w_bf16,
                                                       M, K, J, (float*)ws, ns, f16 != 0, st);
            if (rc2) return rc2;
            return launch_splitk_reduce("linear_fwd_bf16(256).reduce", (const float*)ws, ns, (int64_t)M * J, efin, st);
        }
    }
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
    const int rc = launch_gemm_h<KCLoaderH, KCLoaderH>("linear_fwd_bf16", ap, bp, ep, M, J, (int)(K / BKH), nsplit, 0, f16 != 0, st);
    if (rc || nsplit == 1) return rc;
    return launch_splitk_reduce("linear_fwd_bf16.reduce", (const float*)ws, nsplit, (int64_t)M * J, efinal, st);
}

/* ---- zero-padded bf16 tensors: [N][H+2][W+2][C] with G = W+3 zero pixels in front of padded pixel 0 and G + 64 behind.
 * The pointer passed around is the address of padded pixel 0; goalnet_bf16_padded_layout gives the allocation. ---- */
int goalnet_bf16_padded_layout(int N, int H, int W, int C, int64_t* total_elems, int64_t* offset_elems) {
    GN_REQUIRE(total_elems && offset_elems, GOALNET_E_NULL, "bf16_padded_layout: null pointer");
    GN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, GOALNET_E_SHAPE, "bf16_padded_layout: non-positive dim");
    const int64_t G = W + 3, Mp = (int64_t)N * (H + 2) * (W + 2);
    *offset_elems = G * C;
    *total_elems = (G + Mp + G + 64) * C;
    return 0;
}

int goalnet_to_bf16_padded(const float* x, const float* scale, const float* shift, void* y_pad, int N, int H, int W, int C, int f16, void* stream) {
    GN_REQUIRE(x && y_pad, GOALNET_E_NULL, "to_bf16_padded: null pointer");
    GN_REQUIRE((scale == nullptr) == (shift == nullptr), GOALNET_E_NULL, "to_bf16_padded: scale/shift must both be set or both NULL");
    GN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, GOALNET_E_SHAPE, "to_bf16_padded: bad dims (C %% 8)");
    GN_REQUIRE(aligned16(x) && aligned16(y_pad) && aligned16(scale) && aligned16(shift), GOALNET_E_ALIGN, "to_bf16_padded: alignment");
    const int64_t n8 = (int64_t)N * H * W * (C / 8);
    if (f16) hipLaunchKernelGGL((to_bf16_padded_kernel<false, true>), dim3(grid1d(n8)), dim3(256), 0, (hipStream_t)stream, (const void*)x, scale, shift,
                                (__hip_bfloat16*)y_pad, n8, H, W, C);
    else hipLaunchKernelGGL((to_bf16_padded_kernel<false, false>), dim3(grid1d(n8)), dim3(256), 0, (hipStream_t)stream, (const void*)x, scale, shift,
                            (__hip_bfloat16*)y_pad, n8, H, W, C);
    GN_LAUNCH_CHECK("to_bf16_padded");
    return 0;
}

int goalnet_to_bf16_padded_p16(const void* x_bf16, const float* scale, const float* shift, void* y_pad, int N, int H, int W, int C, int f16, void* stream) {
    GN_REQUIRE(x_bf16 && y_pad, GOALNET_E_NULL, "to_bf16_padded_p16: null pointer");
    GN_REQUIRE((scale == nullptr) == (shift == nullptr), GOALNET_E_NULL, "to_bf16_padded_p16: scale/shift must both be set or both NULL");
    GN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, GOALNET_E_SHAPE, "to_bf16_padded_p16: bad dims (C %% 8)");
    GN_REQUIRE(aligned16(x_bf16) && aligned16(y_pad) && aligned16(scale) && aligned16(shift), GOALNET_E_ALIGN, "to_bf16_padded_p16: alignment");
    const int64_t n8 = (int64_t)N * H * W * (C / 8);
    if (f16) hipLaunchKernelGGL((to_bf16_padded_kernel<true, true>), dim3(grid1d(n8)), dim3(256), 0, (hipStream_t)stream,
                                x_bf16, scale, shift, (__hip_bfloat16*)y_pad, n8, H, W, C);
    else hipLaunchKernelGGL((to_bf16_padded_kernel<true, false>), dim3(grid1d(n8)), dim3(256), 0, (hipStream_t)stream,
                            x_bf16, scale, shift, (__hip_bfloat16*)y_pad, n8, H, W, C);
    GN_LAUNCH_CHECK("to_bf16_padded_p16");
    return 0;
}

static int conv_splits_h(int64_t M, int Cin, int Cout) {
    return conv_fwd_splits(((M + BM - 1) / BM) * ((Cout + BN - 1) / BN), 9 * Cin / BKH);
}

// large problems: the 256 x 256 phased tile (gemm_bf16_256.hip); GOALNET_BF16_TILE=128 / 256 forces a choice (tests, A/B runs)
static bool conv_use_256(int64_t M, int Cout) {
    const char* forced = getenv("GOALNET_BF16_TILE");
    return forced ? forced[0] == '2' : (Cout >= 256 && M >= 65536);
}

size_t goalnet_conv3x3_fwd_bf16p_ws_bytes(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    const int64_t M = (int64_t)N * H * W;
    const int s = conv_splits_h(M, Cin, Cout);
    return s > 1 ? (size_t)s * (size_t)M * (size_t)Cout * sizeof(float) : 0;
}

int goalnet_conv3x3_fwd_bf16p(const void* x_pad, const void* w_bf16, const float* bias, int relu, float* y,
                              int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, int f16, void* stream) {
    GN_REQUIRE(x_pad && w_bf16 && y, GOALNET_E_NULL, "conv3x3_fwd_bf16p: null pointer");
    GN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, GOALNET_E_SHAPE, "conv3x3_fwd_bf16p: non-positive dim");
    GN_REQUIRE(Cin % BKH == 0 && Cout % 4 == 0, GOALNET_E_SHAPE, "conv3x3_fwd_bf16p: Cin %% 64, Cout %% 4");
    GN_REQUIRE(aligned16(x_pad) && aligned16(w_bf16) && aligned16(y), GOALNET_E_ALIGN, "conv3x3_fwd_bf16p: alignment");
    const int64_t M = (int64_t)N * H * W;
    GN_REQUIRE((int64_t)N * (H + 2) * (W + 2) < (1ll << 31) - 4096, GOALNET_E_SHAPE, "conv3x3_fwd_bf16p: too many pixels");
    hipStream_t st = (hipStream_t)stream;
    ConvAPadLoaderH::P ap{(const __hip_bfloat16*)x_pad, H, W, Cin, M};
    KCLoaderH::P bp{(const __hip_bfloat16*)w_bf16, (int64_t)9 * Cin, Cout};
    const EpiP efinal{EPI_BIAS_RELU, y, Cout, (int)M, Cout, bias, relu, nullptr, 0, nullptr, 0, 0};
    EpiP ep = efinal;
    // large problems: the 256 x 256 phased tile (gemm_bf16_256.hip); GOALNET_BF16_TILE=128 / 256 forces a choice (tests, A/B runs)
    {
        if (conv_use_256(M, Cout)) return launch_conv_bf16_256("conv3x3_fwd_bf16p(256)", (const __hip_bfloat16*)x_pad, H, W, Cin, M,
                                             (const __hip_bfloat16*)w_bf16, Cout, efinal, f16 != 0, st);
    }
    const int nsplit = ws ? conv_splits_h(M, Cin, Cout) : 1;
    if (nsplit > 1) {
        GN_REQUIRE(aligned16(ws) && ws_bytes >= goalnet_conv3x3_fwd_bf16p_ws_bytes(N, H, W, Cin, Cout), GOALNET_E_WORKSPACE,
                   "conv3x3_fwd_bf16p: workspace too small or misaligned");
        ep = EpiP{EPI_RAW, (float*)ws, Cout, (int)M, Cout, nullptr, 0, nullptr, 0, nullptr, 0, M * Cout};
    }
    const int rc = (Cout <= 64 && !getenv("GOALNET_BF16_N64_OFF"))       // 128 x 64 tile (conv2's data gradient: 256 -> 64 channels)
        ? launch_gemm_h<ConvAPadLoaderH, KCLoaderH, true>("conv3x3_fwd_bf16p", ap, bp, ep, M, Cout, 9 * Cin / BKH, nsplit, 0, f16 != 0, st)
        : launch_gemm_h<ConvAPadLoaderH, KCLoaderH>("conv3x3_fwd_bf16p", ap, bp, ep, M, Cout, 9 * Cin / BKH, nsplit, 0, f16 != 0, st);
    if (rc || nsplit == 1) return rc;
    return launch_splitk_reduce("conv3x3_fwd_bf16p.reduce", (const float*)ws, nsplit, M * Cout, efinal, st);
}

}  // extern "C"
template <class AL, class BL, bool F16, bool N64> static const char* gemm_bf16_kernel_name() { return __PRETTY_FUNCTION__; }
extern "C" {

/* which kernel goalnet_conv3x3_fwd_bf16p(_o16) launches for these dims with a bias / ReLU epilogue (forward == 1) or a raw
 * one (the data gradient): the dispatch above, not executed */
const char* goalnet_conv3x3_fwd_bf16p_kernel_name(int N, int H, int W, int Cin, int Cout, int forward) {
    (void)Cin;
    if (conv_use_256((int64_t)N * H * W, Cout)) return conv_bf16_256_kernel_name(forward ? 0 : 1);
    return (Cout <= 64 && !getenv("GOALNET_BF16_N64_OFF")) ? gemm_bf16_kernel_name<ConvAPadLoaderH, KCLoaderH, false, true>()
                                                            : gemm_bf16_kernel_name<ConvAPadLoaderH, KCLoaderH, false, false>();
}

/* 1 when goalnet_conv3x3_fwd_bf16p_o16 serves these dims (the shapes the 256 x 256 tile takes), else 0 */
int goalnet_conv3x3_fwd_bf16p_o16_ok(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % BKH != 0 || Cout % 8 != 0) return 0;
    return conv_use_256((int64_t)N * H * W, Cout) ? 1 : 0;
}

/* y_bf16[N][H][W][Cout] = bf16(act(conv3x3(x_pad, w) + bias)): the fp32 accumulator (+ bias, ReLU) is rounded once, at
 * the store. bias NULL, relu 0: the data-gradient use (w = flipped weights). */
int goalnet_conv3x3_fwd_bf16p_o16(const void* x_pad, const void* w_bf16, const float* bias, int relu, void* y_bf16,
                                  int N, int H, int W, int Cin, int Cout, int f16, void* stream) {
    GN_REQUIRE(x_pad && w_bf16 && y_bf16, GOALNET_E_NULL, "conv3x3_fwd_bf16p_o16: null pointer");
    GN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, GOALNET_E_SHAPE, "conv3x3_fwd_bf16p_o16: non-positive dim");
    GN_REQUIRE(Cin % BKH == 0 && Cout % 8 == 0, GOALNET_E_SHAPE, "conv3x3_fwd_bf16p_o16: Cin %% 64, Cout %% 8");
    GN_REQUIRE(aligned16(x_pad) && aligned16(w_bf16) && aligned16(y_bf16) && aligned16(bias), GOALNET_E_ALIGN, "conv3x3_fwd_bf16p_o16: alignment");
    const int64_t M = (int64_t)N * H * W;
    GN_REQUIRE((int64_t)N * (H + 2) * (W + 2) < (1ll << 31) - 4096, GOALNET_E_SHAPE, "conv3x3_fwd_bf16p_o16: too many pixels");
    GN_REQUIRE(goalnet_conv3x3_fwd_bf16p_o16_ok(N, H, W, Cin, Cout), GOALNET_E_SHAPE,
               "conv3x3_fwd_bf16p_o16: dims not served (ask goalnet_conv3x3_fwd_bf16p_o16_ok; use goalnet_conv3x3_fwd_bf16p)");
    const EpiP ep{EPI_BIAS_RELU, nullptr, Cout, (int)M, Cout, bias, relu, nullptr, 0, nullptr, 0, 0, y_bf16};
    return launch_conv_bf16_256("conv3x3_fwd_bf16p_o16(256)", (const __hip_bfloat16*)x_pad, H, W, Cin, M, (const __hip_bfloat16*)w_bf16,
                                Cout, ep, f16 != 0, (hipStream_t)stream);
}

static int wgrad_splits_h(int64_t Mp, int Cin, int Cout) {
    const int64_t tiles = (int64_t)((Cout + BM - 1) / BM) * ((9 * Cin + BN - 1) / BN);
    const int ktiles = (int)((Mp + BKH - 1) / BKH);
    int64_t s = (2048 + tiles - 1) / tiles;
    const int64_t smax = ktiles / 4 > 1 ? ktiles / 4 : 1;      // >= 4 K-tiles per split (small sub-batches: few pixels)
    if (s > smax) s = smax;
    if (s > 512) s = 512;
    const int kps = (int)((ktiles + s - 1) / s);
    return (ktiles + kps - 1) / kps;
}

// large problems reduce on the 256 x 256 phased tile (gemm_bf16_256.hip); GOALNET_BF16_TILE=128 / 256 forces a choice
static bool wgrad_use_256(int64_t Mp, int Cin, int Cout) {
    const char* forced = getenv("GOALNET_BF16_TILE");
    if (forced) return forced[0] == '2';
    // 256 x 256 tiles must cover the [Cout][9 Cin] output with >= 70 % useful area (conv2: 256 x 576 -> 3 tiles, 75 %;
    // measured 914 TF/s of useful work there against 513 on 128 x 128 tiles)
    const int64_t covered = (int64_t)((Cout + 255) / 256) * ((9 * Cin + 255) / 256) * 65536;
    return Cout >= 256 && Mp >= 262144 && (int64_t)Cout * 9 * Cin * 10 >= covered * 7;
}

size_t goalnet_conv3x3_wgrad_bf16_ws_bytes(int N, int H, int W, int Cin, int Cout) {
    const int64_t Mp = (int64_t)N * (H + 2) * (W + 2);
    const int a = wgrad_splits_h(Mp, Cin, Cout), b = wgrad_splits_256(Mp, Cin, Cout);       // enough for either kernel
    return (size_t)(a > b ? a : b) * (size_t)Cout * 9 * Cin * sizeof(float);
}

/* dw[Cout][3][3][Cin] (fp32) = sum over the padded pixel grid of dy_pad[pm][co] * x_pad[pm + shift(tap)][ci] */
int goalnet_conv3x3_wgrad_bf16(const void* x_pad, const void* dy_pad, float* dw, void* ws, size_t ws_bytes,
                               int N, int H, int W, int Cin, int Cout, int f16, void* stream) {
    GN_REQUIRE(x_pad && dy_pad && dw && ws, GOALNET_E_NULL, "conv3x3_wgrad_bf16: null pointer");
    GN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin % 8 == 0 && Cout % 8 == 0 && Cin > 0 && Cout > 0, GOALNET_E_SHAPE,
               "conv3x3_wgrad_bf16: channels must be positive multiples of 8");
    GN_REQUIRE(aligned16(x_pad) && aligned16(dy_pad) && aligned16(dw) && aligned16(ws), GOALNET_E_ALIGN, "conv3x3_wgrad_bf16: alignment");
    const int64_t Mp = (int64_t)N * (H + 2) * (W + 2);
    GN_REQUIRE(Mp < (1ll << 31) - 4096, GOALNET_E_SHAPE, "conv3x3_wgrad_bf16: too many pixels");
    GN_REQUIRE(ws_bytes >= goalnet_conv3x3_wgrad_bf16_ws_bytes(N, H, W, Cin, Cout), GOALNET_E_WORKSPACE, "conv3x3_wgrad_bf16: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int64_t slab = (int64_t)Cout * 9 * Cin;
    if (wgrad_use_256(Mp, Cin, Cout)) {
        const int ns = wgrad_splits_256(Mp, Cin, Cout);
        const int rc = launch_wgrad_bf16_256("conv3x3_wgrad_bf16(256)", (const __hip_bfloat16*)x_pad, (const __hip_bfloat16*)dy_pad, W + 2,
                                             Cin, Cout, Mp, (float*)ws, ns, f16 != 0, st);
        if (rc) return rc;
        EpiP er{EPI_RAW, dw, (int64_t)9 * Cin, Cout, 9 * Cin, nullptr, 0, nullptr, 0, nullptr, 0, 0};
        return launch_splitk_reduce("conv3x3_wgrad_bf16(256).reduce", (const float*)ws, ns, slab, er, st);
    }
    const int nsplit = wgrad_splits_h(Mp, Cin, Cout);
    const int ktiles = (int)((Mp + BKH - 1) / BKH);
    MCLoaderH::P ap{(const __hip_bfloat16*)dy_pad, Cout, Cout, (int)Mp};
    ConvWgradBLoaderH::P bp{(const __hip_bfloat16*)x_pad, W + 2, Cin, Mp};
    EpiP ep{EPI_RAW, (float*)ws, (int64_t)9 * Cin, Cout, 9 * Cin, nullptr, 0, nullptr, 0, nullptr, 0, slab};
    const int rc = launch_gemm_h<MCLoaderH, ConvWgradBLoaderH>("conv3x3_wgrad_bf16", ap, bp, ep, Cout, 9 * Cin, ktiles, nsplit, 1, f16 != 0, st);
    if (rc) return rc;
    EpiP er{EPI_RAW, dw, (int64_t)9 * Cin, Cout, 9 * Cin, nullptr, 0, nullptr, 0, nullptr, 0, 0};
    return launch_splitk_reduce("conv3x3_wgrad_bf16.reduce", (const float*)ws, nsplit, slab, er, st);
}

/* dx[m][k] = (sum_j dy_bf16[m][j] * w_bf16[j][k]) * mult[m][k]   (fp32 out, mult nullable); J % 64 == 0 */
int goalnet_linear_bwd_dx_bf16(const void* dy_bf16, int64_t lddy, const void* w_bf16, const float* mult, int64_t ldmult,
                               float* dx, int64_t lddx, int M, int64_t K, int J, int f16, void* stream) {
    GN_REQUIRE(dy_bf16 && w_bf16 && dx, GOALNET_E_NULL, "linear_bwd_dx_bf16: null pointer");
    GN_REQUIRE(M > 0 && J > 0 && K > 0 && K < (1ll << 31) - 256, GOALNET_E_SHAPE, "linear_bwd_dx_bf16: bad dims");
    GN_REQUIRE(J % BKH == 0 && K % 8 == 0 && lddy % 8 == 0 && lddx % 4 == 0, GOALNET_E_SHAPE, "linear_bwd_dx_bf16: J %% 64, K %% 8");
    GN_REQUIRE(aligned16(dy_bf16) && aligned16(w_bf16) && aligned16(dx), GOALNET_E_ALIGN, "linear_bwd_dx_bf16: alignment");
    if (!mult && linear_use_256(M, K, J))
        return launch_linear_dx_bf16_256("linear_bwd_dx_bf16(256)", (const __hip_bfloat16*)dy_bf16, lddy, (const __hip_bfloat16*)w_bf16, M, K,
                                         J, dx, nullptr, lddx, f16 != 0, (hipStream_t)stream);
    KCLoaderH::P ap{(const __hip_bfloat16*)dy_bf16, lddy, M};
    MCLoaderH::P bp{(const __hip_bfloat16*)w_bf16, K, (int)K, J};
    EpiP ep{mult ? EPI_MUL : EPI_RAW, dx, lddx, M, (int)K, nullptr, 0, mult, ldmult, nullptr, 0, 0};
    return launch_gemm_h<KCLoaderH, MCLoaderH>("linear_bwd_dx_bf16", ap, bp, ep, M, K, J / BKH, 1, 1, f16 != 0, (hipStream_t)stream);
}

/* 1 when goalnet_linear_bwd_dx_bf16_o16 serves these dims (the shapes the 256 x 256 tile takes), else 0 */
int goalnet_linear_bwd_dx_bf16_o16_ok(int M, int64_t K, int J) {
    return M > 0 && J > 0 && K > 0 && J % BKH == 0 && K % 8 == 0 && linear_use_256(M, K, J) ? 1 : 0;
}

/* dx_bf16[m][k] = bf16(sum_j dy_bf16[m][j] * w_bf16[j][k]): the data gradient rounded once, at the store (fp32 accumulation) */
int goalnet_linear_bwd_dx_bf16_o16(const void* dy_bf16, int64_t lddy, const void* w_bf16, void* dx_bf16, int64_t lddx,
                                   int M, int64_t K, int J, int f16, void* stream) {
    GN_REQUIRE(dy_bf16 && w_bf16 && dx_bf16, GOALNET_E_NULL, "linear_bwd_dx_bf16_o16: null pointer");
    GN_REQUIRE(M > 0 && J > 0 && K > 0 && K < (1ll << 31) - 256, GOALNET_E_SHAPE, "linear_bwd_dx_bf16_o16: bad dims");
    GN_REQUIRE(J % BKH == 0 && K % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0, GOALNET_E_SHAPE, "linear_bwd_dx_bf16_o16: J %% 64, K %% 8, lddx %% 8");
    GN_REQUIRE(aligned16(dy_bf16) && aligned16(w_bf16) && aligned16(dx_bf16), GOALNET_E_ALIGN, "linear_bwd_dx_bf16_o16: alignment");
    GN_REQUIRE(goalnet_linear_bwd_dx_bf16_o16_ok(M, K, J), GOALNET_E_SHAPE,
               "linear_bwd_dx_bf16_o16: dims not served (ask goalnet_linear_bwd_dx_bf16_o16_ok; use goalnet_linear_bwd_dx_bf16)");
    return launch_linear_dx_bf16_256("linear_bwd_dx_bf16_o16(256)", (const __hip_bfloat16*)dy_bf16, lddy, (const __hip_bfloat16*)w_bf16, M, K,
                                     J, nullptr, (__hip_bfloat16*)dx_bf16, lddx, f16 != 0, (hipStream_t)stream);
}

/* dw[j][k] = sum_m dy_bf16[m][j] * x_bf16[m][k]   (fp32 out) */
int goalnet_linear_bwd_dw_bf16(const void* dy_bf16, int64_t lddy, const void* x_bf16, int64_t ldx, float* dw,
                               int M, int64_t K, int J, int f16, void* stream) {
    GN_REQUIRE(dy_bf16 && x_bf16 && dw, GOALNET_E_NULL, "linear_bwd_dw_bf16: null pointer");
    GN_REQUIRE(M > 0 && J > 0 && K > 0 && K < (1ll << 31) - 256, GOALNET_E_SHAPE, "linear_bwd_dw_bf16: bad dims");
    GN_REQUIRE(J % 8 == 0 && K % 8 == 0 && lddy % 8 == 0 && ldx % 8 == 0, GOALNET_E_SHAPE, "linear_bwd_dw_bf16: J, K, lds %% 8");
    GN_REQUIRE(aligned16(dy_bf16) && aligned16(x_bf16) && aligned16(dw), GOALNET_E_ALIGN, "linear_bwd_dw_bf16: alignment");
    {
        // >= 256 output rows, a reduction of at least 4 K-tiles and an output wide enough to fill the chip with 256^2 tiles
        const char* forced = getenv("GOALNET_BF16_TILE");
        const bool big = forced ? forced[0] == '2' : (J >= 256 && M >= 256 && K >= (1 << 18));
        if (big) return launch_linear_dw_bf16_256("linear_bwd_dw_bf16(256)", (const __hip_bfloat16*)dy_bf16, lddy, (const __hip_bfloat16*)x_bf16,
                                                  ldx, M, K, J, dw, f16 != 0, (hipStream_t)stream);
    }
    MCLoaderH::P ap{(const __hip_bfloat16*)dy_bf16, lddy, J, M};
    MCLoaderH::P bp{(const __hip_bfloat16*)x_bf16, ldx, (int)K, M};
    EpiP ep{EPI_RAW, dw, K, J, (int)K, nullptr, 0, nullptr, 0, nullptr, 0, 0};
    // measured at M = 1024, K = 2.5 M: 3.8 ms with one stage and 4 blocks per CU vs 4.7 ms with two stages and 2 (the data
    // gradient, whose W tiles are 256-B pieces 5 MB apart, is the other way round: 10.2 vs 6.6 ms, and keeps two stages)
    return launch_gemm_h_1stage<MCLoaderH, MCLoaderH>("linear_bwd_dw_bf16", ap, bp, ep, J, K, (M + BKH - 1) / BKH, 1, f16 != 0, (hipStream_t)stream);
}

}  // extern "C"
