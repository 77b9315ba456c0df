// precision = "bf16x6": the 3 x 3 convolutions of VisBl blocks 2 and 3 (/root/reference/utils.py:156-164, 179-187) with fp32-grade
// products on the 16-bit MFMA. An fp32 operand value v is stored as THREE bf16 values
//         hi = bf16(v),   mid = bf16(v - hi),   lo = bf16(v - hi - mid)            (v = hi + mid + lo exactly: 3 x 8 significand bits)
// side by side along the channel axis ([hi | mid | lo], 3 C stored channels), and a b = sum of the six largest of the nine partial
// products (hi hi, hi mid, mid hi, hi lo, lo hi, mid mid; the three dropped ones are < 2^-23 |a b|). Each partial product of two
// bf16 values is exact in the MFMA's fp32 accumulator, so what separates the result from an fp32-MFMA convolution is the order
// and number of fp32 additions (6 x as many): measured 1.4e-6 of the output scale against 4e-7 at K = 2304 (tests). The six terms
// are six K-segments of ONE launch of the 256 x 256 phased kernel (gemm_bf16_256.hip, SEGMAP_A / SEGMAP_B): 6 x the flops of a
// bf16 convolution at the 16-bit MFMA rate (2.5 PF dense) against the fp32 MFMA's 157 TF/s.
//
// This file: the two split passes (activations into the zero-padded layout with the BatchNorm affine applied, weights row by
// row) and the C-ABI entry points. fp32 everywhere else: results, accumulators, bias, ReLU.
#include "gemm_bf16_common.h"

using namespace goalnet;

namespace goalnet {
int launch_conv_x6_256(const char* name, const __hip_bfloat16* x_pad3, int H, int W, int Cin, int64_t M, const __hip_bfloat16* w3,
                       int Cout, const EpiP& ep, hipStream_t st);
int wgrad_x6_splits_256(int64_t Mp, int Cin, int Cout);
int launch_wgrad_x6_256(const char* name, const __hip_bfloat16* x_pad3, const __hip_bfloat16* dy_pad3, int Wp2, int Cin, int Cout,
                        int64_t Mp, float* slabs, int nsplit, hipStream_t st);
int linear_fwd_x6_splits_256(int M, int64_t K, int J);
int launch_linear_fwd_x6_256(const char* name, const __hip_bfloat16* x3s, const __hip_bfloat16* w3s, int M, int64_t K, int J, float* slabs,
                             int nsplit, hipStream_t st);
int launch_linear_dx_x6_256(const char* name, const __hip_bfloat16* dy3s, const __hip_bfloat16* w3s, int M, int64_t K, int J, float* dx,
                            int64_t lddx, hipStream_t st);
int launch_linear_dw_x6_256(const char* name, const __hip_bfloat16* dy3s, const __hip_bfloat16* x3s, int M, int64_t K, int J, float* dw,
                            hipStream_t st);
}  // namespace goalnet

namespace {

__device__ __forceinline__ float bf16_part(float v, unsigned short& bits) {
    const __hip_bfloat16 h = __float2bfloat16(v);                       // round to nearest even
    bits = *reinterpret_cast<const unsigned short*>(&h);
    return __bfloat162float(h);
}

// v -> (hi, mid, lo); the subtractions are exact (Sterbenz-like: hi and v share their leading bits)
__device__ __forceinline__ void split3(float v, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    const float fh = bf16_part(v, hi);
    const float r1 = v - fh;
    const float fm = bf16_part(r1, mid);
    const float r2 = r1 - fm;
    bf16_part(r2, lo);
}

__device__ __forceinline__ void split3x8(const float (&v)[8], u32x4& h, u32x4& m, u32x4& l) {
    unsigned short hs[8], ms[8], ls[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) split3(v[i], hs[i], ms[i], ls[i]);
    h = u32x4{(unsigned)hs[0] | ((unsigned)hs[1] << 16), (unsigned)hs[2] | ((unsigned)hs[3] << 16), (unsigned)hs[4] | ((unsigned)hs[5] << 16), (unsigned)hs[6] | ((unsigned)hs[7] << 16)};
    m = u32x4{(unsigned)ms[0] | ((unsigned)ms[1] << 16), (unsigned)ms[2] | ((unsigned)ms[3] << 16), (unsigned)ms[4] | ((unsigned)ms[5] << 16), (unsigned)ms[6] | ((unsigned)ms[7] << 16)};
    l = u32x4{(unsigned)ls[0] | ((unsigned)ls[1] << 16), (unsigned)ls[2] | ((unsigned)ls[3] << 16), (unsigned)ls[4] | ((unsigned)ls[5] << 16), (unsigned)ls[6] | ((unsigned)ls[7] << 16)};
}

// x fp32 [N][H][W][C] -> [N][H+2][W+2][3 C] bf16 (interior only; the caller zeroed the buffer once), optional per-channel affine
// (the BatchNorm applied in fp32, fmaf as in to_bf16_padded_kernel). One thread = 8 channels of one pixel: 32 B in, 3 x 16 B out.
__global__ __launch_bounds__(256) void split3_padded_kernel(const float* __restrict__ x, const float* __restrict__ scale,
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
        const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        if (scale) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaf(v[k], scale[c + k], shift[c + k]);
        }
        u32x4 hh, mm, ll;
        split3x8(v, hh, mm, ll);
        __hip_bfloat16* o = y + pm * 3 * C + c;
        *reinterpret_cast<u32x4*>(o) = hh;
        *reinterpret_cast<u32x4*>(o + C) = mm;
        *reinterpret_cast<u32x4*>(o + 2 * C) = ll;
    }
}

// x fp32 [rows][C] (row stride ldx) -> [rows][3 C] bf16 (weights: a row = one (output channel, tap); linear5's operands: a row = a
// frame / an output unit). Optional affine per column c with channel c % bnC (BatchNorm3 folded into linear5's input).
__global__ __launch_bounds__(256) void split3_rows_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int bnC, __hip_bfloat16* __restrict__ y,
                                                         int64_t rows, int C) {
    // blockIdx.y walks the rows, blockIdx.x / threads the 8-column groups: no 64-bit division per element
    const int c8n = C >> 3;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const float* xr = x + row * ldx;
        __hip_bfloat16* yr = y + row * 3 * (int64_t)C;
        for (int g = (int)(blockIdx.x * blockDim.x + threadIdx.x); g < c8n; g += (int)(gridDim.x * blockDim.x)) {
            const int c = g * 8;
            const float4 a = reinterpret_cast<const float4*>(xr + c)[0], b = reinterpret_cast<const float4*>(xr + c)[1];
            float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            if (scale) {
                const int ch = c % bnC;
                const float4 s0 = *reinterpret_cast<const float4*>(scale + ch), s1 = *reinterpret_cast<const float4*>(scale + ch + 4);
                const float4 t0 = *reinterpret_cast<const float4*>(shift + ch), t1 = *reinterpret_cast<const float4*>(shift + ch + 4);
                v[0] = fmaf(v[0], s0.x, t0.x); v[1] = fmaf(v[1], s0.y, t0.y); v[2] = fmaf(v[2], s0.z, t0.z); v[3] = fmaf(v[3], s0.w, t0.w);
                v[4] = fmaf(v[4], s1.x, t1.x); v[5] = fmaf(v[5], s1.y, t1.y); v[6] = fmaf(v[6], s1.z, t1.z); v[7] = fmaf(v[7], s1.w, t1.w);
            }
            u32x4 hh, mm, ll;
            split3x8(v, hh, mm, ll);
            *reinterpret_cast<u32x4*>(yr + c) = hh;
            *reinterpret_cast<u32x4*>(yr + C + c) = mm;
            *reinterpret_cast<u32x4*>(yr + 2 * (int64_t)C + c) = ll;
        }
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

int goalnet_split3_padded(const float* x, const float* scale, const float* shift, void* y_pad3, int N, int H, int W, int C, void* stream) {
    GN_REQUIRE(x && y_pad3, GOALNET_E_NULL, "split3_padded: null pointer");
    GN_REQUIRE((scale == nullptr) == (shift == nullptr), GOALNET_E_NULL, "split3_padded: scale/shift must both be set or both NULL");
    GN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, GOALNET_E_SHAPE, "split3_padded: bad dims (C %% 8)");
    GN_REQUIRE(aligned16(x) && aligned16(y_pad3), GOALNET_E_ALIGN, "split3_padded: alignment");
    const int64_t n8 = (int64_t)N * H * W * (C / 8);
    hipLaunchKernelGGL(split3_padded_kernel, dim3(grid1d(n8)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, (__hip_bfloat16*)y_pad3, n8, H, W, C);
    GN_LAUNCH_CHECK("split3_padded");
    return 0;
}

int goalnet_split3_rows(const float* x, int64_t ldx, const float* scale, const float* shift, int bnC, void* y3, int64_t rows, int64_t C,
                        void* stream) {
    GN_REQUIRE(x && y3, GOALNET_E_NULL, "split3_rows: null pointer");
    GN_REQUIRE((scale == nullptr) == (shift == nullptr), GOALNET_E_NULL, "split3_rows: scale/shift must both be set or both NULL");
    GN_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && ldx >= C && ldx % 4 == 0, GOALNET_E_SHAPE, "split3_rows: bad dims (C %% 8, ldx %% 4)");
    GN_REQUIRE(!scale || (bnC > 0 && bnC % 8 == 0 && C % bnC == 0), GOALNET_E_SHAPE, "split3_rows: bnC must divide C and be a multiple of 8");
    GN_REQUIRE(aligned16(x) && aligned16(y3), GOALNET_E_ALIGN, "split3_rows: alignment");
    GN_REQUIRE(C < (1ll << 31) - 8 && (!scale || (aligned16(scale) && aligned16(shift))), GOALNET_E_SHAPE, "split3_rows: C < 2^31, aligned scale / shift");
    const int64_t gx = (C / 8 + 255) / 256;                             // blocks along a row (<= 2048), rows on grid.y
    const unsigned bx = (unsigned)(gx > 2048 ? 2048 : gx);
    const unsigned by = (unsigned)(rows > 65535 ? 65535 : rows);
    hipLaunchKernelGGL(split3_rows_kernel, dim3(bx, by), dim3(256), 0, (hipStream_t)stream, x, ldx, scale, shift, bnC, (__hip_bfloat16*)y3, rows, (int)C);
    GN_LAUNCH_CHECK("split3_rows");
    return 0;
}

/* y[N][H][W][Cout] (fp32) = act(conv3x3(x, w) + bias) from split operands: x_pad3 zero-padded [N][H+2][W+2][3 Cin] (W + 3 zero
 * guard pixels in front and behind, as goalnet_conv3x3_fwd_bf16p), w3 [Cout][9][3 Cin]. bias nullable; relu 0 / 1. The data
 * gradient is the same call on the split gradient and the split flipped weights. */
int goalnet_conv3x3_fwd_x6(const void* x_pad3, const void* w3, const float* bias, int relu, float* y,
                           int N, int H, int W, int Cin, int Cout, void* stream) {
    GN_REQUIRE(x_pad3 && w3 && y, GOALNET_E_NULL, "conv3x3_fwd_x6: null pointer");
    GN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, GOALNET_E_SHAPE, "conv3x3_fwd_x6: non-positive dim");
    GN_REQUIRE(Cin % BKH == 0 && Cout % 4 == 0, GOALNET_E_SHAPE, "conv3x3_fwd_x6: Cin %% 64, Cout %% 4");
    GN_REQUIRE(aligned16(x_pad3) && aligned16(w3) && aligned16(y), GOALNET_E_ALIGN, "conv3x3_fwd_x6: alignment");
    const int64_t M = (int64_t)N * H * W;
    GN_REQUIRE((int64_t)N * (H + 2) * (W + 2) < (1ll << 31) - 4096, GOALNET_E_SHAPE, "conv3x3_fwd_x6: too many pixels");
    // byte offsets inside one tile's window of the padded tensor are 32-bit (ConvAPadLoader256)
    GN_REQUIRE((int64_t)(256 + 4 * (W + 2) + 2 * (int64_t)(H + 2) * (W + 2)) * 3 * Cin * 2 < (1ll << 32) - 4096, GOALNET_E_SHAPE,
               "conv3x3_fwd_x6: frame too large for one tile window");
    const EpiP ep{EPI_BIAS_RELU, y, Cout, (int)M, Cout, bias, relu, nullptr, 0, nullptr, 0, 0};
    return launch_conv_x6_256("conv3x3_fwd_x6", (const __hip_bfloat16*)x_pad3, H, W, Cin, M, (const __hip_bfloat16*)w3, Cout, ep, (hipStream_t)stream);
}

size_t goalnet_conv3x3_wgrad_x6_ws_bytes(int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    const int64_t Mp = (int64_t)N * (H + 2) * (W + 2);
    return (size_t)wgrad_x6_splits_256(Mp, Cin, Cout) * (size_t)Cout * 9 * Cin * sizeof(float);
}

/* dw[Cout][3][3][Cin] (fp32) = sum over the padded pixel grid of dy[pm][co] * x[pm + shift(tap)][ci], both operands split:
 * x_pad3 [padded pixels][3 Cin], dy_pad3 [padded pixels][3 Cout] (zero borders and guards as for goalnet_conv3x3_wgrad_bf16) */
int goalnet_conv3x3_wgrad_x6(const void* x_pad3, const void* dy_pad3, float* dw, void* ws, size_t ws_bytes,
                             int N, int H, int W, int Cin, int Cout, void* stream) {
    GN_REQUIRE(x_pad3 && dy_pad3 && dw && ws, GOALNET_E_NULL, "conv3x3_wgrad_x6: null pointer");
    GN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin % 8 == 0 && Cout % 8 == 0 && Cin > 0 && Cout > 0, GOALNET_E_SHAPE,
               "conv3x3_wgrad_x6: channels must be positive multiples of 8");
    GN_REQUIRE(aligned16(x_pad3) && aligned16(dy_pad3) && aligned16(dw) && aligned16(ws), GOALNET_E_ALIGN, "conv3x3_wgrad_x6: alignment");
    const int64_t Mp = (int64_t)N * (H + 2) * (W + 2);
    GN_REQUIRE(Mp < (1ll << 31) - 4096, GOALNET_E_SHAPE, "conv3x3_wgrad_x6: too many pixels");
    GN_REQUIRE(ws_bytes >= goalnet_conv3x3_wgrad_x6_ws_bytes(N, H, W, Cin, Cout), GOALNET_E_WORKSPACE, "conv3x3_wgrad_x6: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int64_t slab = (int64_t)Cout * 9 * Cin;
    const int ns = wgrad_x6_splits_256(Mp, Cin, Cout);
    const int rc = launch_wgrad_x6_256("conv3x3_wgrad_x6", (const __hip_bfloat16*)x_pad3, (const __hip_bfloat16*)dy_pad3, W + 2, Cin, Cout, Mp,
                                       (float*)ws, ns, st);
    if (rc) return rc;
    EpiP er{EPI_RAW, dw, (int64_t)9 * Cin, Cout, 9 * Cin, nullptr, 0, nullptr, 0, nullptr, 0, 0};
    return launch_splitk_reduce("conv3x3_wgrad_x6.reduce", (const float*)ws, ns, slab, er, st);
}

/* ---- linear5 (/root/reference/utils.py:166-170, 189-193) on split operands; rows [hi | mid | lo] side by side ----------------------
 * y[m][j] = dropmask * act(sum_k x[m][k] w[j][k] + bias[j]) with x3s [M][3 K], w3s [J][3 K] (goalnet_split3_rows); epilogue fields as
 * goalnet_linear_fwd_bf16. Served by the 256 x 256 tile only: goalnet_linear_x6_ok says whether the dims are. */
int goalnet_linear_x6_ok(int M, int64_t K, int J) {
    return M >= 256 && J >= 256 && J % BKH == 0 && K % BKH == 0 && K >= (1ll << 16) && 3 * K * 2 * 256 < (1ll << 32) - 65536 ? 1 : 0;   // 32-bit byte offsets inside a 256-row tile
}

size_t goalnet_linear_fwd_x6_ws_bytes(int M, int64_t K, int J) {
    if (!goalnet_linear_x6_ok(M, K, J)) return 0;
    return (size_t)linear_fwd_x6_splits_256(M, K, J) * (size_t)M * (size_t)J * sizeof(float);
}

int goalnet_linear_fwd_x6(const void* x3s, const void* w3s, const float* bias, int relu, const float* dropmask, int64_t ldmask,
                          float* y, int64_t ldy, float* mult_out, int64_t ldmult, int M, int64_t K, int J, void* ws, size_t ws_bytes,
                          void* stream) {
    GN_REQUIRE(x3s && w3s && y && ws, GOALNET_E_NULL, "linear_fwd_x6: null pointer");
    GN_REQUIRE(goalnet_linear_x6_ok(M, K, J) && ldy % 4 == 0, GOALNET_E_SHAPE, "linear_fwd_x6: dims not served (goalnet_linear_x6_ok)");
    GN_REQUIRE(aligned16(x3s) && aligned16(w3s) && aligned16(y) && aligned16(ws), GOALNET_E_ALIGN, "linear_fwd_x6: alignment");
    GN_REQUIRE(ws_bytes >= goalnet_linear_fwd_x6_ws_bytes(M, K, J), GOALNET_E_WORKSPACE, "linear_fwd_x6: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int ns = linear_fwd_x6_splits_256(M, K, J);
    const EpiP efin{(dropmask || mult_out) ? EPI_FULL : EPI_BIAS_RELU, y, ldy, M, J, bias, relu, dropmask, ldmask, mult_out, ldmult, 0};
    const int rc = launch_linear_fwd_x6_256("linear_fwd_x6", (const __hip_bfloat16*)x3s, (const __hip_bfloat16*)w3s, M, K, J, (float*)ws, ns, st);
    if (rc) return rc;
    return launch_splitk_reduce("linear_fwd_x6.reduce", (const float*)ws, ns, (int64_t)M * J, efin, st);
}

/* dx[m][k] (fp32) = sum_j dy[m][j] w[j][k] from dy3s [M][3 J] and w3s [J][3 K] */
int goalnet_linear_bwd_dx_x6(const void* dy3s, const void* w3s, float* dx, int64_t lddx, int M, int64_t K, int J, void* stream) {
    GN_REQUIRE(dy3s && w3s && dx, GOALNET_E_NULL, "linear_bwd_dx_x6: null pointer");
    GN_REQUIRE(goalnet_linear_x6_ok(M, K, J) && lddx % 4 == 0 && K < (1ll << 31) - 256, GOALNET_E_SHAPE, "linear_bwd_dx_x6: dims not served");
    GN_REQUIRE(aligned16(dy3s) && aligned16(w3s) && aligned16(dx), GOALNET_E_ALIGN, "linear_bwd_dx_x6: alignment");
    return launch_linear_dx_x6_256("linear_bwd_dx_x6", (const __hip_bfloat16*)dy3s, (const __hip_bfloat16*)w3s, M, K, J, dx, lddx, (hipStream_t)stream);
}

/* dw[j][k] (fp32) = sum_m dy[m][j] x[m][k] from dy3s [M][3 J] and x3s [M][3 K] */
int goalnet_linear_bwd_dw_x6(const void* dy3s, const void* x3s, float* dw, int M, int64_t K, int J, void* stream) {
    GN_REQUIRE(dy3s && x3s && dw, GOALNET_E_NULL, "linear_bwd_dw_x6: null pointer");
    GN_REQUIRE(goalnet_linear_x6_ok(M, K, J) && K < (1ll << 31) - 256, GOALNET_E_SHAPE, "linear_bwd_dw_x6: dims not served");
    GN_REQUIRE(aligned16(dy3s) && aligned16(x3s) && aligned16(dw), GOALNET_E_ALIGN, "linear_bwd_dw_x6: alignment");
    return launch_linear_dw_x6_256("linear_bwd_dw_x6", (const __hip_bfloat16*)dy3s, (const __hip_bfloat16*)x3s, M, K, J, dw, (hipStream_t)stream);
}

}  // extern "C"
