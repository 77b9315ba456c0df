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
// precision = "fp16x3" is the same scheme with TWO fp16 parts of the value scaled by a power of two: hi = fp16(v s), mid = fp16(v s - hi)
// (11 + 11 significand bits; s puts the tensor's largest magnitude into [2^14, 2^15), goalnet_absmax -> scale_of_amax) and the THREE
// products hi hi, hi mid, mid hi (the dropped mid mid is 2^-22 |a b|): half the MFMA work of bf16x6. The GEMM epilogues undo the two
// scales with one ldexpf (EpiP::oscale = the exponent from goalnet_split_scales: exact, no intermediate that could overflow); a bias is
// added after that, not carried in the accumulators.
//
// This file: the split passes (activations into the zero-padded layout with the BatchNorm affine applied, weights and linear5's
// operands row by row), the magnitude pass of fp16x3 and the C-ABI entry points. fp32 everywhere else: results, accumulators, bias, ReLU.
#include "gemm_bf16_common.h"

using namespace goalnet;

namespace goalnet {
int launch_conv_split_256(const char* name, int parts, const __hip_bfloat16* x_pads, int H, int W, int Cin, int64_t M,
                          const __hip_bfloat16* ws, int Cout, const EpiP& ep, hipStream_t st);
int launch_conv_split_h64(const char* name, int parts, const __hip_bfloat16* x_pads, int H, int W, int Cin, int64_t M,
                          const __hip_bfloat16* ws, int Cout, const EpiP& ep, hipStream_t st);
int wgrad_split_splits_256(int parts, int64_t Mp, int Cin, int Cout);
int launch_wgrad_split_256(const char* name, int parts, const __hip_bfloat16* x_pads, const __hip_bfloat16* dy_pads, int Wp2, int Cin,
                           int Cout, int64_t Mp, float* slabs, int nsplit, hipStream_t st);
int linear_fwd_split_splits_256(int parts, int M, int64_t K, int J);
int launch_linear_fwd_split_256(const char* name, int parts, const __hip_bfloat16* xs, const __hip_bfloat16* ws, int M, int64_t K, int J,
                                float* slabs, int nsplit, hipStream_t st);
int launch_linear_dx_split_256(const char* name, int parts, const __hip_bfloat16* dys, const __hip_bfloat16* ws, int M, int64_t K, int J,
                               float* dx, int64_t lddx, const int* oscale, hipStream_t st);
int launch_linear_dw_split_256(const char* name, int parts, const __hip_bfloat16* dys, const __hip_bfloat16* xs, int M, int64_t K, int J,
                               float* dw, const int* oscale, hipStream_t st);
}  // namespace goalnet

namespace {

__device__ __forceinline__ float bf16_part(float v, unsigned short& bits) {
    const __hip_bfloat16 h = __float2bfloat16(v);                       // round to nearest even
    bits = *reinterpret_cast<const unsigned short*>(&h);
    return __bfloat162float(h);
}
__device__ __forceinline__ float f16_part(float v, unsigned short& bits) {
    const _Float16 h = (_Float16)v;                                     // round to nearest even; subnormals kept
    bits = __builtin_bit_cast(unsigned short, h);
    return (float)h;
}

// the power of two that puts the largest magnitude of a tensor into [2^14, 2^15) (binary16's largest finite value is 65504):
// amax_bits = bit pattern of max |v| (goalnet_absmax). 1 for an all-zero tensor; clamped for magnitudes below 2^-113.
__device__ __forceinline__ float scale_of_amax(unsigned amax_bits) {
    if (amax_bits == 0u) return 1.f;
    int e = (int)(amax_bits >> 23);
    e = e < 14 ? 14 : e;
    return __uint_as_float((unsigned)(268 - e) << 23);
}

// PARTS = 3: v -> bf16 (hi, mid, lo), v = hi + mid + lo exactly. PARTS = 2: v s -> fp16 (hi, mid): 22 significand bits of the scaled
// value (s a power of two: v s is exact). The subtractions are exact (hi and v share their leading bits).
template <int PARTS>
__device__ __forceinline__ void split_value(float v, float s, unsigned short (&part)[3]) {
    if constexpr (PARTS == 3) {
        const float fh = bf16_part(v, part[0]);
        const float r1 = v - fh;
        const float fm = bf16_part(r1, part[1]);
        bf16_part(r1 - fm, part[2]);
    } else {
        const float vs = v * s;
        const float fh = f16_part(vs, part[0]);
        f16_part(vs - fh, part[1]);
        part[2] = 0;
    }
}

template <int PARTS>
__device__ __forceinline__ void split_store8(const float (&v)[8], float s, __hip_bfloat16* o, int64_t part_stride) {
    unsigned short q[8][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) split_value<PARTS>(v[i], s, q[i]);
#pragma unroll
    for (int p = 0; p < PARTS; ++p)
        *reinterpret_cast<u32x4*>(o + p * part_stride) = u32x4{(unsigned)q[0][p] | ((unsigned)q[1][p] << 16), (unsigned)q[2][p] | ((unsigned)q[3][p] << 16),
                                                             (unsigned)q[4][p] | ((unsigned)q[5][p] << 16), (unsigned)q[6][p] | ((unsigned)q[7][p] << 16)};
}

__device__ __forceinline__ void affine8(float (&v)[8], const float* __restrict__ scale, const float* __restrict__ shift, int ch) {
    const float4 s0 = *reinterpret_cast<const float4*>(scale + ch), s1 = *reinterpret_cast<const float4*>(scale + ch + 4);
    const float4 t0 = *reinterpret_cast<const float4*>(shift + ch), t1 = *reinterpret_cast<const float4*>(shift + ch + 4);
    v[0] = fmaf(v[0], s0.x, t0.x); v[1] = fmaf(v[1], s0.y, t0.y); v[2] = fmaf(v[2], s0.z, t0.z); v[3] = fmaf(v[3], s0.w, t0.w);
    v[4] = fmaf(v[4], s1.x, t1.x); v[5] = fmaf(v[5], s1.y, t1.y); v[6] = fmaf(v[6], s1.z, t1.z); v[7] = fmaf(v[7], s1.w, t1.w);
}

// x fp32 [N][H][W][C] -> [N][H+2][W+2][PARTS C] 16-bit (interior only; the caller zeroed the buffer once), optional per-channel affine
// (the BatchNorm applied in fp32, fmaf as in to_bf16_padded_kernel). One thread = 8 channels of one pixel.
template <int PARTS>
__global__ __launch_bounds__(256) void split_padded_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const unsigned* __restrict__ amax_bits,
                                                          __hip_bfloat16* __restrict__ y, int64_t n8, int H, int W, int C) {
    const int c8n = C >> 3;
    const float s = PARTS == 2 ? scale_of_amax(*amax_bits) : 1.f;
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
        if (scale) affine8(v, scale, shift, c);
        split_store8<PARTS>(v, s, y + pm * PARTS * C + c, C);
    }
}

// x fp32 [rows][C] (row stride ldx) -> [rows][PARTS C] 16-bit (weights: a row = one (output channel, tap); linear5's operands: a row = a
// frame / an output unit). Optional affine per column c with channel c % bnC (BatchNorm3 folded into linear5's input).
// blockIdx.y walks the rows, blockIdx.x / threads the 8-column groups: no 64-bit division per element.
template <int PARTS>
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int bnC, const unsigned* __restrict__ amax_bits,
                                                        __hip_bfloat16* __restrict__ y, int64_t rows, int C) {
    const int c8n = C >> 3;
    const float s = PARTS == 2 ? scale_of_amax(*amax_bits) : 1.f;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const float* xr = x + row * ldx;
        __hip_bfloat16* yr = y + row * PARTS * (int64_t)C;
        for (int g = (int)(blockIdx.x * blockDim.x + threadIdx.x); g < c8n; g += (int)(gridDim.x * blockDim.x)) {
            const int c = g * 8;
            const float4 a = reinterpret_cast<const float4*>(xr + c)[0], b = reinterpret_cast<const float4*>(xr + c)[1];
            float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            if (scale) affine8(v, scale, shift, c % bnC);
            split_store8<PARTS>(v, s, yr + c, C);
        }
    }
}

// max |x[r][c] * scale[c % bnC] + shift[c % bnC]| over a [rows][C] matrix as a bit pattern (non-negative floats order like their
// bits; NaN patterns sort above inf and propagate), atomicMax into *amax_bits (zeroed by the caller). Same fmaf as the split passes.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, int bnC, int64_t rows, int C,
                                                    unsigned* __restrict__ amax_bits) {
    const int c8n = C >> 3;
    unsigned m = 0u;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const float* xr = x + row * ldx;
        for (int g = (int)(blockIdx.x * blockDim.x + threadIdx.x); g < c8n; g += (int)(gridDim.x * blockDim.x)) {
            const int c = g * 8;
            const float4 a = reinterpret_cast<const float4*>(xr + c)[0], b = reinterpret_cast<const float4*>(xr + c)[1];
            float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            if (scale) affine8(v, scale, shift, c % bnC);
#pragma unroll
            for (int k = 0; k < 8; ++k) { const unsigned u = __float_as_uint(v[k]) & 0x7fffffffu; m = u > m ? u : m; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const unsigned o = (unsigned)__shfl_xor((int)m, off, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(amax_bits, m);
}

// *oscale = -(k_a + k_b) with s = 2^k the scales of the two operands: what the GEMM epilogues add to the exponent of their sums
// (EpiP::oscale; ldexpf — the product of the two scales itself can leave fp32's range, e.g. two tensors of magnitude 1e-20)
__device__ __forceinline__ int exp_of_amax(unsigned amax_bits) {
    if (amax_bits == 0u) return 0;
    int e = (int)(amax_bits >> 23);
    e = e < 14 ? 14 : e;
    return 141 - e;                                                    // scale_of_amax = 2^(141 - e)
}
__global__ void split_scales_kernel(const unsigned* __restrict__ amax_a, const unsigned* __restrict__ amax_b, int* __restrict__ oscale) {
    oscale[0] = -(exp_of_amax(*amax_a) + exp_of_amax(*amax_b));
}

unsigned grid1d(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" {

#define GN_PARTS_OK(who) GN_REQUIRE(parts == 2 || parts == 3, GOALNET_E_SHAPE, who ": parts must be 3 (bf16 triples) or 2 (scaled fp16 pairs)")

int goalnet_absmax(const float* x, int64_t ldx, const float* scale, const float* shift, int bnC, int64_t rows, int64_t C,
                   unsigned* amax_bits, void* stream) {
    GN_REQUIRE(x && amax_bits, GOALNET_E_NULL, "absmax: null pointer");
    GN_REQUIRE((scale == nullptr) == (shift == nullptr), GOALNET_E_NULL, "absmax: scale/shift must both be set or both NULL");
    GN_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C < (1ll << 31) - 8 && ldx >= C && ldx % 4 == 0, GOALNET_E_SHAPE, "absmax: bad dims (C %% 8, ldx %% 4)");
    GN_REQUIRE(!scale || (bnC > 0 && bnC % 8 == 0 && C % bnC == 0 && aligned16(scale) && aligned16(shift)), GOALNET_E_SHAPE,
               "absmax: bnC must divide C and be a multiple of 8; aligned scale / shift");
    GN_REQUIRE(aligned16(x), GOALNET_E_ALIGN, "absmax: alignment");
    // contiguous rows are folded so that short, numerous rows (NHWC pixels) still give every block work
    int64_t r = rows, c = C;
    if (ldx == C) { while (c < 8192 && r % 2 == 0) { c *= 2; r /= 2; } }       // column % bnC is still the channel: bnC divides C
    const int64_t gx = (c / 8 + 255) / 256;
    const unsigned bx = (unsigned)(gx > 64 ? 64 : gx);
    const int64_t by64 = r > 8192 / bx ? 8192 / bx : r;
    hipLaunchKernelGGL(absmax_kernel, dim3(bx, (unsigned)(by64 < 1 ? 1 : by64)), dim3(256), 0, (hipStream_t)stream, x, c == C ? ldx : c, scale, shift,
                       bnC, r, (int)c, amax_bits);
    GN_LAUNCH_CHECK("absmax");
    return 0;
}

int goalnet_split_scales(const unsigned* amax_a, const unsigned* amax_b, int* oscale, void* stream) {
    GN_REQUIRE(amax_a && amax_b && oscale, GOALNET_E_NULL, "split_scales: null pointer");
    hipLaunchKernelGGL(split_scales_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, amax_a, amax_b, oscale);
    GN_LAUNCH_CHECK("split_scales");
    return 0;
}

int goalnet_split_padded(int parts, const float* x, const float* scale, const float* shift, const unsigned* amax_bits, void* y_pads,
                         int N, int H, int W, int C, void* stream) {
    GN_PARTS_OK("split_padded");
    GN_REQUIRE(x && y_pads && (parts == 3 || amax_bits), GOALNET_E_NULL, "split_padded: null pointer (parts = 2 needs amax_bits)");
    GN_REQUIRE((scale == nullptr) == (shift == nullptr), GOALNET_E_NULL, "split_padded: scale/shift must both be set or both NULL");
    GN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, GOALNET_E_SHAPE, "split_padded: bad dims (C %% 8)");
    GN_REQUIRE(aligned16(x) && aligned16(y_pads) && aligned16(scale) && aligned16(shift), GOALNET_E_ALIGN, "split_padded: alignment");
    const int64_t n8 = (int64_t)N * H * W * (C / 8);
    if (parts == 3) hipLaunchKernelGGL(split_padded_kernel<3>, dim3(grid1d(n8)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, amax_bits,
                                       (__hip_bfloat16*)y_pads, n8, H, W, C);
    else hipLaunchKernelGGL(split_padded_kernel<2>, dim3(grid1d(n8)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, amax_bits,
                            (__hip_bfloat16*)y_pads, n8, H, W, C);
    GN_LAUNCH_CHECK("split_padded");
    return 0;
}

int goalnet_split_rows(int parts, const float* x, int64_t ldx, const float* scale, const float* shift, int bnC, const unsigned* amax_bits,
                       void* ys, int64_t rows, int64_t C, void* stream) {
    GN_PARTS_OK("split_rows");
    GN_REQUIRE(x && ys && (parts == 3 || amax_bits), GOALNET_E_NULL, "split_rows: null pointer (parts = 2 needs amax_bits)");
    GN_REQUIRE((scale == nullptr) == (shift == nullptr), GOALNET_E_NULL, "split_rows: scale/shift must both be set or both NULL");
    GN_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && C < (1ll << 31) - 8 && ldx >= C && ldx % 4 == 0, GOALNET_E_SHAPE, "split_rows: bad dims (C %% 8, ldx %% 4)");
    GN_REQUIRE(!scale || (bnC > 0 && bnC % 8 == 0 && C % bnC == 0 && aligned16(scale) && aligned16(shift)), GOALNET_E_SHAPE,
               "split_rows: bnC must divide C and be a multiple of 8; aligned scale / shift");
    GN_REQUIRE(aligned16(x) && aligned16(ys), GOALNET_E_ALIGN, "split_rows: alignment");
    const int64_t gx = (C / 8 + 255) / 256;                             // blocks along a row (<= 2048), rows on grid.y
    const unsigned bx = (unsigned)(gx > 2048 ? 2048 : gx);
    const unsigned by = (unsigned)(rows > 65535 ? 65535 : rows);
    if (parts == 3) hipLaunchKernelGGL(split_rows_kernel<3>, dim3(bx, by), dim3(256), 0, (hipStream_t)stream, x, ldx, scale, shift, bnC, amax_bits,
                                       (__hip_bfloat16*)ys, rows, (int)C);
    else hipLaunchKernelGGL(split_rows_kernel<2>, dim3(bx, by), dim3(256), 0, (hipStream_t)stream, x, ldx, scale, shift, bnC, amax_bits,
                            (__hip_bfloat16*)ys, rows, (int)C);
    GN_LAUNCH_CHECK("split_rows");
    return 0;
}

/* y[N][H][W][Cout] (fp32) = act(conv3x3(x, w) + bias) from split operands: x_pads zero-padded [N][H+2][W+2][parts Cin] (W + 3 zero
 * guard pixels in front and behind, as goalnet_conv3x3_fwd_bf16p), ws [Cout][9][parts Cin]. bias nullable; relu 0 / 1. The data
 * gradient is the same call on the split gradient and the split flipped weights. parts = 2: oscale from goalnet_split_scales. */
int goalnet_conv3x3_fwd_split(int parts, const void* x_pads, const void* ws, const float* bias, int relu, float* y,
                              int N, int H, int W, int Cin, int Cout, const int* oscale, void* stream) {
    GN_PARTS_OK("conv3x3_fwd_split");
    GN_REQUIRE(x_pads && ws && y && (parts == 3 || oscale), GOALNET_E_NULL, "conv3x3_fwd_split: null pointer (parts = 2 needs oscale)");
    GN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, GOALNET_E_SHAPE, "conv3x3_fwd_split: non-positive dim");
    GN_REQUIRE(Cin % BKH == 0 && Cout % 4 == 0, GOALNET_E_SHAPE, "conv3x3_fwd_split: Cin %% 64, Cout %% 4");
    GN_REQUIRE(aligned16(x_pads) && aligned16(ws) && aligned16(y), GOALNET_E_ALIGN, "conv3x3_fwd_split: alignment");
    const int64_t M = (int64_t)N * H * W;
    GN_REQUIRE((int64_t)N * (H + 2) * (W + 2) < (1ll << 31) - 4096, GOALNET_E_SHAPE, "conv3x3_fwd_split: too many pixels");
    // byte offsets inside one tile's window of the padded tensor are 32-bit (ConvAPadLoader256)
    GN_REQUIRE((int64_t)(256 + 4 * (W + 2) + 2 * (int64_t)(H + 2) * (W + 2)) * parts * Cin * 2 < (1ll << 32) - 4096, GOALNET_E_SHAPE,
               "conv3x3_fwd_split: frame too large for one tile window");
    EpiP ep{EPI_BIAS_RELU, y, Cout, (int)M, Cout, bias, relu, nullptr, 0, nullptr, 0, 0};
    ep.oscale = parts == 2 ? oscale : nullptr;
    // <= 64 output channels (a data gradient into 64 channels): the 128 x 64 tile; a 256-wide tile would be three quarters empty
    if (Cout <= 64) return launch_conv_split_h64("conv3x3_fwd_split(128x64)", parts, (const __hip_bfloat16*)x_pads, H, W, Cin, M,
                                                 (const __hip_bfloat16*)ws, Cout, ep, (hipStream_t)stream);
    return launch_conv_split_256("conv3x3_fwd_split", parts, (const __hip_bfloat16*)x_pads, H, W, Cin, M, (const __hip_bfloat16*)ws, Cout, ep,
                                 (hipStream_t)stream);
}

size_t goalnet_conv3x3_wgrad_split_ws_bytes(int parts, int N, int H, int W, int Cin, int Cout) {
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (parts != 2 && parts != 3)) return 0;
    const int64_t Mp = (int64_t)N * (H + 2) * (W + 2);
    return (size_t)wgrad_split_splits_256(parts, Mp, Cin, Cout) * (size_t)Cout * 9 * Cin * sizeof(float);
}

/* dw[Cout][3][3][Cin] (fp32) = sum over the padded pixel grid of dy[pm][co] * x[pm + shift(tap)][ci], both operands split:
 * x_pads [padded pixels][parts Cin], dy_pads [padded pixels][parts Cout] (zero borders and guards as for goalnet_conv3x3_wgrad_bf16) */
int goalnet_conv3x3_wgrad_split(int parts, const void* x_pads, const void* dy_pads, float* dw, void* ws, size_t ws_bytes,
                                int N, int H, int W, int Cin, int Cout, const int* oscale, void* stream) {
    GN_PARTS_OK("conv3x3_wgrad_split");
    GN_REQUIRE(x_pads && dy_pads && dw && ws && (parts == 3 || oscale), GOALNET_E_NULL, "conv3x3_wgrad_split: null pointer (parts = 2 needs oscale)");
    GN_REQUIRE(N > 0 && H > 0 && W > 0 && Cin % 8 == 0 && Cout % 8 == 0 && Cin > 0 && Cout > 0, GOALNET_E_SHAPE,
               "conv3x3_wgrad_split: channels must be positive multiples of 8");
    GN_REQUIRE(aligned16(x_pads) && aligned16(dy_pads) && aligned16(dw) && aligned16(ws), GOALNET_E_ALIGN, "conv3x3_wgrad_split: alignment");
    const int64_t Mp = (int64_t)N * (H + 2) * (W + 2);
    GN_REQUIRE(Mp < (1ll << 31) - 4096, GOALNET_E_SHAPE, "conv3x3_wgrad_split: too many pixels");
    GN_REQUIRE(ws_bytes >= goalnet_conv3x3_wgrad_split_ws_bytes(parts, N, H, W, Cin, Cout), GOALNET_E_WORKSPACE, "conv3x3_wgrad_split: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int64_t slab = (int64_t)Cout * 9 * Cin;
    const int ns = wgrad_split_splits_256(parts, Mp, Cin, Cout);
    const int rc = launch_wgrad_split_256("conv3x3_wgrad_split", parts, (const __hip_bfloat16*)x_pads, (const __hip_bfloat16*)dy_pads, W + 2, Cin, Cout,
                                          Mp, (float*)ws, ns, st);
    if (rc) return rc;
    EpiP er{EPI_RAW, dw, (int64_t)9 * Cin, Cout, 9 * Cin, nullptr, 0, nullptr, 0, nullptr, 0, 0};
    er.oscale = parts == 2 ? oscale : nullptr;
    return launch_splitk_reduce("conv3x3_wgrad_split.reduce", (const float*)ws, ns, slab, er, st);
}

/* ---- linear5 (/root/reference/utils.py:166-170, 189-193) on split operands; parts side by side along the row ------------------------
 * y[m][j] = dropmask * act(sum_k x[m][k] w[j][k] + bias[j]) with xs [M][parts K], ws [J][parts K] (goalnet_split_rows); epilogue fields
 * as goalnet_linear_fwd_bf16. Served by the 256 x 256 tile only: goalnet_linear_split_ok says whether the dims are. */
int goalnet_linear_split_ok(int parts, int M, int64_t K, int J) {
    return (parts == 2 || parts == 3) && M >= 256 && J >= 256 && J % BKH == 0 && K % BKH == 0 && K >= (1ll << 16) &&
           parts * K * 2 * 256 < (1ll << 32) - 65536 ? 1 : 0;           // 32-bit byte offsets inside a 256-row tile
}

size_t goalnet_linear_fwd_split_ws_bytes(int parts, int M, int64_t K, int J) {
    if (!goalnet_linear_split_ok(parts, M, K, J)) return 0;
    return (size_t)linear_fwd_split_splits_256(parts, M, K, J) * (size_t)M * (size_t)J * sizeof(float);
}

int goalnet_linear_fwd_split(int parts, const void* xs, const void* wsp, const float* bias, int relu, const float* dropmask, int64_t ldmask,
                             float* y, int64_t ldy, float* mult_out, int64_t ldmult, int M, int64_t K, int J, void* ws, size_t ws_bytes,
                             const int* oscale, void* stream) {
    GN_REQUIRE(xs && wsp && y && ws && (parts == 3 || oscale), GOALNET_E_NULL, "linear_fwd_split: null pointer (parts = 2 needs oscale)");
    GN_REQUIRE(goalnet_linear_split_ok(parts, M, K, J) && ldy % 4 == 0, GOALNET_E_SHAPE, "linear_fwd_split: dims not served (goalnet_linear_split_ok)");
    GN_REQUIRE(aligned16(xs) && aligned16(wsp) && aligned16(y) && aligned16(ws), GOALNET_E_ALIGN, "linear_fwd_split: alignment");
    GN_REQUIRE(ws_bytes >= goalnet_linear_fwd_split_ws_bytes(parts, M, K, J), GOALNET_E_WORKSPACE, "linear_fwd_split: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int ns = linear_fwd_split_splits_256(parts, M, K, J);
    EpiP efin{(dropmask || mult_out) ? EPI_FULL : EPI_BIAS_RELU, y, ldy, M, J, bias, relu, dropmask, ldmask, mult_out, ldmult, 0};
    efin.oscale = parts == 2 ? oscale : nullptr;
    const int rc = launch_linear_fwd_split_256("linear_fwd_split", parts, (const __hip_bfloat16*)xs, (const __hip_bfloat16*)wsp, M, K, J, (float*)ws, ns, st);
    if (rc) return rc;
    return launch_splitk_reduce("linear_fwd_split.reduce", (const float*)ws, ns, (int64_t)M * J, efin, st);
}

/* dx[m][k] (fp32) = sum_j dy[m][j] w[j][k] from dys [M][parts J] and ws [J][parts K] */
int goalnet_linear_bwd_dx_split(int parts, const void* dys, const void* wsp, float* dx, int64_t lddx, int M, int64_t K, int J,
                                const int* oscale, void* stream) {
    GN_REQUIRE(dys && wsp && dx && (parts == 3 || oscale), GOALNET_E_NULL, "linear_bwd_dx_split: null pointer (parts = 2 needs oscale)");
    GN_REQUIRE(goalnet_linear_split_ok(parts, M, K, J) && lddx % 4 == 0 && K < (1ll << 31) - 256, GOALNET_E_SHAPE, "linear_bwd_dx_split: dims not served");
    GN_REQUIRE(aligned16(dys) && aligned16(wsp) && aligned16(dx), GOALNET_E_ALIGN, "linear_bwd_dx_split: alignment");
    return launch_linear_dx_split_256("linear_bwd_dx_split", parts, (const __hip_bfloat16*)dys, (const __hip_bfloat16*)wsp, M, K, J, dx, lddx,
                                      parts == 2 ? oscale : nullptr, (hipStream_t)stream);
}

/* dw[j][k] (fp32) = sum_m dy[m][j] x[m][k] from dys [M][parts J] and xs [M][parts K] */
int goalnet_linear_bwd_dw_split(int parts, const void* dys, const void* xs, float* dw, int M, int64_t K, int J, const int* oscale, void* stream) {
    GN_REQUIRE(dys && xs && dw && (parts == 3 || oscale), GOALNET_E_NULL, "linear_bwd_dw_split: null pointer (parts = 2 needs oscale)");
    GN_REQUIRE(goalnet_linear_split_ok(parts, M, K, J) && K < (1ll << 31) - 256, GOALNET_E_SHAPE, "linear_bwd_dw_split: dims not served");
    GN_REQUIRE(aligned16(dys) && aligned16(xs) && aligned16(dw), GOALNET_E_ALIGN, "linear_bwd_dw_split: alignment");
    return launch_linear_dw_split_256("linear_bwd_dw_split", parts, (const __hip_bfloat16*)dys, (const __hip_bfloat16*)xs, M, K, J, dw,
                                      parts == 2 ? oscale : nullptr, (hipStream_t)stream);
}

}  // extern "C"
