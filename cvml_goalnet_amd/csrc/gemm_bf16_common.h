// Pieces shared by the bf16 GEMM engines (gemm_bf16.hip: 128 x 128 tile, gemm_bf16_256.hip: 256 x 256 phased tile):
// vector types, the swizzled K-contiguous LDS row image, buffer resources and the LDS-DMA primitive.
#pragma once
#include <hip/hip_bf16.h>

#include "gemm_common.h"

namespace goalnet {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int BKH = 64;                 // bf16 elements per K-tile (128 B per row)
constexpr int ROWB = 128;               // bytes per LDS row
constexpr int OP_BYTES = GEMM_BM * ROWB; // 16 KB: 128 rows of one operand
constexpr unsigned OOB = 0xFFFFFF00u;

__device__ __forceinline__ int kc_boff(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

// n / d for 0 <= n < 2^31, 0 < d < 2^31 with inv = 1.0 / d: (n + 0.5) / d is at least 0.5 / d away from an integer and the
// product carries a relative error of 2^-52, so the truncation is exact. A 64-bit integer division costs ~200 instructions
// per lane on this hardware; the loaders do up to eight per tile (measured: ~4 us of every 256 x 256 tile's set-up).
__device__ __forceinline__ int fast_div(int n, double inv) { return (int)(((double)n + 0.5) * inv); }

__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {          // round-to-nearest-even, a in the low half
    const __hip_bfloat16 x = __float2bfloat16(a), y = __float2bfloat16(b);
    return (unsigned)(*reinterpret_cast<const unsigned short*>(&x)) | ((unsigned)(*reinterpret_cast<const unsigned short*>(&y)) << 16);
}

// ---- the 16-bit storage format of the reduced-precision engine: bf16 (F16 = false) or IEEE fp16 (F16 = true). Both are 16
// bits per element, so every layout, loader, LDS image and DMA path is shared; only the fp32 <-> 16-bit conversions and the
// MFMA opcode (v_mfma_f32_32x32x16_bf16 / _f16, same operand and accumulator layouts, same rate) differ. fp16 carries 11
// significand bits against bf16's 8 (8 x less rounding noise in the forward) but only 5 exponent bits: gradients need the
// loss scale of cvml_goalnet_amd/avm.py.
template <bool F16>
__device__ __forceinline__ unsigned pack2_h16(float a, float b) {           // round-to-nearest-even, a in the low half
    if (F16) {
        const _Float16 x = (_Float16)a, y = (_Float16)b;
        return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
    }
    return pack2_bf16(a, b);
}
template <bool F16> __device__ __forceinline__ float unpack_lo(unsigned u) {
    if (F16) return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu));
    return __uint_as_float(u << 16);
}
template <bool F16> __device__ __forceinline__ float unpack_hi(unsigned u) {
    if (F16) return (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
    return __uint_as_float(u & 0xffff0000u);
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint32_t clamp_u32(int64_t v) {
    return v <= 0 ? 0u : (v > 0xFFFFFF00ll ? 0xFFFFFF00u : (uint32_t)v);
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, char* lds_dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds_dst, 16, (int)voff, (int)soff, 0, 0);
}

// implemented in gemm_bf16_256.hip: conv 3x3 on zero-padded bf16 activations with the 256 x 256 phased tile
// (f16: the operands / the 16-bit output are IEEE fp16 instead of bf16)
int launch_conv_bf16_256(const char* name, const __hip_bfloat16* x_pad, int H, int W, int Cin, int64_t M, const __hip_bfloat16* w,
                         int Cout, const EpiP& ep, bool f16, hipStream_t st);

const char* conv_bf16_256_kernel_name(int role);
int wgrad_splits_256(int64_t Mp, int Cin, int Cout);
int linear_fwd_splits_256(int M, int64_t K, int J);
int launch_linear_fwd_bf16_256(const char* name, const __hip_bfloat16* x, int64_t ldx, const __hip_bfloat16* w, int M, int64_t K,
                               int J, float* slabs, int nsplit, bool f16, hipStream_t st);
int launch_linear_dx_bf16_256(const char* name, const __hip_bfloat16* dy, int64_t lddy, const __hip_bfloat16* w, int M, int64_t K,
                              int J, float* dx, __hip_bfloat16* dx16, int64_t lddx, bool f16, hipStream_t st);
int launch_linear_dw_bf16_256(const char* name, const __hip_bfloat16* dy, int64_t lddy, const __hip_bfloat16* x, int64_t ldx, int M,
                              int64_t K, int J, float* dw, bool f16, hipStream_t st);
int launch_wgrad_bf16_256(const char* name, const __hip_bfloat16* x_pad, const __hip_bfloat16* dy_pad, int Wp2, int Cin, int Cout,
                          int64_t Mp, float* slabs, int nsplit, bool f16, hipStream_t st);

}  // namespace goalnet
