// Shared helpers for the gfx950 kernels of libgoalnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/goalnet_hip.h"

namespace goalnet {

void set_error(const char* fmt, ...);

// one launch check: reports the hipError_t as the (positive) return code, never synchronises
#define GN_LAUNCH_CHECK(name)                                                            \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            goalnet::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
            return (int)e__;                                                             \
        }                                                                                \
    } while (0)

#define GN_REQUIRE(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            goalnet::set_error(__VA_ARGS__);   \
            return (code);                     \
        }                                      \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// bijective XCD-aware remap of a 1-D grid: blocks b and b+8 share an XCD (round-robin dispatch), so give
// each XCD a contiguous chunk of the virtual tile order. Placement only affects speed, never results.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u;
    const unsigned xcd = bid & 7u, slot = bid >> 3;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace goalnet
