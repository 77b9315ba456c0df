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

// ---- inter-block hand-over inside ONE launch (last-block finalisers, grid barriers) without flushing the L2 ----------------
// The eight XCDs' L2s are not coherent with each other, so `__threadfence()` (an agent-scope release + acquire) compiles to
// `buffer_wbl2 sc1` + `buffer_inv sc1`: a write-back and invalidation of the WHOLE L2 — measured ~30 us per use in the 10-frame
// step, where a whole kernel otherwise takes 5. Data that another block of the same launch must see is instead written and read
// with relaxed agent-scope atomics (`global_store / global_load ... sc1`: write-through / re-fetch of just those lines, coherent
// by the memory model's definition), the writers wait for their stores to complete (vmcnt(0)) before the block's ticket / arrival
// atomic goes out, and the readers issue their loads after they have seen it. No cache-wide operation anywhere.
__device__ __forceinline__ void st_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_dev(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_dev(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every thread of the block: my st_dev stores are complete; then the block meets (s_barrier)
__device__ __forceinline__ void dev_stores_done_block() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
}
// after the block has seen the other blocks' ticket / arrival: loads below this point stay below it
__device__ __forceinline__ void dev_loads_after() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

// ---- bounded grid barrier on top of that hand-over: for launches whose blocks are all co-resident (grid <= one block per CU) --------
// sync: int32[3] in device memory, zero on entry: [0] arrivals, [1] departures, [2] error flag. grid_wait returns false when the
// other blocks did not arrive within ~1 s (the caller leaves at once; [2] stays set): no wave spins forever. grid_leave, called once
// by every block after its last wait, restores [0] and [1] to zero for the next launch.
constexpr int GN_SPIN_LIMIT = 1 << 22;
__device__ __forceinline__ void grid_arrive(int* sync) {
    dev_stores_done_block();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&sync[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool grid_wait(int* sync, int phase, int nblocks) {
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        const int target = phase * nblocks;
        int spins = 0;
        while (__hip_atomic_load(&sync[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && spins < GN_SPIN_LIMIT) {
            __builtin_amdgcn_s_sleep(1);
            ++spins;
        }
        s_ok = spins < GN_SPIN_LIMIT;
        if (!s_ok) __hip_atomic_store(&sync[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    dev_loads_after();
    return s_ok != 0;
}
__device__ __forceinline__ bool grid_barrier(int* sync, int phase, int nblocks) {
    grid_arrive(sync);
    return grid_wait(sync, phase, nblocks);
}
__device__ __forceinline__ void grid_leave(int* sync, int nblocks) {
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(&sync[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == nblocks - 1) {
            __hip_atomic_store(&sync[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sync[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// DPP (data-parallel primitive) moves: a cross-lane operand fetch inside the VALU, ~10x cheaper than the ds_bpermute
// behind __shfl_xor. old = 0 for lanes outside ROW_MASK.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
// sum over each aligned group of 16 lanes (a DPP row), returned to all 16 lanes
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}
// sum over the 64 lanes, returned as a wave-uniform value (fixed order => deterministic)
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_mov<0xB1>(v);          // quad_perm:[1,0,3,2]
    v += dpp_mov<0x4E>(v);          // quad_perm:[2,3,0,1]
    v += dpp_mov<0x141>(v);         // row_half_mirror: the other quad pair of each 8 lanes
    v += dpp_mov<0x140>(v);         // row_mirror: the other half of each row of 16
    v += dpp_mov<0x142, 0xA>(v);    // row_bcast:15 -> rows 1 and 3 add the totals of rows 0 and 2
    v += dpp_mov<0x143, 0xC>(v);    // row_bcast:31 -> rows 2 and 3 add the total of rows 0..1
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace goalnet
