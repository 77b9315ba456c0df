// Counter-based splitmix64 streams shared by the generators (fill.hip) and the device-counter variants (stepstate.hip).
// Device twin of cvml_goalnet_amd/synth.py: element i of stream `tid` depends only on (seed, tid, i).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace goalnet {

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
__host__ __device__ __forceinline__ uint64_t stream_key(uint64_t seed, uint32_t tid) {
    return mix64(seed + (uint64_t)(tid + 1u) * 0xD1342543DE82EF95ull);
}
__device__ __forceinline__ float unit24(uint64_t key, int64_t i) {
    const uint64_t b = mix64(key + (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull);
    return (float)(uint32_t)(b >> 40) * 5.9604644775390625e-08f;   // 2^-24, exact
}

}  // namespace goalnet
