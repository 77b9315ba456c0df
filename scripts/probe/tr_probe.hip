// Probe of ds_read_b64_tr_b16 semantics on gfx950: which element does each lane receive?
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -o /tmp/tr_probe scripts/probe/tr_probe.hip && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const short* x, short* y) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = x[i];
    __syncthreads();
    const int l = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int q = l >> 2, p = l & 3;
    __attribute__((address_space(3))) s16x4* ptr = (__attribute__((address_space(3))) s16x4*)&lds[q * 128 + 16 * g + 4 * p];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    reinterpret_cast<s16x4*>(y)[threadIdx.x] = v;
}
int main() {
    short h[4096], o[256];
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 128; ++c) h[r * 128 + c] = (short)(r * 1000 + c);
    short *dx, *dy;
    hipMalloc(&dx, sizeof(h)); hipMalloc(&dy, sizeof(o));
    hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy);
    hipMemcpy(o, dy, sizeof(o), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) if (l < 20 || l % 16 == 0) printf("lane %2d: %5d %5d %5d %5d\n", l, o[4*l], o[4*l+1], o[4*l+2], o[4*l+3]);
    return 0;
}
