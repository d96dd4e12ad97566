#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* x) {
    float v = x[threadIdx.x];
    unsigned u = __float_as_uint(v);
    auto q = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    float s = __uint_as_float(q[0]) + __uint_as_float(q[1]);
    unsigned us = __float_as_uint(s);
    auto r = __builtin_amdgcn_permlane32_swap(us, us, false, false);
    x[64 + threadIdx.x] = s;
    x[128 + threadIdx.x] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
int main() {
    float h[192]; for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i / 16)) * 1000.f + (i % 16);
    float* d; hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        float w16 = (float)((1 << (i / 16)) + (1 << ((i / 16) ^ 1))) * 1000.f + 2 * (i % 16);
        float w = 15000.f + 4 * (i % 16);
        if (h[64 + i] != w16 || h[128 + i] != w) { ++bad; printf("lane %d: xor16 %g (want %g) all %g (want %g)\n", i, h[64 + i], w16, h[128 + i], w); }
    }
    printf("permlane swap sums: %s\n", bad ? "MISMATCH" : "ok");
    return bad != 0;
}
