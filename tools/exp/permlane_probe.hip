// Probe (not product): which result of v_permlane32_swap carries the lower half's value into the upper half.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float *a, float *o)
{
    float x = threadIdx.x < 32 ? a[threadIdx.x] : -1.f;
    unsigned xi = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(xi, xi, false, false);
    o[threadIdx.x] = __uint_as_float(r[0]);
    o[64 + threadIdx.x] = __uint_as_float(r[1]);
}
int main()
{
    float h[64], o[128], *da, *dout;
    for (int i = 0; i < 64; ++i) h[i] = (float)(100 + i);
    hipMalloc(&da, 256); hipMalloc(&dout, 512);
    hipMemcpy(da, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dout);
    hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
    printf("r[0]: lane 0 %.0f lane 5 %.0f lane 32 %.0f lane 37 %.0f\n", o[0], o[5], o[32], o[37]);
    printf("r[1]: lane 0 %.0f lane 5 %.0f lane 32 %.0f lane 37 %.0f\n", o[64], o[69], o[96], o[101]);
    return 0;
}
