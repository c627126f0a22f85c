#include <hip/hip_runtime.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned long long *in, unsigned long long *out)
{
    unsigned long long key = in[threadIdx.x];
    unsigned lo = (unsigned)key, hi = (unsigned)(key >> 32);
    u32x2 a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    u32x2 b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    // partner value: for lane < 32 it is element [1] (what came from the upper half), for lane >= 32 element [0]
    unsigned plo = threadIdx.x < 32 ? a[1] : a[0], phi = threadIdx.x < 32 ? b[1] : b[0];
    unsigned long long o = ((unsigned long long)phi << 32) | plo;
    u32x2 c = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    u32x2 d = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const bool odd = (threadIdx.x >> 4) & 1;
    unsigned qlo = odd ? c[0] : c[1], qhi = odd ? d[0] : d[1];
    out[threadIdx.x] = o;
    out[64 + threadIdx.x] = ((unsigned long long)qhi << 32) | qlo;
    out[128 + threadIdx.x] = __shfl_xor(key, 32);
    out[192 + threadIdx.x] = __shfl_xor(key, 16);
}
int main()
{
    unsigned long long h[64], r[256], *di, *dout;
    for (int i = 0; i < 64; ++i) h[i] = 0x1000000010000ull * i + i;
    hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(r));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(r, dout, sizeof(r), hipMemcpyDeviceToHost);
    int bad32 = 0, bad16 = 0;
    for (int i = 0; i < 64; ++i) bad32 += r[i] != r[128 + i], bad16 += r[64 + i] != r[192 + i];
    printf("xor32 mismatches %d, xor16 mismatches %d\n", bad32, bad16);
    if (bad32 || bad16) for (int i = 0; i < 64; i += 5) printf("lane %d: swap32 %llx want %llx | swap16 %llx want %llx\n", i, r[i], r[128+i], r[64+i], r[192+i]);
    return 0;
}
