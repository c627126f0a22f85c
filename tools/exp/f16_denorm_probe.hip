// Does the half-precision path of the assign filter keep subnormals?  (a) float -> half conversion of values below 2^-14,
// (b) v_mfma_f32_32x32x16_f16 with subnormal A / B elements.  Prints what the hardware does; tests/test_gpu_kmeans.py pins it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *in, unsigned short *cv, float *out)
{
    // (a)
    f2 v = {in[0], in[1]};
    h2 h = __builtin_convertvector(v, h2);
    cv[0] = __builtin_bit_cast(unsigned short, h[0]);
    cv[1] = __builtin_bit_cast(unsigned short, h[1]);
    // (b) A[i][k] = a for all, B[k][j] = b
    h8 a, b;
    for (int q = 0; q < 8; ++q) a[q] = (_Float16)in[2], b[q] = (_Float16)in[3];
    f16v acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
    // subnormal x subnormal never matters (product < 2^-28); subnormal A, normal B the other way round too
    acc = (f16v){0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[1] = acc[0];
}
int main()
{
    float h_in[4] = {ldexpf(1.0f, -20), ldexpf(1.5f, -24), ldexpf(3.0f, -22), 1024.0f};
    float *in, *out; unsigned short *cv;
    hipMalloc(&in, 16); hipMalloc(&out, 8); hipMalloc(&cv, 4);
    hipMemcpy(in, h_in, 16, hipMemcpyHostToDevice);
    k<<<1, 64>>>(in, cv, out);
    float o[2]; unsigned short c[2];
    hipMemcpy(o, out, 8, hipMemcpyDeviceToHost); hipMemcpy(c, cv, 4, hipMemcpyDeviceToHost);
    printf("cvt(2^-20) = 0x%04x (subnormal kept: 0x0010), cvt(1.5 2^-24) = 0x%04x (RNE to even: 0x0002)\n", c[0], c[1]);
    printf("mfma 16 x (3 2^-22 * 1024) = %g / %g, exact %g (0 = subnormal inputs flushed)\n", o[0], o[1], 16 * ldexpf(3.0f, -12));
    return 0;
}
