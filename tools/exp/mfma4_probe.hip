// Probe (not product): lane mapping and rounding of v_mfma_f32_4x4x1_16B_f32 against a chain of fmaf.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_probe(const float *A, const float *B, float *out, int n)
{
    // block b = lane / 4; A element i = lane % 4 of block b; B element j = lane % 4 of block b; K = n steps
    const int lane = threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < n; ++k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[k * 64 + lane], B[k * 64 + lane], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}
int main()
{
    const int n = 256;
    float *hA = (float *)malloc(4 * n * 64), *hB = (float *)malloc(4 * n * 64), hO[256];
    srand(5);
    for (int i = 0; i < n * 64; ++i) { hA[i] = (rand() / (float)RAND_MAX - 0.5f) * 3.f; hB[i] = (rand() / (float)RAND_MAX - 0.5f) * 3.f; }
    float *dA, *dB, *dO;
    hipMalloc(&dA, 4 * n * 64); hipMalloc(&dB, 4 * n * 64); hipMalloc(&dO, 1024);
    hipMemcpy(dA, hA, 4 * n * 64, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 4 * n * 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dO, n);
    hipMemcpy(hO, dO, 1024, hipMemcpyDeviceToHost);
    // hypothesis: D[block b][i = r][j = lane % 4] in lane, register r;  A(i) from lane 4 b + i, B(j) from lane 4 b + j
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int r = 0; r < 4; ++r) {
            const int b = lane / 4, j = lane % 4, i = r;
            float acc = 0.f;
            for (int k = 0; k < n; ++k) acc = fmaf(hA[k * 64 + 4 * b + i], hB[k * 64 + 4 * b + j], acc);
            if (acc != hO[lane * 4 + r]) { if (bad < 5) printf("lane %d r %d: got %.9g want %.9g\n", lane, r, hO[lane * 4 + r], acc); ++bad; }
        }
    printf("mismatches under the (i = register, j = lane %% 4) hypothesis: %d of 256\n", bad);
    if (bad) {  // the transposed hypothesis
        int bad2 = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r) {
                const int b = lane / 4, i = lane % 4, j = r;
                float acc = 0.f;
                for (int k = 0; k < n; ++k) acc = fmaf(hA[k * 64 + 4 * b + i], hB[k * 64 + 4 * b + j], acc);
                if (acc != hO[lane * 4 + r]) ++bad2;
            }
        printf("mismatches under the transposed hypothesis: %d of 256\n", bad2);
    }
    return 0;
}
