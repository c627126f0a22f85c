// Experiment harness (not product): the assign filter k_assign_f16_rw<NT, NW, GS> of acav_kmeans.hip launched alone on
// synthetic rows, for timing-only ablations (-DACAV_ABL_NOMFMA / NOAFRAG / NOCDMA / NOXDMA) and tile-shape comparisons.
// Labels are garbage under an ablation.  Build: tools/exp/build_assign.sh <name> [-D...]; run: ./<name> rows d K
#include "../../acav100m_amd/csrc/acav_kmeans_assign.hip"
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void k_fill(float *p, size_t n, unsigned seed)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(int)(h & 0xffff) * (1.0f / 32768.0f) - 1.0f;
    }
}
__global__ void k_tobf(const float *c, fl16 *o, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) o[i] = (fl16)c[i];
}

template <bool NT, int NW, bool GS, int DCR = 2, int SCHED = 0>
static void bench(const char *name, const float *x, int64_t n, int d, int K, const fl16 *cb, const float *cn, const float *counts,
                  const CentersAux *aux, int64_t *lab, int *rl, unsigned *rc, Top2Rec *grec)
{
    const int ngroups = (K + 255) / 256;
    const int fsmem = FD_DX * NW * 4096 + DCR * FD_SLOT;
    const int64_t tile_rows = NW * 32, ntiles = (n + tile_rows - 1) / tile_rows;
    const int64_t grid = GS ? (ntiles + 7) / 8 * 8 * ngroups : ntiles;
    auto kern = k_assign_f16_rw<NT, NW, GS, DCR, SCHED>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, fsmem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemset(rc, 0, 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), fsmem, 0, x, n, d, cb, cn, counts, K, 1.0f, 5.0f, aux, 1e-2f,
                           1e-4f, 1e-6f, lab, rl, rc, (AssignCtl *)nullptr, grec, CandOut{});
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (rep) ms.push_back(t);
    }
#ifdef ACAV_RW_PROF
    {
        unsigned long long hp[16];
        CK(hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_rw_prof), sizeof(hp)));
        for (int o = 0; o < 16; o += 8)
            printf("   wave %s: per stage: vmwait %.0f  barrier %.0f  frag reads %.0f  dma issue %.0f  cvt+mfma(+late dma) %.0f cycles | stage %.0f cycles; "
                   "%.2f GHz\n", o ? "last" : "0", (double)hp[o] / hp[o + 5], (double)hp[o + 1] / hp[o + 5], (double)hp[o + 2] / hp[o + 5],
                   (double)hp[o + 3] / hp[o + 5], (double)hp[o + 4] / hp[o + 5], (double)(hp[o] + hp[o + 1] + hp[o + 2] + hp[o + 3] + hp[o + 4]) / hp[o + 5],
                   (double)hp[o + 6] / hp[o + 7] * 0.1);
        memset(hp, 0, sizeof(hp));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rw_prof), hp, sizeof(hp)));
    }
#endif
    std::sort(ms.begin(), ms.end());
    const double flops = 2.0 * n * (double)K * d;
    printf("%-28s %.3f ms (min %.3f)  %.0f TFLOP/s = %.3f of 2.5 PF   rows %.2f TB/s\n", name, ms[ms.size() / 2], ms[0],
           flops / ms[ms.size() / 2] * 1e-9, flops / ms[ms.size() / 2] * 1e-9 / 2500.0, n * (double)d * 4 / ms[ms.size() / 2] * 1e-9);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
    const int d = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1024;
    float *x, *c, *cn, *counts;
    fl16 *cb;
    CentersAux *aux;
    int64_t *lab;
    int *rl;
    unsigned *rc;
    Top2Rec *grec;
    CK(hipMalloc(&x, (size_t)n * d * 4));
    CK(hipMalloc(&c, (size_t)K * d * 4));
    CK(hipMalloc(&cb, (size_t)K * d * 2));
    CK(hipMalloc(&cn, K * 4));
    CK(hipMalloc(&counts, K * 4));
    CK(hipMalloc(&aux, sizeof(CentersAux)));
    CK(hipMalloc(&lab, n * 8));
    CK(hipMalloc(&rl, n * 4));
    CK(hipMalloc(&rc, 4));
    CK(hipMalloc(&grec, sizeof(Top2Rec) * (size_t)((K + 255) / 256) * n));
    hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, x, (size_t)n * d, 1u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, c, (size_t)K * d, 2u);
    hipLaunchKernelGGL(k_tobf, dim3((unsigned)(((size_t)K * d + 255) / 256)), dim3(256), 0, 0, c, cb, (size_t)K * d);
    hipLaunchKernelGGL(k_fill, dim3(4), dim3(256), 0, 0, cn, (size_t)K, 3u);
    hipLaunchKernelGGL(k_fill, dim3(4), dim3(256), 0, 0, counts, (size_t)K, 4u);
    {  // scales of the half-precision operands: 1 (the harness's values are in [-1, 1))
        CentersAux h{};
        h.sx = h.sc = h.inv_ss = 1.0f;
        CK(hipMemcpy(aux, &h, sizeof(h), hipMemcpyHostToDevice));
    }
    CK(hipDeviceSynchronize());
    printf("rows %lld d %d K %d\n", (long long)n, d, K);
#define B(...) bench<__VA_ARGS__>(#__VA_ARGS__, x, n, d, K, cb, cn, counts, aux, lab, rl, rc, grec)
    if (K > 256) {
        B(true, 8, true, 2, 0);
        B(true, 8, true, 2, 1);
        B(true, 8, true, 2, 2);
        B(true, 8, true, 3, 2);
        B(false, 8, true, 3, 2);
        B(true, 4, true, 2, 0);
        B(true, 4, true, 2, 2);
    } else {
        B(true, 4, false, 2, 0);
        B(true, 4, false, 2, 2);
        B(true, 8, false, 2, 1);
        B(true, 8, false, 3, 2);
    }
    return 0;
}
