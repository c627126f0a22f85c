// Experiment harness (not product): the K <= 256 assign filter under SUSTAINED load, with the shader clock measured on the device.
//   ./sustained_bench rows d K launches [gap_us] [data: mix|uni]
// * `launches` launches of k_assign_f16_rw<true, 4, false, 2, 0> (the product's tile and schedule; the product's EMIT = 2 epilogue
//   measures the same) back to back on one stream, a HIP event pair around every launch, nothing between them
//   (gap_us > 0: the host sleeps that long between launches -- the "after a pause" regime of the timed bench pass).
// * next to them, on a second stream, ONE wave of k_clock_probe: every 100 us it stores (s_memrealtime, s_memtime delta over
//   s_memrealtime delta) -- the effective shader clock of the CU it sits on, 100 us resolution, no counters, no serialisation.
// * rows: the bench's Gaussian mixture (component centre + 0.3 N(0,1)) or uniform (a power-hungrier pattern, NOTES_r03).
// Build: tools/exp/build_sustained.sh [name] [-DACAV_ABL_NOMFMA ...]; driven (with amdsmi sampling beside it) by tools/filter_sustained.py.
// Labels are not checked here (ablation builds produce garbage): timing and power only.
#include "../../acav100m_amd/csrc/acav_kmeans_assign.hip"
#include <algorithm>
#include <chrono>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned hash_u32(unsigned long long i, unsigned seed)
{
    unsigned h = (unsigned)i * 2654435761u ^ (unsigned)(i >> 32) * 0x9E3779B9u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ float gauss(unsigned long long i, unsigned seed)
{
    const float u1 = ((float)(hash_u32(i, seed) >> 8) + 1.0f) * (1.0f / 16777217.0f);
    const float u2 = (float)(hash_u32(i, seed ^ 0x5bd1e995u) >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
}
__global__ void k_centres(float *c, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) c[i] = gauss(i, 77u); }
__global__ void k_rows(float *x, const float *c, size_t n, int d, int K, int uniform)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n * (size_t)d; i += stride) {
        const size_t row = i / d;
        const int col = (int)(i % d);
        if (uniform) x[i] = (float)(int)(hash_u32(i, 1u) & 0xffff) * (1.0f / 32768.0f) - 1.0f;
        else x[i] = c[(size_t)(hash_u32(row, 9u) % (unsigned)K) * d + col] + 0.3f * gauss(i, 3u);
    }
}
__global__ void k_tohalf(const float *c, fl16 *o, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) o[i] = (fl16)c[i]; }
__global__ void k_norms(const float *c, float *cn, int K, int d)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    float s = 0.f;
    for (int j = 0; j < d; ++j) s += c[(size_t)k * d + j] * c[(size_t)k * d + j];
    cn[k] = s;
}
// one wave; lane 0 samples.  out[2 i] = s_memrealtime at the end of sample i (100 MHz), out[2 i + 1] = shader cycles per 100 MHz tick x 1000
__global__ void k_clock_probe(unsigned long long *out, int nsamp, unsigned long long period, volatile int *stop, int *taken)
{
    if (threadIdx.x != 0) return;
    int i = 0;
    for (; i < nsamp && !*stop; ++i) {
        const unsigned long long w0 = wall_clock64(), c0 = clock64();
        unsigned long long w1;
        do { __builtin_amdgcn_s_sleep(32); w1 = wall_clock64(); } while (w1 - w0 < period);
        const unsigned long long c1 = clock64();
        out[2 * i] = w1;
        out[2 * i + 1] = (c1 - c0) * 1000ull / (w1 - w0);
    }
    *taken = i;
}

static long long now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
    const int d = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 256;
    const int launches = argc > 4 ? atoi(argv[4]) : 200;
    const int gap_us = argc > 5 ? atoi(argv[5]) : 0;
    const int uniform = argc > 6 && !strcmp(argv[6], "uni");
    float *x, *c, *cn, *counts;
    fl16 *cb;
    CentersAux *aux;
    int64_t *lab;
    int *rl;
    unsigned *rc;
    CK(hipMalloc(&x, (size_t)n * d * 4));
    CK(hipMalloc(&c, (size_t)K * d * 4));
    CK(hipMalloc(&cb, (size_t)K * d * 2));
    CK(hipMalloc(&cn, K * 4));
    CK(hipMalloc(&counts, K * 4));
    CK(hipMalloc(&aux, sizeof(CentersAux)));
    CK(hipMalloc(&lab, n * 8));
    CK(hipMalloc(&rl, n * 4));
    CK(hipMalloc(&rc, 4));
    hipLaunchKernelGGL(k_centres, dim3((unsigned)(((size_t)K * d + 255) / 256)), dim3(256), 0, 0, c, (size_t)K * d);
    hipLaunchKernelGGL(k_rows, dim3(65536), dim3(256), 0, 0, x, c, (size_t)n, d, K, uniform);
    hipLaunchKernelGGL(k_tohalf, dim3((unsigned)(((size_t)K * d + 255) / 256)), dim3(256), 0, 0, c, cb, (size_t)K * d);
    hipLaunchKernelGGL(k_norms, dim3((K + 255) / 256), dim3(256), 0, 0, c, cn, K, d);
    {
        std::vector<float> ones((size_t)K, 1000.0f);
        CK(hipMemcpy(counts, ones.data(), K * 4, hipMemcpyHostToDevice));
        CentersAux h{};
        h.sx = h.sc = h.inv_ss = 1.0f;
        CK(hipMemcpy(aux, &h, sizeof(h), hipMemcpyHostToDevice));
    }
    CK(hipDeviceSynchronize());

#ifdef SUSTAINED_NW8  // the 256-row tile: one workgroup per CU, centre ring 3, DMA pieces spread -- half the centre bytes per row
    constexpr int NW = 8;
    auto kern = k_assign_f16_rw<true, NW, false, 3, 2>;
    const int fsmem = FD_DX * NW * 4096 + 3 * FD_SLOT;
#elif defined(SUSTAINED_K1024)  // K > 256: the product's (row tile, centre group) pair kernel -- MFMA-bound; is ITS clock capped too?
    constexpr int NW = 8;
    auto kern = k_assign_f16_rw<true, NW, true, 3, 2>;
    const int fsmem = FD_DX * NW * 4096 + 3 * FD_SLOT;
#else
    constexpr int NW = 4;
    auto kern = k_assign_f16_rw<true, NW, false, 2, 0>;
    const int fsmem = FD_DX * NW * 4096 + 2 * FD_SLOT;
#endif
#ifdef SUSTAINED_K1024
    const int ngroups = (K + 255) / 256;
    const int64_t ntiles = (n + NW * 32 - 1) / (NW * 32);
    const int64_t grid = (ntiles + 7) / 8 * 8 * ngroups;
    Top2Rec *grec;
    CK(hipMalloc(&grec, sizeof(Top2Rec) * (size_t)ngroups * n));
#else
    const int64_t grid = (n + NW * 32 - 1) / (NW * 32);
    Top2Rec *grec = nullptr;
#endif
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, fsmem));
    hipStream_t sk, sp;
    CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(2 * (size_t)launches);
    for (auto &e : ev) CK(hipEventCreate(&e));

    const int nsamp = 60000;  // 6 s of 100 us samples at most
    unsigned long long *probe;
    int *stop, *taken;
    CK(hipMalloc(&probe, sizeof(unsigned long long) * 2 * nsamp));
    CK(hipHostMalloc(&stop, sizeof(int), hipHostMallocMapped));
    CK(hipHostMalloc(&taken, sizeof(int), hipHostMallocMapped));
    *stop = 0, *taken = 0;
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, sp, probe, nsamp, 10000ull, stop, taken);
    std::this_thread::sleep_for(std::chrono::milliseconds(300));  // idle lead-in: the clock at rest

    // anchor: one tiny probe-clock read maps s_memrealtime onto the host's wall clock is not available; the lead-in length does it
    const long long t_host0 = now_ns();
    for (int i = 0; i < launches; ++i) {
        CK(hipEventRecord(ev[2 * i], sk));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), fsmem, sk, x, n, d, cb, cn, counts, K, 1.0f, 5.0f, aux, 1e-2f, 1e-4f,
                           1e-6f, lab, rl, rc, (AssignCtl *)nullptr, grec, CandOut{});
        CK(hipEventRecord(ev[2 * i + 1], sk));
        if (gap_us > 0) {
            CK(hipEventSynchronize(ev[2 * i + 1]));
            std::this_thread::sleep_for(std::chrono::microseconds(gap_us));
        }
    }
    CK(hipStreamSynchronize(sk));
    const long long t_host1 = now_ns();
    std::this_thread::sleep_for(std::chrono::milliseconds(100));  // tail: the clock after the load
    *stop = 1;
    CK(hipStreamSynchronize(sp));

    std::vector<float> ms((size_t)launches), at((size_t)launches);
    for (int i = 0; i < launches; ++i) {
        CK(hipEventElapsedTime(&ms[i], ev[2 * i], ev[2 * i + 1]));
        CK(hipEventElapsedTime(&at[i], ev[0], ev[2 * i]));
    }
    std::vector<unsigned long long> pr(2 * (size_t)*taken);
    CK(hipMemcpy(pr.data(), probe, sizeof(unsigned long long) * pr.size(), hipMemcpyDeviceToHost));
    const double bytes = (double)n * d * 4 + (double)n * 8;
    printf("{\"rows\": %lld, \"d\": %d, \"K\": %d, \"launches\": %d, \"gap_us\": %d, \"data\": \"%s\", \"ablation\": \"%s\",\n", (long long)n, d, K, launches,
           gap_us, uniform ? "uniform" : "mixture",
#ifdef ACAV_ABL_NOMFMA
           "no MFMA"
#elif defined(SUSTAINED_NW8)
           "none (256-row tile)"
#elif defined(SUSTAINED_K1024)
           "none (K > 256 pair kernel)"
#else
           "none"
#endif
    );
    printf(" \"host_start_ns\": %lld, \"host_end_ns\": %lld, \"wall_ms_per_launch\": %.4f,\n", t_host0, t_host1, (t_host1 - t_host0) * 1e-6 / launches);
    printf(" \"launch_ms\": [");
    for (int i = 0; i < launches; ++i) printf("%s%.4f", i ? ", " : "", ms[i]);
    printf("],\n \"launch_at_ms\": [");
    for (int i = 0; i < launches; ++i) printf("%s%.3f", i ? ", " : "", at[i]);
    // the probe trace, thinned to 1 ms means (10 samples); time in ms from the probe's first sample
    printf("],\n \"probe_ms\": [");
    const size_t ns = pr.size() / 2;
    bool first = true;
    for (size_t i = 0; i + 10 <= ns; i += 10, first = false) printf("%s%.2f", first ? "" : ", ", (double)(pr[2 * (i + 9)] - pr[0]) * 1e-5);
    printf("],\n \"probe_ghz\": [");
    first = true;
    for (size_t i = 0; i + 10 <= ns; i += 10, first = false) {
        double s = 0;
        for (size_t j = i; j < i + 10; ++j) s += (double)pr[2 * j + 1];
        printf("%s%.3f", first ? "" : ", ", s / 10 * 1e-3 * 0.1);
    }
    std::vector<float> tail(ms.begin() + launches / 2, ms.end());
    std::sort(tail.begin(), tail.end());
    const double settled = tail[tail.size() / 2];
    printf("],\n \"tflops_settled\": %.1f, \"first_ms\": %.4f, \"settled_ms\": %.4f, \"frac_first\": %.4f, \"frac_settled\": %.4f}\n", 2.0 * n * (double)K * d / (settled * 1e-3) * 1e-12, ms[0], settled,
           bytes / (ms[0] * 1e-3) / 8e12, bytes / (settled * 1e-3) / 8e12);
    return 0;
}
