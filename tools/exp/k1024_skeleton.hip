// Experiment (not product): the DMA / issue SKELETON of a different K > 256 assign filter, timed before anything
// is built.  Product (k_assign_f16_rw<true, 8, true, 3, 2>): one workgroup per (256-row tile, 256-centre group) pair, the tile's fp32
// rows AND the half-precision centres both stream through LDS-DMA rings (32 + 16 KB per 32-column stage, 3 slots each = 144 KB), one
// barrier per stage; its stage takes ~1 us whatever the shader clock is (2 090 cycles at 1.75 GHz with the MFMAs, 2 250 at 2.4 GHz
// without them: profiles/r06_filter_sustained.txt) -- the time a DMA piece needs to land, two stages of look-ahead.
// Candidate: the rows come by plain coalesced global_load_dwordx4 into VGPRs (their own vmcnt budget, D stages of look-ahead in
// registers), are converted to half there (v_cvt_pk_f16_f32) and written to LDS as HALF (16 KB per stage: a ring of 6 fits where the
// fp32 ring held 3); the centres keep their LDS-DMA ring.  Per stage every wave then reads what the MFMAs would read (its 32 rows'
// fragments and all 256 centres' fragments) and folds them into a dummy register -- the same "no MFMA" skeleton as the product's
// -DACAV_ABL_NOMFMA ablation (1.83 ms per 1M x 1024, K = 1024).  Build the kernel only if this is >= 15 % faster.
//   ./k1024_skeleton [rows] [d] [K]       (labels: none; timing only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return __builtin_amdgcn_readfirstlane((unsigned)(__SIZE_TYPE__)(const __attribute__((address_space(3))) void *)(p));
}
__device__ __forceinline__ f4 ld16_nt(const float *p)  // asynchronous as far as the compiler knows NOTHING: the explicit vmcnt waits below order it
{
    f4 v;
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void dma16(const void *gbase_uniform, unsigned voff, unsigned lds)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(gbase_uniform), "s"(lds) : "memory", "m0");
}

// D = stages of row look-ahead in registers (4 dwordx4 per lane and stage), RS = slots of the half-row ring in LDS, CS = centre ring slots
template <int D, int RS, int CS>
__global__ __launch_bounds__(512) void k_skel(const float *__restrict__ x, long long n, int d, const _Float16 *__restrict__ cb, int ngroups,
                                              unsigned *__restrict__ sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    _Float16 *sX = reinterpret_cast<_Float16 *>(smem);                       // [RS][256 rows][32] half = 16 KB per slot
    _Float16 *sC = reinterpret_cast<_Float16 *>(smem + RS * 16384);          // [CS][256 centres][32] half = 16 KB per slot
    const int tid = threadIdx.x, lane = tid & 63;
    const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = blockIdx.x >> 3;
    const long long tile = (long long)(j / ngroups) * 8 + (blockIdx.x & 7);  // the pairs of a tile side by side on one XCD (as the product)
    const int cg = j % ngroups;
    if (tile * 256 >= n) return;
    const long long row0 = tile * 256;
    const int nchunks = d / 32;
    // this lane's pieces of the wave's 32 rows: piece q covers rows wq * 32 + q * 8 + (lane >> 3), 16 bytes at column (lane & 7) * 4
    const float *px[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        long long r = row0 + wq * 32 + q * 8 + (lane >> 3);
        if (r >= n) r = n - 1;
        px[q] = x + (size_t)r * d + (lane & 7) * 4;
    }
    unsigned voffc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int rr = (wq * 2 + q) * 16 + (lane >> 2);
        voffc[q] = (unsigned)rr * 64u + ((lane & 3) << 4);
    }
    const char *gc = reinterpret_cast<const char *>(cb + (size_t)cg * 256 * 32);
    const size_t cstage = (size_t)ngroups * 256 * 64;
    const unsigned cring = lds_addr(sC) + wq * 2048;
    f4 xr[D][4];
    // prologue: D stages of rows in flight, CS - 1 stages of centres
#pragma unroll
    for (int s = 0; s < D; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            xr[s][q] = ld16_nt(px[q] + (s < nchunks ? s : 0) * 32);
    int wc = 0;
#pragma unroll
    for (int s = 0; s < CS - 1; ++s) {
#pragma unroll
        for (int q = 0; q < 2; ++q) dma16(gc, voffc[q], cring + wc * 16384 + q * 1024);
        gc += cstage;
        wc = wc + 1 == CS ? 0 : wc + 1;
    }
    unsigned acc = 0;
    int ws = 0, rs = 0, rcs = 0;
    constexpr int LA = D > CS ? D : CS;
    for (int c0 = 0; c0 < nchunks; c0 += D) {  // (nchunks % D == 0: the look-ahead slots are indexed at compile time -- registers, not scratch)
#pragma unroll
      for (int sl = 0; sl < D; ++sl) {
        const int c = c0 + sl;
        // vmcnt is in issue order.  Per iteration: 4 row loads (stage c + D), then -- after the barrier -- 2 centre pieces (stage c + CS - 1).
        // Stage c's rows were issued D iterations ago: younger are the rows of D - 1 iterations and the centre pieces of D iterations.
        if (c == 0 || c + LA >= nchunks) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (first stage: the whole prologue; tail: the counts shrink)
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (D - 1) + 2 * D) : "memory");
        f4 cur[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = xr[sl][q];
        // convert to half and write this wave's 32 x 32 block of the stage into the half-row ring
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const h2 a = __builtin_convertvector(__builtin_shufflevector(cur[q], cur[q], 0, 1), h2);
            const h2 b = __builtin_convertvector(__builtin_shufflevector(cur[q], cur[q], 2, 3), h2);
            unsigned long long packed = (unsigned long long)__builtin_bit_cast(unsigned, a) | ((unsigned long long)__builtin_bit_cast(unsigned, b) << 32);
            *reinterpret_cast<unsigned long long *>(sX + (size_t)ws * 8192 + (wq * 32 + q * 8 + (lane >> 3)) * 32 + (lane & 7) * 4) = packed;
        }
        ws = ws + 1 == RS ? 0 : ws + 1;
        // refill the look-ahead slot
#pragma unroll
        for (int q = 0; q < 4; ++q) xr[sl][q] = ld16_nt(px[q] + (c + D < nchunks ? c + D : c) * 32);
        // centre stage c was issued CS - 1 iterations ago, after that iteration's row loads: younger are the rows of CS - 1 iterations
        // and the centre pieces of CS - 2
        if (c != 0 && c + LA < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (CS - 1) + 2 * (CS - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // the slot stage c - 1 vacated: centre stage c + CS - 1
        if (c + CS - 1 < nchunks) {
#pragma unroll
            for (int q = 0; q < 2; ++q) dma16(gc, voffc[q], cring + wc * 16384 + q * 1024);
            gc += cstage;
            wc = wc + 1 == CS ? 0 : wc + 1;
        }
        // what the MFMAs would read: the wave's own rows (lane: row lane & 31, 16 halves) and all 256 centres (8 tiles x 2 k-steps)
        const u4 *pr = reinterpret_cast<const u4 *>(sX + (size_t)rs * 8192 + (wq * 32 + (lane & 31)) * 32 + (lane >> 5) * 16);
        const u4 r0 = pr[0], r1 = pr[1];
        acc ^= r0.x ^ r0.y ^ r0.z ^ r0.w ^ r1.x ^ r1.y ^ r1.z ^ r1.w;
        const u4 *pc = reinterpret_cast<const u4 *>(sC + (size_t)rcs * 8192 + (lane & 31) * 32 + (lane >> 5) * 8);
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            const u4 a0 = pc[ct * 128], a1 = pc[ct * 128 + 2];  // tile ct: centres 32 ct .. 32 ct + 31; the two k-steps
            acc ^= a0.x ^ a0.y ^ a0.z ^ a0.w ^ a1.x ^ a1.y ^ a1.z ^ a1.w;
        }
        rs = rs + 1 == RS ? 0 : rs + 1;
        rcs = rcs + 1 == CS ? 0 : rcs + 1;
      }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int D, int RS, int CS>
static void run(const char *name, const float *x, long long n, int d, int K, const _Float16 *cb, unsigned *sink)
{
    const int ngroups = (K + 255) / 256;
    const long long ntiles = (n + 255) / 256;
    const long long grid = (ntiles + 7) / 8 * 8 * ngroups;
    const int smem = (RS + CS) * 16384;
    auto kern = k_skel<D, RS, CS>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int rep = 0; rep < 12; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, 0, x, n, d, cb, ngroups, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (rep >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    printf("%-74s LDS %3d KB  %.3f ms (min %.3f)\n", name, smem / 1024, ms[ms.size() / 2], ms[0]);
    fflush(stdout);
}

__global__ void k_fill(float *p, size_t n)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)i * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(int)(h & 0xffff) * (1.0f / 32768.0f) - 1.0f;
    }
}

int main(int argc, char **argv)
{
    const long long n = argc > 1 ? atoll(argv[1]) : 1000000;
    const int d = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 1024;
    float *x;
    _Float16 *cb;
    unsigned *sink;
    CK(hipMalloc(&x, (size_t)n * d * 4));
    CK(hipMalloc(&cb, (size_t)((K + 255) / 256 * 256) * d * 2));
    CK(hipMalloc(&sink, 64));
    hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, x, (size_t)n * d);
    CK(hipMemset(cb, 0x3c, (size_t)((K + 255) / 256 * 256) * d * 2));
    CK(hipDeviceSynchronize());
    printf("rows %lld d %d K %d: skeleton of the candidate kernel (rows by register-staged loads -> half ring in LDS; centres by LDS-DMA); the product's\n"
           "own no-MFMA skeleton: 1.83 ms, the product with MFMAs 2.33 ms (1M x 1024, K = 1024)\n", n, d, K);
    run<2, 3, 3>("row look-ahead 2 stages in registers, half-row ring 3, centre ring 3", x, n, d, K, cb, sink);
    run<4, 3, 3>("row look-ahead 4 stages, half-row ring 3, centre ring 3", x, n, d, K, cb, sink);
    run<8, 3, 3>("row look-ahead 8 stages, half-row ring 3, centre ring 3", x, n, d, K, cb, sink);
    run<4, 3, 4>("row look-ahead 4 stages, half-row ring 3, centre ring 4", x, n, d, K, cb, sink);
    run<4, 3, 6>("row look-ahead 4 stages, half-row ring 3, centre ring 6", x, n, d, K, cb, sink);
    run<4, 2, 6>("row look-ahead 4 stages, half-row ring 2, centre ring 6", x, n, d, K, cb, sink);
    run<8, 2, 7>("row look-ahead 8 stages, half-row ring 2, centre ring 7", x, n, d, K, cb, sink);
    return 0;
}
