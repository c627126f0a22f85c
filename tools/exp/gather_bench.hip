// Experiment (not product): how fast can the content-dependent step of the MI permutation be?  out[j] = A[perm[j]]
// for a uniformly random perm of L = 1M ints, in the forms the tiled Fisher-Yates could hand it over.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void g1(const int *__restrict__ A, const unsigned *__restrict__ perm, int *__restrict__ out, int L)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < L) out[i] = A[perm[i]];
}
__global__ __launch_bounds__(256) void g1s(const int *__restrict__ A, const unsigned *__restrict__ perm, int *__restrict__ out, int L)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < L) { int v = A[perm[i]]; if (i >= 20) out[i - 20] = v; }
}
struct Desc { long long pad[8]; const int *A[2]; const unsigned *perm[48]; int *out[2]; int L0, iters; };
__global__ __launch_bounds__(256) void g1d(const Desc *__restrict__ cd, int it)
{
    const Desc &c = cd[blockIdx.y];
    const int L = c.L0 - it * 4;
    int i = blockIdx.x * 256 + threadIdx.x;
    if (it >= c.iters || i >= L) return;
    int v = c.A[it & 1][c.perm[it % 48][i]];
    if (i >= 20) c.out[(it + 1) & 1][i - 20] = v;
}
__global__ __launch_bounds__(256) void g1l(const int *__restrict__ A, const unsigned *__restrict__ perm, int *__restrict__ out, int L)
{
    __shared__ int pad[2560];  // 10 KB of static LDS per workgroup, as the product kernel carries
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0x7fffffff) pad[threadIdx.x] = i;
    if (i < L) out[i] = A[perm[i]] + (i == 0x7ffffffe ? pad[5] : 0);
}
// the current form: 37 % E-refs resolved through g
__global__ __launch_bounds__(256) void g0(const int *__restrict__ A, const unsigned *__restrict__ src, const int *__restrict__ g,
                                          int *__restrict__ out, int L)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < L) {
        unsigned s = src[i];
        int a = s & 0x7fffffff;
        if (s >> 31) { int ga; while ((ga = g[a]) >= 0) a = ga; }
        out[i] = A[a];
    }
}
template <int NT>
__global__ __launch_bounds__(256) void g4(const int *__restrict__ A, const unsigned *__restrict__ perm, int *__restrict__ out, int L)
{
    int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < L) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        typedef int i4 __attribute__((ext_vector_type(4)));
        u4 p = NT ? __builtin_nontemporal_load(reinterpret_cast<const u4 *>(perm + i)) : *reinterpret_cast<const u4 *>(perm + i);
        i4 v;
        v.x = A[p.x]; v.y = A[p.y]; v.z = A[p.z]; v.w = A[p.w];
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<i4 *>(out + i)); else *reinterpret_cast<i4 *>(out + i) = v;
    } else for (; i < L; ++i) out[i] = A[perm[i]];
}
__global__ __launch_bounds__(256) void sc1(const int *__restrict__ A, const unsigned *__restrict__ inv, int *__restrict__ out, int L)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < L) out[inv[i]] = A[i];
}
__global__ __launch_bounds__(256) void sc4(const int *__restrict__ A, const unsigned *__restrict__ inv, int *__restrict__ out, int L)
{
    int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < L) {
        uint4 p = *reinterpret_cast<const uint4 *>(inv + i);
        int4 v = *reinterpret_cast<const int4 *>(A + i);
        out[p.x] = v.x; out[p.y] = v.y; out[p.z] = v.z; out[p.w] = v.w;
    } else for (; i < L; ++i) out[inv[i]] = A[i];
}
// XCD-sliced gather: pairs (j, a) sorted by source slice; workgroup b (XCD b % 8) walks slice b % 8
__global__ __launch_bounds__(256) void gx(const int *__restrict__ A, const int2 *__restrict__ pairs, const int *__restrict__ off,
                                          int *__restrict__ out, int per_x)
{
    const int x = blockIdx.x & 7, w = blockIdx.x >> 3;
    const int lo = off[x], hi = off[x + 1];
    for (int i = lo + w * 256 + threadIdx.x; i < hi; i += per_x * 256) {
        int2 p = pairs[i];
        out[p.x] = A[p.y];
    }
}
// same, only the source index list (dest = list order): gather slice-local, write coalesced into a slice-ordered out
__global__ __launch_bounds__(256) void gxs(const int *__restrict__ A, const unsigned *__restrict__ perm, int *__restrict__ out, int L)
{
    // perm sorted so that block b reads sources in slice (b % 8): emulates an L2-local random read + coalesced write
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < L) out[i] = A[perm[i]];
}

int main(int argc, char **argv)
{
    int L = argc > 1 ? atoi(argv[1]) : 1000000;
    int iters = 300;
    std::mt19937 rng(1);
    std::vector<unsigned> perm(L), inv(L), src(L);
    std::vector<int> g(L, -1), A(L);
    for (int i = 0; i < L; ++i) { perm[i] = i; A[i] = i * 7 + 1; }
    std::shuffle(perm.begin(), perm.end(), rng);
    for (int i = 0; i < L; ++i) inv[perm[i]] = i;
    // E-ref emulation: 37 % of entries point to a random step whose g is -1 for 70 %, else one more hop
    for (int i = 0; i < L; ++i) {
        if (rng() % 100 < 37) { src[i] = 0x80000000u | (rng() % L); } else src[i] = perm[i];
        g[i] = (rng() % 100 < 37) ? (int)(rng() % (i + 1)) - 1 : -1;
        if (g[i] >= i) g[i] = -1;
    }
    // slice-sorted pairs
    std::vector<int2> pairs(L); std::vector<int> off(9, 0);
    const int slice = (L + 7) / 8;
    { std::vector<std::vector<int2>> b(8);
      for (int j = 0; j < L; ++j) b[perm[j] / slice].push_back(make_int2(j, (int)perm[j]));
      int o = 0; for (int x = 0; x < 8; ++x) { off[x] = o; for (auto &p : b[x]) pairs[o++] = p; } off[8] = o; }
    // block-interleaved slice perm: block b reads sources from slice b % 8
    std::vector<unsigned> permx(L);
    { std::vector<std::vector<unsigned>> b(8);
      for (int j = 0; j < L; ++j) b[perm[j] / slice].push_back(perm[j]);
      size_t pos[8] = {0}; int nb = (L + 255) / 256;
      for (int blk = 0; blk < nb; ++blk) for (int t = 0; t < 256 && blk * 256 + t < L; ++t) {
          int x = blk & 7; if (pos[x] >= b[x].size()) { for (x = 0; x < 8 && pos[x] >= b[x].size(); ++x); }
          permx[blk * 256 + t] = b[x][pos[x]++]; } }
    int *dA, *dB, *dg, *doff; unsigned *dperm, *dinv, *dsrc, *dpermx; int2 *dpairs;
    CK(hipMalloc(&dA, 4 * L)); CK(hipMalloc(&dB, 4 * L)); CK(hipMalloc(&dg, 4 * L)); CK(hipMalloc(&doff, 36));
    CK(hipMalloc(&dperm, 4 * L)); CK(hipMalloc(&dinv, 4 * L)); CK(hipMalloc(&dsrc, 4 * L)); CK(hipMalloc(&dpermx, 4 * L));
    CK(hipMalloc(&dpairs, 8 * L));
    CK(hipMemcpy(dA, A.data(), 4 * L, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, g.data(), 4 * L, hipMemcpyHostToDevice));
    CK(hipMemcpy(dperm, perm.data(), 4 * L, hipMemcpyHostToDevice)); CK(hipMemcpy(dinv, inv.data(), 4 * L, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsrc, src.data(), 4 * L, hipMemcpyHostToDevice)); CK(hipMemcpy(dpermx, permx.data(), 4 * L, hipMemcpyHostToDevice));
    CK(hipMemcpy(dpairs, pairs.data(), 8 * L, hipMemcpyHostToDevice)); CK(hipMemcpy(doff, off.data(), 36, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nb = (L + 255) / 256, nb4 = (L / 4 + 255) / 256;
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 20; ++i) launch(i);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) launch(i);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s L=%d  %.2f us per launch (back to back)\n", name, L, ms * 1000.f / iters);
    };
    run("g0 current (src+g chains)", [&](int i) { hipLaunchKernelGGL(g0, nb, 256, 0, 0, (i & 1) ? dB : dA, dsrc, dg, (i & 1) ? dA : dB, L); });
    run("g1 out[i]=A[perm[i]]", [&](int i) { hipLaunchKernelGGL(g1, nb, 256, 0, 0, (i & 1) ? dB : dA, dperm, (i & 1) ? dA : dB, L); });
    run("g1s shifted store (i-20)", [&](int i) { hipLaunchKernelGGL(g1s, nb, 256, 0, 0, (i & 1) ? dB : dA, dperm, (i & 1) ? dA : dB, L); });
    run("g1l +10 KB static LDS", [&](int i) { hipLaunchKernelGGL(g1l, nb, 256, 0, 0, (i & 1) ? dB : dA, dperm, (i & 1) ? dA : dB, L); });
    { Desc d{}; d.A[0] = dA; d.A[1] = dB; d.out[0] = dA; d.out[1] = dB; for (int q = 0; q < 48; ++q) d.perm[q] = dperm; d.L0 = L; d.iters = 1 << 30;
      Desc *dd; CK(hipMalloc(&dd, sizeof(d))); CK(hipMemcpy(dd, &d, sizeof(d), hipMemcpyHostToDevice));
      run("g1d descriptor + shifted store", [&](int i) { hipLaunchKernelGGL(g1d, dim3(nb, 1), 256, 0, 0, dd, i & 1); }); }
    run("g4 x4", [&](int i) { hipLaunchKernelGGL(g4<0>, nb4, 256, 0, 0, (i & 1) ? dB : dA, dperm, (i & 1) ? dA : dB, L); });
    run("g4 x4 nontemporal", [&](int i) { hipLaunchKernelGGL(g4<1>, nb4, 256, 0, 0, (i & 1) ? dB : dA, dperm, (i & 1) ? dA : dB, L); });
    run("sc1 out[inv[i]]=A[i]", [&](int i) { hipLaunchKernelGGL(sc1, nb, 256, 0, 0, (i & 1) ? dB : dA, dinv, (i & 1) ? dA : dB, L); });
    run("sc4 x4", [&](int i) { hipLaunchKernelGGL(sc4, nb4, 256, 0, 0, (i & 1) ? dB : dA, dinv, (i & 1) ? dA : dB, L); });
    for (int per_x : {32, 64, 128})
        run(per_x == 32 ? "gx sliced pairs 32/xcd" : per_x == 64 ? "gx sliced pairs 64/xcd" : "gx sliced pairs 128/xcd",
            [&](int i) { hipLaunchKernelGGL(gx, 8 * per_x, 256, 0, 0, (i & 1) ? dB : dA, dpairs, doff, (i & 1) ? dA : dB, per_x); });
    run("gxs slice-local read", [&](int i) { hipLaunchKernelGGL(gxs, nb, 256, 0, 0, (i & 1) ? dB : dA, dpermx, (i & 1) ? dA : dB, L); });
    // empty-ish kernel: the launch floor
    run("floor (L=256)", [&](int i) { hipLaunchKernelGGL(g1, 1, 256, 0, 0, dA, dperm, dB, 256); });
    return 0;
}
