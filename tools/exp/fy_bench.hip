// Experiment harness (not product): the tiled Fisher-Yates kernels of acav_mi.hip, each timed ALONE on synthetic draws
// (in the product they overlap on two streams and slow each other down).  Build: see tools/exp/build.sh.
#include "../../acav100m_amd/csrc/acav_mi.hip"
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

int main(int argc, char **argv)
{
    const int L = argc > 1 ? atoi(argv[1]) : 1000000;
    const int G = FY_GROUP, B = 20, k = 4, dl = 4, reps = 40;
    FyPlan fp;
    fp.build(L);
    printf("L %d  tiles %d cap %d gsh %d capg %d  tile smem %zu\n", L, fp.NT, fp.ecap, fp.gsh, fp.capg, fp.tile_smem());
    std::mt19937 rng(3);
    const size_t ndraw = (size_t)2 * G * L + 1024;
    std::vector<unsigned> draws(ndraw);
    for (auto &d : draws) d = rng();
    TileChunk c{};
    unsigned *dring; CK(hipMalloc(&dring, 4 * ndraw)); CK(hipMemcpy(dring, draws.data(), 4 * ndraw, hipMemcpyHostToDevice));
    c.ring = dring; c.head = 0; c.ring_words = 1ll << 62;
    unsigned short *dtab; CK(hipMalloc(&dtab, 2 * fp.table.size())); CK(hipMemcpy(dtab, fp.table.data(), 2 * fp.table.size(), hipMemcpyHostToDevice));
    int *deb; CK(hipMalloc(&deb, 4 * fp.ebound.size())); CK(hipMemcpy(deb, fp.ebound.data(), 4 * fp.ebound.size(), hipMemcpyHostToDevice));
    c.table = dtab; c.ebound = deb;
    const size_t nb = (size_t)G * fp.NT * FY_SHARDS * fp.capg, nc = (size_t)G * fp.NT * FY_SHARDS;
    CK(hipMalloc(&c.bucket, 8 * nb)); CK(hipMalloc(&c.gcount, 4 * nc)); CK(hipMemset(c.gcount, 0, 4 * nc));
    for (int q = 0; q < FY_GROUP; ++q) { CK(hipMalloc(&c.g[q], 4 * (size_t)L)); CK(hipMalloc(&c.src[q], 4 * (size_t)L)); }
    for (int q = 0; q < FY_NBUF; ++q) { CK(hipMalloc(&c.perm[q], 4 * (size_t)L)); CK(hipMemset(c.perm[q], 0, 4 * (size_t)L)); }
    CK(hipMalloc(&c.A[0], 4 * (size_t)(L + B))); CK(hipMalloc(&c.A[1], 4 * (size_t)(L + B)));
    CK(hipMemset(c.A[0], 0, 4 * (size_t)(L + B))); CK(hipMemset(c.A[1], 0, 4 * (size_t)(L + B)));
    CK(hipMalloc(&c.err, 4)); CK(hipMemset(c.err, 0, 4));
    // a tiny MI problem so that the selection in workgroup 0 runs as in the product (V = L ids, all label 0)
    const int D = 2, C = 256, P = 1;
    int *dasg, *dpairs, *dbatch, *dNc, *dac, *dbc; double *dS3, *dphi, *dG; MiScalars *dsc; long long *dS;
    CK(hipMalloc(&dasg, 4 * (size_t)L * D));
    { std::vector<int> a((size_t)L * D); for (auto &v : a) v = rng() % C; CK(hipMemcpy(dasg, a.data(), 4 * a.size(), hipMemcpyHostToDevice)); }
    int hp[2] = {0, 1}; CK(hipMalloc(&dpairs, 8)); CK(hipMemcpy(dpairs, hp, 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&dbatch, 4 * 1024)); CK(hipMemset(dbatch, 0, 4 * 1024)); CK(hipMalloc(&c.tailinv, 4 * FY_NBUF * SEL_MAXB)); CK(hipMemset(c.tailinv, 0, 4 * FY_NBUF * SEL_MAXB)); CK(hipMalloc(&dNc, 4 * C * C)); CK(hipMalloc(&dac, 4 * C)); CK(hipMalloc(&dbc, 4 * C));
    CK(hipMemset(dNc, 0, 4 * C * C)); CK(hipMemset(dac, 0, 4 * C)); CK(hipMemset(dbc, 0, 4 * C));
    CK(hipMalloc(&dS3, 8 * 3)); CK(hipMemset(dS3, 0, 24));
    { std::vector<double> phi((size_t)L + 2, 0.0); for (size_t i = 1; i < phi.size(); ++i) phi[i] = (double)i * log((double)i);
      CK(hipMalloc(&dphi, 8 * phi.size())); CK(hipMemcpy(dphi, phi.data(), 8 * phi.size(), hipMemcpyHostToDevice)); }
    CK(hipMalloc(&dsc, sizeof(MiScalars))); CK(hipMemset(dsc, 0, sizeof(MiScalars)));
    const int iters = 64;
    CK(hipMalloc(&dS, 8 * iters * k)); CK(hipMalloc(&dG, 8 * iters * k));
    c.asg = dasg; c.pairs = dpairs; c.batch = dbatch; c.Nc = dNc; c.ac = dac; c.bc = dbc; c.SN = dS3; c.Sa = dS3 + 1; c.Sb = dS3 + 2;
    c.phi = dphi; c.sc = dsc; c.S = dS; c.G = dG;
    c.L0 = L; c.iters = iters; c.ntab = (int)fp.table.size(); c.gsh = fp.gsh; c.NT = fp.NT; c.capg = fp.capg; c.ecap = fp.ecap_lds; c.wcap = fp.wcap;
    c.D = D; c.C = C; c.P = P;
    TileChunk *dcd; CK(hipMalloc(&dcd, sizeof(c))); CK(hipMemcpy(dcd, &c, sizeof(c), hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fy_tile_multi), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fp.tile_smem()));
    const bool staged = getenv("FYB_PART_DIRECT") == nullptr;
    const size_t part_smem = fy_part_smem(fp.NT, fp.table.size(), staged);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fy_part_multi<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fy_part_smem(fp.NT, fp.table.size(), true)));
    printf("part: %s, %zu B of LDS\n", staged ? "staged" : "direct", part_smem);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto part = [&]() { if (staged) hipLaunchKernelGGL(k_fy_part_multi<true>, dim3((L + FYA_CH - 1) / FYA_CH, 1, G), dim3(FYA_THREADS), part_smem, 0, dcd, 0, dl); else hipLaunchKernelGGL(k_fy_part_multi<false>, dim3((L + FYA_CH - 1) / FYA_CH, 1, G), dim3(FYA_THREADS), part_smem, 0, dcd, 0, dl); };
    auto tile = [&]() { hipLaunchKernelGGL(k_fy_tile_multi, dim3(fp.NT, 1, G), dim3(FYT_THREADS), fp.tile_smem(), 0, dcd, 0, dl); };
    auto resolve = [&]() { hipLaunchKernelGGL(k_fy_resolve_multi, dim3((L + 255) / 256, 1, G), dim3(256), 0, 0, dcd, 0, dl, B - k); };
    auto timeit = [&](const char *name, auto fn, double base_us) {
        for (int i = 0; i < 3; ++i) fn();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) fn();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1000.0 / reps;
        printf("%-34s %8.2f us per group  (%.2f us per iteration; minus previous: %.2f)\n", name, us, us / G, (us - base_us) / G);
        return us;
    };
    const double t_ms = timeit("memset counters", [&]() { CK(hipMemsetAsync(c.gcount, 0, 4 * nc, 0)); }, 0);
    const double t_p = timeit("memset + part", [&]() { CK(hipMemsetAsync(c.gcount, 0, 4 * nc, 0)); part(); }, t_ms);
    if (getenv("FYB_ONLY_PART")) return 0;  // -DACAV_FY_ABL_BKSEQ leaves buckets the tile kernel must not read
    const double t_pt = timeit("memset + part + tile", [&]() { CK(hipMemsetAsync(c.gcount, 0, 4 * nc, 0)); part(); tile(); }, t_p);
    timeit("part + tile + resolve", [&]() { part(); tile(); resolve(); }, t_pt - t_ms);
    (void)t_ms;
#ifdef ACAV_FY_PROF
    {   // phase ticks of k_fy_tile, one group
        unsigned long long z16[16] = {0}, pr[16];
        CK(hipMemsetAsync(c.gcount, 0, 4 * nc, 0)); part(); CK(hipDeviceSynchronize());
        CK(hipMemcpyToSymbol(HIP_SYMBOL(fy_prof), z16, sizeof(z16)));
        tile(); CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(fy_prof), sizeof(pr)));
        const double wgs = (double)fp.NT * G;
        const char *nm[5] = {"prologue (bounds, counters)", "LDS init", "entries: bucket -> lists", "walk + src scatter", "g store"};
        for (int q = 0; q < 5; ++q) printf("   tile phase %-28s %7.2f us per workgroup\n", nm[q], pr[q] / wgs / 100.0);
    }
#endif
    unsigned err; CK(hipMemcpy(&err, c.err, 4, hipMemcpyDeviceToHost)); printf("err flags %u\n", err);
    // gathers of iterations 0..G-1, alone (perm / src / g of the last group are in place)
    const int sel_m = sel_mode(B, P, k);
    const size_t sel_smem = sel_layout(B, P, D, k, sel_m).total;
    int itg = 0;
    auto g_sc = [&]() { hipLaunchKernelGGL(k_fy_gather_select_multi, dim3((L + 256 * GS_EPT - 1) / (256 * GS_EPT) + 1, 1), dim3(256), sel_smem, 0, dcd, itg, dl, B, k, sel_m, 1); itg = (itg + 1) % G; };
    timeit("G x gather+select", [&]() { for (int i = 0; i < G; ++i) g_sc(); }, 0);
    {   // iteration 0 only: no selection in workgroup 0 -- the gather alone
        auto g0only = [&]() { hipLaunchKernelGGL(k_fy_gather_select_multi, dim3((L + 256 * GS_EPT - 1) / (256 * GS_EPT) + 1, 1), dim3(256), sel_smem, 0, dcd, 0, dl, B, k, sel_m, 1); };
        timeit("G x gather (launch 0: no selection)", [&]() { for (int i = 0; i < G; ++i) g0only(); }, 0);
    }
    {   // the same with ONE perm buffer for every iteration (stays in cache) -- how much of the gather is the first touch of perm?
        TileChunk c2 = c;
        for (int q = 0; q < FY_NBUF; ++q) c2.perm[q] = c.perm[0];
        CK(hipMemcpy(dcd, &c2, sizeof(c2), hipMemcpyHostToDevice));
        timeit("G x gather+select, one perm buffer", [&]() { for (int i = 0; i < G; ++i) g_sc(); }, 0);
        CK(hipMemcpy(dcd, &c, sizeof(c), hipMemcpyHostToDevice));
    }
    {   // do the two streams of the product overlap?  side group (iterations 8..15) on one stream, 8 gathers (0..7) on another
        const char *pri = getenv("FYB_PRIORITY"), *mask = getenv("FYB_CUMASK");
        hipStream_t sa, sb;
        int lo = 0, hi = 0;
        CK(hipDeviceGetStreamPriorityRange(&lo, &hi));  // lo = least priority (largest number)
        if (mask) {
            const int ncu = atoi(mask);  // side stream restricted to the first ncu CUs of every XCD-interleaved mask word
            uint32_t m[8];
            for (int q = 0; q < 8; ++q) m[q] = 0;
            for (int cu = 0; cu < ncu; ++cu) m[cu / 32] |= 1u << (cu % 32);
            CK(hipExtStreamCreateWithCUMask(&sa, 8, m));
        } else {
            CK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, pri ? lo : 0));
        }
        CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, pri ? hi : 0));
        auto side = [&](hipStream_t st) {
            if (staged) hipLaunchKernelGGL(k_fy_part_multi<true>, dim3((L + FYA_CH - 1) / FYA_CH, 1, G), dim3(FYA_THREADS), part_smem, st, dcd, 8, dl);
            else hipLaunchKernelGGL(k_fy_part_multi<false>, dim3((L + FYA_CH - 1) / FYA_CH, 1, G), dim3(FYA_THREADS), part_smem, st, dcd, 8, dl);
            hipLaunchKernelGGL(k_fy_tile_multi, dim3(fp.NT, 1, G), dim3(FYT_THREADS), fp.tile_smem(), st, dcd, 8, dl);
            hipLaunchKernelGGL(k_fy_resolve_multi, dim3((L + 255) / 256, 1, G), dim3(256), 0, st, dcd, 8, dl, B - k);
        };
        const bool nosel = getenv("FYB_NOSEL") != nullptr;  // every gather as launch 0: no selection beside it
        auto gath = [&](hipStream_t st) {
            for (int i = 0; i < G; ++i)
                hipLaunchKernelGGL(k_fy_gather_select_multi, dim3((L + 256 * GS_EPT - 1) / (256 * GS_EPT) + 1, 1), dim3(256), sel_smem, st, dcd, nosel ? 0 : i, dl, B, k, sel_m, 1);
        };
        auto wall = [&](const char *name, auto fn) {
            fn(); CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) fn();
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            printf("%-44s %8.2f us per group (%.2f us per iteration)\n", name, us, us / G);
        };
        printf("priority range %d (low) .. %d (high)%s%s\n", lo, hi, pri ? "  [side low, gathers high]" : "", mask ? "  [side CU-masked]" : "");
        wall("side group alone (stream a)", [&]() { side(sa); });
        wall("8 gathers alone (stream b)", [&]() { gath(sb); });
        wall("both, two streams", [&]() { side(sa); gath(sb); });
    }
#ifdef ACAV_FY_PROF
    {   // phase ticks of the selection in workgroup 0 of the gather kernel
        unsigned long long z16[16] = {0}, pr[16];
        CK(hipDeviceSynchronize());
        CK(hipMemcpyToSymbol(HIP_SYMBOL(fy_prof), z16, sizeof(z16)));
        for (int i = 0; i < G; ++i) g_sc();
        CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(pr, HIP_SYMBOL(fy_prof), sizeof(pr)));
        const char *nm[5] = {"ids -> LDS (after the gather)", "scoring", "means + top-k", "picks + requeue", "commit"};
        for (int q = 0; q < 5; ++q) printf("   select phase %-30s %7.2f us\n", nm[q], pr[8 + q] / (double)G / 100.0);
    }
#endif
    return 0;
}
