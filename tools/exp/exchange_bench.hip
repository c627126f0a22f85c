// Experiment (not product): what does ONE round of the persistent SGD kernels' cross-workgroup exchange cost on an MI355X, and which
// polling structure gets closest to the hardware's store -> load latency?  The product pattern (k_train_persistent_wide, 16 x 16
// form at K = 1024): NWG = 128 workgroups (64 centre groups x 2 row groups); per round every workgroup publishes 16 granules of 8
// bytes {tag | payload} with device-scope stores into ring[cg][row] and then needs ALL 64 x 32 granules of the round.
//   ./exchange_bench [rounds] [work_cycles]
// Modes:
//   0  the product's sweep: lane (row = l & 31, half = l >> 5) loads its 32 granules, waits for all of them, re-reads the missing ones
//   1  the same, but a pass re-reads ALL granules (what "only the missing ones" saves)
//   2  counter hint: after its stores a workgroup adds 1 to one of 8 arrival counters (device-scope atomic, no fence); the sweep polls
//      the 8 counters (one 64-byte line) until they sum to NWG, then reads the granules (tags still checked: a late store -> mode 0 loop)
//   3  ping-pong between TWO workgroups (store -> seen -> store back), for the raw round trip
//   7  a pass re-issues only the load INSTRUCTIONS (one per pair of centre groups) whose 64 granules are not all there yet: a
//      wave-uniform mask (no exec juggling), later passes are short
//   4  every pass re-reads all granules with 16-byte loads (two granules per lane: 16 memory instructions per pass instead of 32)
//   6  two-level: per XCD (blockIdx % 8: round-robin dispatch) ONE leader sweeps memory (8 x 16 KB per pass over the fabric instead
//      of 128 x 16 KB) and republishes the round's granules into a per-XCD buffer with plain stores; the other workgroups of the XCD
//      poll that buffer with sc1 loads (bypass L1, hit the XCD's L2; `sc0` alone is served by a stale L1 line and
//      never ends -- every spin here is BOUNDED since that run: a round that gives up shows as an absurdly fast or slow figure)
// Every workgroup "works" work_cycles (s_sleep) before it publishes; one workgroup in three works 1 500 cycles longer (the update of a
// touched workgroup).  Reported: us per round.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

constexpr int NCG = 64, NRG = 2, NWG = NCG * NRG, RING = 4;
struct Ctl {
    unsigned long long gran[RING][NCG][32];
    unsigned cnt[RING][8][16];  // 8 arrival counters per ring slot, one 64-byte line each
    unsigned long long pp[2][16];
    unsigned dead[16];  // a spin ran out somewhere: every workgroup leaves its round loop (the figure printed is then meaningless)
    unsigned long long xbuf[8][RING][NCG][32];  // mode 6: the round's granules again, one copy per XCD
};

__device__ __forceinline__ void work(int cycles)
{
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}

template <int MODE>
__global__ __launch_bounds__(64) void k_xchg(Ctl *ctl, int rounds, int work_cycles, unsigned long long *sink)
{
    __shared__ float pad[24 * 1024];  // 96 KB: one workgroup per CU, like the product
    if (threadIdx.x == 1000) pad[0] = 1.f;
    const int lane = threadIdx.x;
    const int cg = blockIdx.x % NCG, rg = blockIdx.x / NCG;
    const int srow = lane & 31, half = lane >> 5;
    unsigned long long acc = 0;
    for (int t = 0; t < rounds; ++t) {
        if ((t & 15) == 0 && __hip_atomic_load(&ctl->dead[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        work(work_cycles + ((blockIdx.x + t) % 3 == 0 ? 1500 : 0));
        const unsigned long long tag = (unsigned long long)(t + 1) << 48;
        unsigned long long(*ring)[32] = ctl->gran[t % RING];
        if (lane < 16) __hip_atomic_store(&ring[cg][rg * 16 + lane], tag | (unsigned)(blockIdx.x * 64 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 2) {
            if (lane == 0) __hip_atomic_fetch_add(&ctl->cnt[t % RING][blockIdx.x & 7][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the counters of ring slot t % RING count up over the rounds that use the slot: target = arrivals so far
            const unsigned target = (unsigned)(NWG / 8) * (unsigned)(t / RING + 1);
            for (unsigned spin_ = 0;; ++spin_) {
            if (spin_ >= (1u << 16) || ((spin_ & 255u) == 255u && __hip_atomic_load(&ctl->dead[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(&ctl->dead[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
                unsigned c = target;
                if (lane < 8) c = __hip_atomic_load(&ctl->cnt[t % RING][lane][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(c >= target)) break;
            }
        }
        unsigned long long g[32];
        unsigned need = 0xFFFFFFFFu;
        for (unsigned spin_ = 0;; ++spin_) {
            if (spin_ >= (1u << 16) || ((spin_ & 255u) == 255u && __hip_atomic_load(&ctl->dead[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(&ctl->dead[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
#pragma unroll
            for (int u = 0; u < 32; ++u)
                if ((need >> u) & 1u) g[u] = __hip_atomic_load(&ring[half + 2 * u][srow], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned still = 0;
#pragma unroll
            for (int u = 0; u < 32; ++u)
                if (((need >> u) & 1u) && (g[u] >> 48) != (tag >> 48)) still |= 1u << u;
            if (MODE == 1) {
                if (__all(still == 0)) break;  // need stays all ones: every pass re-reads everything
            } else if (MODE == 7) {
                unsigned um = 0;  // wave-uniform: instruction u still has a lane without this round's tag
#pragma unroll
                for (int u = 0; u < 32; ++u)
                    if (((need >> u) & 1u) && __any((still >> u) & 1u)) um |= 1u << u;
                need = um;
                if (need == 0) break;
            } else {
                need = still;
                if (__all(need == 0)) break;
            }
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) acc += g[u] & 0xffff;
    }
    if (lane == 0) sink[blockIdx.x] = acc;
}

// mode 6 (two-level)
__device__ __forceinline__ unsigned long long load_l2(const unsigned long long *p)
{
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__global__ __launch_bounds__(64) void k_xchg2(Ctl *ctl, int rounds, int work_cycles, unsigned long long *sink, int nrg)
{
    __shared__ float pad[24 * 1024];
    if (threadIdx.x == 1000) pad[0] = 1.f;
    const int lane = threadIdx.x;
    const int cg = blockIdx.x % NCG, rg = blockIdx.x / NCG, xcd = blockIdx.x & 7;
    const bool leader = blockIdx.x < 8;
    if (lane == 0) {  // is block i really on XCD i % 8?  (HW_REG_XCC_ID = hwreg 20, bits 3:0)
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;
        if (xcc != (unsigned)xcd) atomicAdd(&ctl->dead[1], 1u);
    }
    const int rows_per = 32 / nrg;
    const int srow = lane & 31, half = lane >> 5;
    unsigned long long acc = 0;
    for (int t = 0; t < rounds; ++t) {
        if ((t & 15) == 0 && __hip_atomic_load(&ctl->dead[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        work(work_cycles + ((blockIdx.x + t) % 3 == 0 ? 1500 : 0));
        const unsigned long long tag = (unsigned long long)(t + 1) << 48;
        unsigned long long(*ring)[32] = ctl->gran[t % RING];
        unsigned long long(*xb)[32] = ctl->xbuf[xcd][t % RING];
        if (lane < rows_per) __hip_atomic_store(&ring[cg][rg * rows_per + lane], tag | (unsigned)(blockIdx.x * 64 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long g[32];
        if (leader) {
            for (unsigned spin_ = 0;; ++spin_) {
            if (spin_ >= (1u << 16) || ((spin_ & 255u) == 255u && __hip_atomic_load(&ctl->dead[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(&ctl->dead[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
#pragma unroll
                for (int u = 0; u < 32; ++u) g[u] = __hip_atomic_load(&ring[half + 2 * u][srow], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool ok = true;
#pragma unroll
                for (int u = 0; u < 32; ++u) ok = ok && (g[u] >> 48) == (tag >> 48);
                if (__all(ok)) break;
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) xb[half + 2 * u][srow] = g[u];  // plain stores: they stop in this XCD's L2
        } else {
            for (unsigned spin_ = 0;; ++spin_) {
            if (spin_ >= (1u << 16) || ((spin_ & 255u) == 255u && __hip_atomic_load(&ctl->dead[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(&ctl->dead[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
                bool ok = true;
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    unsigned long long v;
                    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(&xb[half + 2 * u][srow]) : "memory");
                    g[u] = v;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < 32; ++u) ok = ok && (g[u] >> 48) == (tag >> 48);
                if (__all(ok)) break;
            }
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) acc += g[u] & 0xffff;
    }
    if (lane == 0) sink[blockIdx.x] = acc;
}


// mode 8: the product's WORKGROUP, not just its sweep -- 256 threads, 132 KB of LDS; after the "FMA phase" waves 1 .. 3 fetch the next round's
// batch rows (dma_kb KB per workgroup, global_load ... lds, 1 KB per instruction) while wave 0 publishes and sweeps (every pass re-reads all),
// a barrier closes the round.  xsrc: 0 = no fetch, 1 = fresh rows every round (first touch comes from HBM, as the product's), 2 = the same
// 128 KB every round (L2 hits: the CU-side share of the interference alone), 3 = fresh rows, fetched AFTER the sweep (exposed, for the bound).
__global__ __launch_bounds__(256) void k_xchg_wg(Ctl *ctl, int rounds, int work_cycles, unsigned long long *sink, const float *xrows, long long xrounds,
                                                 int dma_kb, int xsrc, long long *cyc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_[];
    float *lds = reinterpret_cast<float *>(smem_);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = blockIdx.x % NCG, rg = blockIdx.x / NCG;
    const int srow = lane & 31, half = lane >> 5;
    unsigned long long acc = 0;
    long long c_exch = 0, c_pass = 0, n_pass = 0;
    auto fetch = [&](int t) {
        const float *src = xrows + (xsrc == 2 ? 0 : (size_t)(t % xrounds) * 32768) + (size_t)rg * 16384;  // 32 rows x 1024 floats per round
        for (int it = wave - 1; it < dma_kb; it += 3)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + it * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void *)(lds + it * 256), 16, 0, 0);
    };
    for (int t = 0; t < rounds; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if ((t & 15) == 0 && __hip_atomic_load(&ctl->dead[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        work(work_cycles + ((blockIdx.x + t) % 3 == 0 ? 1500 : 0));
        __syncthreads();
        if (wave > 0 && (xsrc == 1 || xsrc == 2)) fetch(t + 1);
        if (wave == 0) {
            const long long c0 = clock64();
            const unsigned long long tag = (unsigned long long)(t + 1) << 48;
            unsigned long long(*ring)[32] = ctl->gran[t % RING];
            if (lane < 16) __hip_atomic_store(&ring[cg][rg * 16 + lane], tag | (unsigned)(blockIdx.x * 64 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long g[32];
            for (unsigned spin_ = 0;; ++spin_) {
                if (spin_ >= (1u << 16) || ((spin_ & 255u) == 255u && __hip_atomic_load(&ctl->dead[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(&ctl->dead[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                const long long p0 = clock64();
#pragma unroll
                for (int u = 0; u < 32; ++u) g[u] = __hip_atomic_load(&ring[half + 2 * u][srow], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool ok = true;
#pragma unroll
                for (int u = 0; u < 32; ++u) ok = ok && (g[u] >> 48) == (tag >> 48);
                c_pass += clock64() - p0, n_pass += 1;
                if (__all(ok)) break;
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) acc += g[u] & 0xffff;
            c_exch += clock64() - c0;
        }
        if (xsrc == 3) {
            __syncthreads();
            if (wave > 0) fetch(t + 1);
        }
    }
    if (tid == 0) {
        sink[blockIdx.x] = acc;
        cyc[blockIdx.x * 4 + 0] = c_exch, cyc[blockIdx.x * 4 + 1] = c_pass, cyc[blockIdx.x * 4 + 2] = n_pass;
    }
}

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
// mode 4: lane l reads the two granules of rows 2 (l & 15), 2 (l & 15) + 1 of centre group 4 u + (l >> 4): 16 loads of 16 bytes per pass
__global__ __launch_bounds__(64) void k_xchg16(Ctl *ctl, int rounds, int work_cycles, unsigned long long *sink, int nwg_rg)
{
    __shared__ float pad[24 * 1024];
    if (threadIdx.x == 1000) pad[0] = 1.f;
    const int lane = threadIdx.x;
    const int cg = blockIdx.x % NCG, rg = blockIdx.x / NCG;
    const int rows_per = 32 / nwg_rg;
    unsigned long long acc = 0;
    for (int t = 0; t < rounds; ++t) {
        if ((t & 15) == 0 && __hip_atomic_load(&ctl->dead[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        work(work_cycles + ((blockIdx.x + t) % 3 == 0 ? 1500 : 0));
        const unsigned long long tag = (unsigned long long)(t + 1) << 48;
        unsigned long long(*ring)[32] = ctl->gran[t % RING];
        if (lane < rows_per) __hip_atomic_store(&ring[cg][rg * rows_per + lane], tag | (unsigned)(blockIdx.x * 64 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u64x2 g[16];
        for (unsigned spin_ = 0;; ++spin_) {
            if (spin_ >= (1u << 16) || ((spin_ & 255u) == 255u && __hip_atomic_load(&ctl->dead[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(&ctl->dead[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const u64x2 *p = reinterpret_cast<const u64x2 *>(&ring[4 * u + (lane >> 4)][2 * (lane & 15)]);
                u64x2 v;
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
                g[u] = v;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bool ok = true;
#pragma unroll
            for (int u = 0; u < 16; ++u) ok = ok && (g[u].x >> 48) == (tag >> 48) && (g[u].y >> 48) == (tag >> 48);
            if (__all(ok)) break;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += (g[u].x & 0xffff) + (g[u].y & 0xffff);
    }
    if (lane == 0) sink[blockIdx.x] = acc;
}

__global__ __launch_bounds__(64) void k_pingpong(Ctl *ctl, int rounds, unsigned long long *sink)
{
    __shared__ float pad[24 * 1024];
    if (threadIdx.x == 1000) pad[0] = 1.f;
    if (threadIdx.x != 0) return;
    const int me = blockIdx.x;  // 0 or 1
    for (int t = 1; t <= rounds; ++t) {
        if (me == 0) {
            __hip_atomic_store(&ctl->pp[0][0], (unsigned long long)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (unsigned sp_ = 0; sp_ < (1u << 22) && __hip_atomic_load(&ctl->pp[1][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)t; ++sp_) {}
        } else {
            for (unsigned sp_ = 0; sp_ < (1u << 22) && __hip_atomic_load(&ctl->pp[0][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)t; ++sp_) {}
            __hip_atomic_store(&ctl->pp[1][0], (unsigned long long)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    sink[me] = 1;
}

template <int MODE>
static void run(const char *name, Ctl *ctl, unsigned long long *sink, int rounds, int work_cycles)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctl, 0, sizeof(Ctl)));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_xchg<MODE>, dim3(NWG), dim3(64), 0, 0, ctl, rounds, work_cycles, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    unsigned dead = 0;
    CK(hipMemcpy(&dead, ctl->dead, 4, hipMemcpyDeviceToHost));
    printf("%-72s %.2f us per round%s\n", name, best * 1e3 / rounds, dead ? "   (GAVE UP: a spin ran out -- figure meaningless)" : "");
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 20000;
    const int work_cycles = argc > 2 ? atoi(argv[2]) : 3500;
    Ctl *ctl;
    unsigned long long *sink;
    CK(hipMalloc(&ctl, sizeof(Ctl)));
    CK(hipMalloc(&sink, 8 * NCG * 4));
    printf("%d workgroups (%d centre groups x %d row groups), %d rounds, %d cycles of work per round (+1500 in a third of the workgroups)\n", NWG, NCG,
           NRG, rounds, work_cycles);
    run<0>("product sweep (re-read the missing granules)", ctl, sink, rounds, work_cycles);
    run<1>("every pass re-reads all granules", ctl, sink, rounds, work_cycles);
    run<7>("a pass re-issues the load instructions that are still incomplete (uniform mask)", ctl, sink, rounds, work_cycles);
    run<7>("... the same, no work at all", ctl, sink, rounds, 0);
    run<2>("arrival counters as a hint, then the granules", ctl, sink, rounds, work_cycles);
    for (int nrg : {2, 4}) {
        for (int wc : {3500, 0}) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(ctl, 0, sizeof(Ctl)));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_xchg16, dim3(NCG * nrg), dim3(64), 0, 0, ctl, rounds, wc, sink, nrg);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            unsigned dead = 0;
            CK(hipMemcpy(&dead, ctl->dead, 4, hipMemcpyDeviceToHost));
            printf("16-byte loads, every pass re-reads all; %3d workgroups, %4d cycles of work:  %.2f us per round%s\n", NCG * nrg, wc, best * 1e3 / rounds, dead ? "   (GAVE UP)" : "");
        }
    }
    for (int nrg : {2, 4}) {
        for (int wc : {3500, 0}) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(ctl, 0, sizeof(Ctl)));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_xchg2, dim3(NCG * nrg), dim3(64), 0, 0, ctl, rounds, wc, sink, nrg);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            unsigned dead = 0;
            CK(hipMemcpy(&dead, ctl->dead, 4, hipMemcpyDeviceToHost));
            unsigned off_xcd = 0;
            CK(hipMemcpy(&off_xcd, &ctl->dead[1], 4, hipMemcpyDeviceToHost));
            printf("two-level (one leader per XCD sweeps memory, the rest its L2 copy); %3d workgroups, %4d cycles of work:  %.2f us per round%s; %u workgroups "
                   "NOT on XCD blockIdx %% 8\n", NCG * nrg, wc, best * 1e3 / rounds, dead ? "   (GAVE UP)" : "", off_xcd);
        }
    }

    {  // mode 8: the product's workgroup shape with the row fetch under the exchange
        const long long xrounds = 4096;  // 4096 x 128 KB = 512 MB of "batch rows", cycled
        float *xrows;
        long long *cyc;
        CK(hipMalloc(&xrows, (size_t)xrounds * 32768 * 4));
        CK(hipMemset(xrows, 0, (size_t)xrounds * 32768 * 4));
        CK(hipMalloc(&cyc, NWG * 4 * 8));
        CK(hipFuncSetAttribute((const void *)k_xchg_wg, hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024));
        struct { int kb, src; const char *what; } cases[] = {
            {0, 0, "no row fetch"}, {64, 1, "64 KB of FRESH rows under the exchange (the product)"}, {64, 2, "64 KB of L2-resident rows under the exchange"},
            {64, 3, "64 KB of fresh rows AFTER the sweep (exposed)"}, {16, 1, "16 KB of fresh rows under the exchange"}};
        for (auto &c : cases)
            for (int wc : {3500, 0}) {
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0));
                CK(hipEventCreate(&e1));
                float best = 1e30f;
                std::vector<long long> h(NWG * 4);
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemset(ctl, 0, sizeof(Ctl)));
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(k_xchg_wg, dim3(NWG), dim3(256), 132 * 1024, 0, ctl, rounds, wc, sink, xrows, xrounds, c.kb, c.src, cyc);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) {
                        best = ms;
                        CK(hipMemcpy(h.data(), cyc, NWG * 4 * 8, hipMemcpyDeviceToHost));
                    }
                }
                unsigned dead = 0;
                CK(hipMemcpy(&dead, ctl->dead, 4, hipMemcpyDeviceToHost));
                double ex = 0, ps = 0, np = 0;
                for (int i = 0; i < NWG; ++i) ex += h[i * 4], ps += h[i * 4 + 1], np += h[i * 4 + 2];
                printf("workgroup of 4 waves, %-58s %4d cycles of work: %.2f us per round; exchange %.0f cycles, %.2f passes of %.0f cycles%s\n", c.what, wc,
                       best * 1e3 / rounds, ex / NWG / rounds, np / NWG / rounds, ps / (np > 0 ? np : 1), dead ? "   (GAVE UP)" : "");
                fflush(stdout);
            }
    }
    run<0>("product sweep, no work at all (the exchange alone)", ctl, sink, rounds, 0);
    run<1>("every pass re-reads all granules, no work at all", ctl, sink, rounds, 0);
    run<2>("arrival counters, no work at all", ctl, sink, rounds, 0);
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipMemset(ctl, 0, sizeof(Ctl)));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_pingpong, dim3(2), dim3(64), 0, 0, ctl, rounds, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-72s %.2f us per round trip (two store -> load hops)\n", "ping-pong between two workgroups (blocks 0 and 1)", ms * 1e3 / rounds);
    }
    return 0;
}
