// acav_kmeans.hip -- SGD k-means (reference: clustering/code/sgd_clustering.py:10-129) on gfx950.
//
// Kernels (all hand-written HIP for CDNA4, wave64):
//   k_row_norm2      ||v||^2 of every row in the canonical summation order (torch.norm(..)**2, :73-74)
//   k_assign_f32     calc_best over a row partition (:63-79): exact-fp32 distance sweep on the
//                    f32 matrix cores (v_mfma_f32_32x32x2_f32), centroid + row tiles staged through
//                    LDS, fused ||x||^2 / ||c||^2 / under-use discount epilogue, wave-level argmin
//   k_step_dist      the calc_best half of one add() step on a small batch (latency-optimised:
//                    v_mfma_f32_16x16x4_f32, one 16x16 tile per wave)
//   k_step_update    the update half of add() (:113-128): LDS histogram, lr fallback, deterministic
//                    in-order segmented centroid update, ||c||^2 refresh
//
// Bit-exactness contract with oracle/acav_oracle.c ("canonical arithmetic" in its header):
//   dot = 256-column segments, each one sequential fp32 FMA chain (what the f32 MFMA computes,
//         k-ordered), segment sums folded left to right;
//   sumsq = 32 interleaved FMA chains (class = j mod 32) + fixed butterfly tree;
//   every other op is a single correctly-rounded fp32 op, compiled with -ffp-contract=off.
#include <chrono>
#include <thread>

#include "acav_kmeans_shared.h"

namespace {

// --------------------------------------------------------------------------- k_row_norm2
// one half-wave (32 lanes = the 32 canonical chains) per row
__global__ __launch_bounds__(256) void k_row_norm2(const float *__restrict__ v, int rows, int d,
                                                   float *__restrict__ out)
{
    const int lane32 = threadIdx.x & 31;
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
    float p = 0.f;
    if (row < rows) {
        const float *r = v + (size_t)row * d;
        for (int j = lane32; j < d; j += 32) p = __builtin_fmaf(r[j], r[j], p);
    }
    p = p + __shfl_xor(p, 1);
    p = p + __shfl_xor(p, 2);
    p = p + __shfl_xor(p, 4);
    p = p + __shfl_xor(p, 8);
    p = p + __shfl_xor(p, 16);
    if (row < rows && lane32 == 0) out[row] = norm2_from_sumsq(p);
}

// ------------------------------------------------------------------------- SGD step kernels
// One add() = k_step_dist* (labels of the batch) + k_step_update (centre update).  The batch is
// tiny (b = 32 rows in the reference), so both kernels are latency-optimised, not throughput-optimised.
//
// Cross-workgroup argmin: every workgroup folds its (distance, centre) candidates into one 64-bit
// key per batch row with a global atomicMin -- key = orderable(distance) << 32 | centre, so the
// minimum is the lexicographic (distance, first index) minimum whatever the arrival order.
__device__ __forceinline__ unsigned long long pack_key(float v, int k)
{
    unsigned u = __float_as_uint(v);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;  // total order of finite floats as unsigned
    return ((unsigned long long)u << 32) | (unsigned)k;
}
__device__ __forceinline__ float key_value(unsigned long long key)
{
    unsigned u = (unsigned)(key >> 32);
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(u);
}

// The single-chain form (one 256-column segment) with the order of LDS reads and FMAs pinned: left to itself the compiler
// clumps the ds_read_b128s and drains them with lgkmcnt(1) before most groups of FMAs (3 800 cycles for a 1 024-cycle chain,
// measured in k_train_persistent).  A ring of four 8-column groups keeps three groups (12 reads) in flight ahead of the group
// being multiplied; __builtin_amdgcn_sched_group_barrier pins "4 DS reads, then 8 VALU" per group, the waits stay the
// compiler's own (LDS returns in order: lgkmcnt(12)).  Same chain, same order: bit-identical.
// (The same schedule written as inline-asm reads + counted waits was 25 % faster still and WRONG under register pressure: the
// compiler may copy an asm output to another register before the asm wait that makes it valid -- k_train_persistent_wide.)
__device__ __forceinline__ float dot_block_sched(const float *pc, const float *px, int scz, int sxz)
{
    const float *pcu[8], *pxu[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        pcu[u] = pc + ((u << 2) ^ scz);
        pxu[u] = px + ((u << 2) ^ sxz);
    }
    float4 C[4][2], X[4][2];
    float acc = 0.f;
#define ACAV_LDG(slot, g)                                                                                          \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                \
        C[slot][u] = *reinterpret_cast<const float4 *>(pcu[((g) * 2 + u) & 7] + ((((g) * 2 + u) >> 3) << 5));     \
        X[slot][u] = *reinterpret_cast<const float4 *>(pxu[((g) * 2 + u) & 7] + ((((g) * 2 + u) >> 3) << 5));     \
    }
    ACAV_LDG(0, 0)
    ACAV_LDG(1, 1)
    ACAV_LDG(2, 2)
    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
    for (int g = 0; g < 32; ++g) {
        if (g + 3 < 32) { ACAV_LDG((g + 3) & 3, g + 3) }
        const int s = g & 3;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            acc = __builtin_fmaf(C[s][u].x, X[s][u].x, acc);
            acc = __builtin_fmaf(C[s][u].y, X[s][u].y, acc);
            acc = __builtin_fmaf(C[s][u].z, X[s][u].z, acc);
            acc = __builtin_fmaf(C[s][u].w, X[s][u].w, acc);
        }
        if (g + 3 < 32) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // the 4 DS reads of group g + 3 ...
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);                  // ... then the 8 FMAs of group g
    }
#undef ACAV_LDG
    return acc;
}

// Canonical dot over NB resident 256-column blocks (one segment each) of two swizzled LDS rows: NB
// independent FMA chains per lane, interleaved so the VALU is issue-bound instead of latency-bound;
// ds_read_b128 pairs software-pipelined two chunks ahead.  Returns fold(tot_in, seg_0, .., seg_NB-1).
template <int NB>
__device__ __forceinline__ float dot_blocks(const float *pc, const float *px, int scz, int sxz, float tot, bool first)
{
    // chunk tt (4 columns) of a row sits at float offset ((tt ^ s) << 2), s = row & 7: the XOR only touches
    // the low 3 bits of tt, so 8 per-lane base pointers + compile-time offsets address every chunk
#ifndef ACAV_DOT_COMPILER_SCHEDULE
    if constexpr (NB == 1) {
        const float seg = dot_block_sched(pc, px, scz, sxz);
        return first ? seg : tot + seg;
    }
#endif
    const float *pcu[8], *pxu[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        pcu[u] = pc + ((u << 2) ^ scz);
        pxu[u] = px + ((u << 2) ^ sxz);
    }
    float acc[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) acc[q] = 0.f;
    float4 cA[NB][2], xA[NB][2], cB[NB][2], xB[NB][2];
#define ACAV_LD_GROUP(Cq, Xq, g)                                                                              \
    _Pragma("unroll") for (int q = 0; q < NB; ++q) _Pragma("unroll") for (int u = 0; u < 2; ++u) {            \
        Cq[q][u] = *reinterpret_cast<const float4 *>(pcu[((g) * 2 + u) & 7] + q * 256 + ((((g) * 2 + u) >> 3) << 5)); \
        Xq[q][u] = *reinterpret_cast<const float4 *>(pxu[((g) * 2 + u) & 7] + q * 256 + ((((g) * 2 + u) >> 3) << 5)); \
    }
#define ACAV_FMA_GROUP(Cq, Xq)                                                                                \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                           \
        _Pragma("unroll") for (int q = 0; q < NB; ++q) acc[q] = __builtin_fmaf(Cq[q][u].x, Xq[q][u].x, acc[q]); \
        _Pragma("unroll") for (int q = 0; q < NB; ++q) acc[q] = __builtin_fmaf(Cq[q][u].y, Xq[q][u].y, acc[q]); \
        _Pragma("unroll") for (int q = 0; q < NB; ++q) acc[q] = __builtin_fmaf(Cq[q][u].z, Xq[q][u].z, acc[q]); \
        _Pragma("unroll") for (int q = 0; q < NB; ++q) acc[q] = __builtin_fmaf(Cq[q][u].w, Xq[q][u].w, acc[q]); \
    }
    {
        ACAV_LD_GROUP(cA, xA, 0)
#pragma unroll
        for (int g = 0; g < 32; g += 2) {
            ACAV_LD_GROUP(cB, xB, g + 1)
            ACAV_FMA_GROUP(cA, xA)
            if (g + 2 < 32) { ACAV_LD_GROUP(cA, xA, g + 2) }
            ACAV_FMA_GROUP(cB, xB)
        }
    }
#undef ACAV_LD_GROUP
#undef ACAV_FMA_GROUP
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        tot = first ? acc[q] : tot + acc[q];
        first = false;
    }
    return tot;
}

__device__ __forceinline__ float dot_blocks_n(int nblk, const float *pc, const float *px, int scz, int sxz, float tot,
                                              bool first)
{
    switch (nblk) {
        case 1: return dot_blocks<1>(pc, px, scz, sxz, tot, first);
        case 2: return dot_blocks<2>(pc, px, scz, sxz, tot, first);
        case 3: return dot_blocks<3>(pc, px, scz, sxz, tot, first);
        default: return dot_blocks<4>(pc, px, scz, sxz, tot, first);
    }
}

// k_step_dist_dma: grid (ceil(K/8), ceil(b/8)), 4 waves per workgroup (wave w = column block w of a stage, as in
// k_train_persistent: a lone wave can only issue ~1 ds_read_b128 per 20+ cycles).  Lane l: centre l>>3, row l&7.
// The 8 centre rows + 8 batch rows of a stage (up to 1024 columns = 64 KB) are pulled into LDS with
// direct global->LDS DMA (global_load_lds_dwordx4, no VGPR staging); each wave pulls its own 256-column
// block, waits for it, and runs one dependent v_fma_f32 chain per lane over the block's columns in ascending
// order (a wave only reads what its own DMA wrote); wave 0 then folds the per-block sums left to right --
// the canonical segmented dot.  The FMA chain is bitwise the f32 MFMA's, at 4-cycle instead of 10-cycle
// dependent latency per column.
// LDS rows keep their 16-byte chunks XOR-permuted by (row & 7) -- applied to the DMA SOURCE address,
// the LDS image stays lane-linear -- so the 8 rows one ds_read_b128 touches sit in 8 bank groups.
constexpr int SD_NC = 8;
constexpr int SD_NR = 8;
constexpr int SD_DS = 1024;

#define ACAV_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

__global__ __launch_bounds__(256) void k_step_dist_dma(const float *__restrict__ x, int b, int d,
                                                       const float *__restrict__ centers,
                                                       const float *__restrict__ cn,
                                                       const float *__restrict__ counts,
                                                       const float *__restrict__ xn, int K, float thr, float r,
                                                       unsigned long long *__restrict__ keys)
{
    __shared__ __attribute__((aligned(16))) float sT[(SD_NC + SD_NR) * SD_DS];
    __shared__ float sPart[2][4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);  // wave w owns column block w of a stage
    const int kk = lane >> 3;
    const int ii = lane & 7;
    const int kbase = blockIdx.x * SD_NC, rbase = blockIdx.y * SD_NR;
    const bool ragged_rows = (kbase + SD_NC > K) || (rbase + SD_NR > b);

    float acc = 0.f;  // wave 0: the canonical left fold of the 256-column segments
    int par = 0;
    for (int j0 = 0; j0 < d; j0 += SD_DS, par ^= 1) {
        const int ncols = min(SD_DS, d - j0);
        const int nblk = (ncols + 255) >> 8;
        if (wave < nblk) {
            const int blk = wave;
            const bool ragged = ragged_rows || (blk == nblk - 1 && (ncols & 255));
            if (ragged) {
                // slots the DMA will not write must read as 0 (fma(c, 0, acc) == acc)
                for (int i = lane; i < (SD_NC + SD_NR) * 64; i += 64) {
                    const int row = i >> 6, c4 = i & 63;
                    *reinterpret_cast<float4 *>(sT + row * SD_DS + blk * 256 + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int row = 0; row < SD_NC + SD_NR; ++row) {
                const bool is_c = row < SD_NC;
                const int grow = is_c ? kbase + row : rbase + row - SD_NC;
                const bool row_ok = is_c ? grow < K : grow < b;
                const float *src = (is_c ? centers : x) + (size_t)grow * d;
                const int col = j0 + blk * 256 + ((lane ^ (row & 7)) << 2);
                if (row_ok && col < d) {
                    __builtin_amdgcn_global_load_lds(
                        (const __attribute__((address_space(1))) void *)(src + col),
                        (__attribute__((address_space(3))) void *)(sT + row * SD_DS + blk * 256), 16, 0, 0);
                }
            }
            ACAV_WAIT_VMCNT(0);  // this wave's block has landed (a wave only reads what its own DMA wrote)
            sPart[par][wave][lane] = dot_blocks<1>(sT + kk * SD_DS + blk * 256, sT + (SD_NC + ii) * SD_DS + blk * 256,
                                                   kk << 2, ii << 2, 0.f, true);
        }
        __syncthreads();
        if (wave == 0)
            for (int w = 0; w < nblk; ++w) acc = (j0 == 0 && w == 0) ? sPart[par][0][lane] : acc + sPart[par][w][lane];
        // the next stage writes the other sPart buffer; a wave overwrites only its own block of sT
    }
    if (wave != 0) return;
    const int k = kbase + kk, row = rbase + ii;
    unsigned long long key = ~0ull;
    if (k < K && row < b) key = pack_key(dist_epilogue(acc, xn[row], cn[k], counts[k] < thr, r), k);
    unsigned long long o = __shfl_xor(key, 8);
    key = o < key ? o : key;
    o = __shfl_xor(key, 16);
    key = o < key ? o : key;
    o = __shfl_xor(key, 32);
    key = o < key ? o : key;
    if (lane < 8 && rbase + lane < b) atomicMin(&keys[rbase + lane], key);
}

// k_step_dist_mfma: generic fallback (any d, any alignment).  grid (ceil(K/32), ceil(b/32)); 4 waves,
// each one 16x16 tile on v_mfma_f32_16x16x4_f32, operands staged through registers.
constexpr int SD_BK = 64;
constexpr int SD_LD = 66;  // ds_read_b32 of [i][4t+g]: bank = 2i+g (+4t) -> conflict-free per 32 lanes

__global__ __launch_bounds__(256) void k_step_dist_mfma(const float *__restrict__ x, int b, int d,
                                                        const float *__restrict__ centers,
                                                        const float *__restrict__ cn,
                                                        const float *__restrict__ counts,
                                                        const float *__restrict__ xn, int K, float thr, float r,
                                                        unsigned long long *__restrict__ keys)
{
    __shared__ __attribute__((aligned(16))) float sC[32 * SD_LD];
    __shared__ __attribute__((aligned(16))) float sX[32 * SD_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wave & 1, rw = wave >> 1;
    const int l15 = lane & 15, g = lane >> 4;
    const int kbase = blockIdx.x * 32;
    const int rbase = blockIdx.y * 32;
    const bool vec_ok = (d & 3) == 0;
    const int srow = tid >> 3, sq = tid & 7;  // row 0..31; float4 #sq and #sq+8 of the 64-column stage
    const int nchunks = (d + SD_BK - 1) / SD_BK;

    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, tot = {0.f, 0.f, 0.f, 0.f};
    float4 xr[2], cr[2];
    auto issue_loads = [&](int c) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int j = c * SD_BK + (sq + 8 * m) * 4;
            xr[m] = ld4_guard(x + (size_t)(rbase + srow) * d, j, d, rbase + srow < b, vec_ok);
            cr[m] = ld4_guard(centers + (size_t)(kbase + srow) * d, j, d, kbase + srow < K, vec_ok);
        }
    };
    issue_loads(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float *px = sX + srow * SD_LD + (sq + 8 * m) * 4;
            float *pc = sC + srow * SD_LD + (sq + 8 * m) * 4;
            *reinterpret_cast<float2 *>(px) = make_float2(xr[m].x, xr[m].y);
            *reinterpret_cast<float2 *>(px + 2) = make_float2(xr[m].z, xr[m].w);
            *reinterpret_cast<float2 *>(pc) = make_float2(cr[m].x, cr[m].y);
            *reinterpret_cast<float2 *>(pc + 2) = make_float2(cr[m].z, cr[m].w);
        }
        __syncthreads();
        if (c + 1 < nchunks) issue_loads(c + 1);
        const float *pa = sC + (cw * 16 + l15) * SD_LD + g;
        const float *pb = sX + (rw * 16 + l15) * SD_LD + g;
#pragma unroll
        for (int t = 0; t < SD_BK / 4; ++t)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[4 * t], pb[4 * t], acc, 0, 0, 0);
        if ((c & 3) == 3 || c + 1 == nchunks) {  // end of a 256-column segment
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                tot[e] = c < 4 ? acc[e] : tot[e] + acc[e];
                acc[e] = 0.f;
            }
        }
    }
    // D[i][j]: column j = lane&15 (row of x), row i = 4*(lane>>4) + reg (centre)
    const int row = rbase + rw * 16 + l15;
    unsigned long long key = ~0ull;
    if (row < b) {
        const float xnv = xn[row];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = kbase + cw * 16 + 4 * g + e;
            if (k < K) {
                const unsigned long long kc = pack_key(dist_epilogue(tot[e], xnv, cn[k], counts[k] < thr, r), k);
                key = kc < key ? kc : key;
            }
        }
    }
    unsigned long long o = __shfl_xor(key, 16);
    key = o < key ? o : key;
    o = __shfl_xor(key, 32);
    key = o < key ? o : key;
    if (g == 0 && row < b) atomicMin(&keys[row], key);
}

// k_step_update: grid = b blocks.  Block i owns centre best[i] iff no earlier row of the batch has
// the same label; it then applies   c <- c*(1 - n_k*lr) + sum_{rows of the batch with label k, in
// batch order} lr*x   (sgd_clustering.py:120-127; the sum order is torch_scatter's CPU order) and
// refreshes ||c||^2.  Block 0 also folds the batch histogram (LDS) into counts and does the
// lr-fallback bookkeeping (:116-119).  keys_next: the key buffer of the following step, reset here.
struct StepScalars {
    long long fallback;  // self.fallback
    float mean;          // return value of add(): mean of the row minima
    float lr_used;       // the (possibly fallen-back) fp32 lr of the last step
};

constexpr int SU_MAXB = 1024;
constexpr int SU_MAXK = 8192;

__global__ __launch_bounds__(256) void k_step_update(const float *__restrict__ x, int b, int d,
                                                     float *__restrict__ centers, float *__restrict__ cn,
                                                     float *__restrict__ counts, int K,
                                                     const unsigned long long *__restrict__ keys,
                                                     unsigned long long *__restrict__ keys_next,
                                                     const int64_t *__restrict__ forced, double lr,
                                                     StepScalars *__restrict__ sc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int *sBest = reinterpret_cast<int *>(smem_raw);  // [b]
    float *sMin = reinterpret_cast<float *>(sBest + b);   // [b]
    int *sCnt = reinterpret_cast<int *>(sMin + b);         // [K] batch histogram
    int *sFirst = sCnt + K;                                // [K] first batch row of each label
    float *sRow = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(sFirst + K) + 15) & ~uintptr_t(15));  // [d]
    __shared__ int sCmax;
    const int tid = threadIdx.x;
    const int me = blockIdx.x;
    if (tid == 0) sCmax = 0;
    for (int k = tid; k < K; k += blockDim.x) {
        sCnt[k] = 0;
        sFirst[k] = 0x7fffffff;
    }
    for (int i = tid; i < b; i += blockDim.x) {
        if (forced) {
            sBest[i] = (int)forced[i];
            sMin[i] = 0.f;
        } else {
            const unsigned long long key = keys[i];
            sBest[i] = (int)(key & 0xffffffffull);
            sMin[i] = key_value(key);
        }
    }
    __syncthreads();
    for (int i = tid; i < b; i += blockDim.x) {
        atomicAdd(&sCnt[sBest[i]], 1);
        atomicMin(&sFirst[sBest[i]], i);
    }
    __syncthreads();
    int local_max = 0;
    for (int k = tid; k < K; k += blockDim.x) local_max = max(local_max, sCnt[k]);
    if (local_max) atomicMax(&sCmax, local_max);
    __syncthreads();
    const float cmax = (float)sCmax;
    bool fell = false;
    if ((double)cmax * lr >= 1.0) {  // Python float64 comparison (:116)
        lr = 0.5 / (double)cmax;
        fell = true;
    }
    const float lr32 = (float)lr;
    const int mine = sBest[me];
    const int cnt_mine = sCnt[mine];
    const bool first = sFirst[mine] == me;

    if (me == 0) {
        for (int k = tid; k < K; k += blockDim.x)
            if (sCnt[k]) counts[k] = counts[k] + (float)sCnt[k];  // exact small integers in fp32
        if (keys_next)
            for (int i = tid; i < b; i += blockDim.x) keys_next[i] = ~0ull;
        if (tid < 64) {  // mean of the row minima (add()'s return value; float64 sum, order-free to 1e-12)
            double s = 0.0;
            for (int i = tid; i < b; i += 64) s += (double)sMin[i];
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt <<= 1) s += __shfl_xor(s, dlt);
            if (tid == 0) {
                if (fell) sc->fallback += 1;
                sc->lr_used = lr32;
                sc->mean = (float)(s / (double)b);
            }
        }
    }
    if (!first) return;  // uniform per block

    const float f = 1.0f - (float)cnt_mine * lr32;
    float *crow = centers + (size_t)mine * d;
    // rows of the batch with this label, ascending (torch_scatter's CPU order): compacted once into sBest's slot
    // range [0, cnt_mine) of sList so the update loop below can issue its row loads ahead of the ordered adds
    int *sList = reinterpret_cast<int *>(sMin);  // sMin is dead from here on (block 0 has folded it)
    __syncthreads();
    {
        __shared__ int sWaveCnt[4];
        int base = 0;
        for (int i0 = 0; i0 < b; i0 += 256) {  // uniform trip count
            const int i = i0 + tid;
            const bool m = i < b && sBest[i] == mine;
            const unsigned long long bal = __ballot(m);
            const int lane = tid & 63, wv = tid >> 6;
            if (lane == 0) sWaveCnt[wv] = __popcll(bal);
            __syncthreads();
            int off = base;
            for (int q = 0; q < wv; ++q) off += sWaveCnt[q];
            if (m) sList[off + __popcll(bal & ((1ull << lane) - 1ull))] = i;
            base += sWaveCnt[0] + sWaveCnt[1] + sWaveCnt[2] + sWaveCnt[3];
            __syncthreads();
        }
    }
    if ((d & 3) == 0) {
        for (int j = tid * 4; j < d; j += blockDim.x * 4) {
            const float4 c4 = *reinterpret_cast<const float4 *>(crow + j);
            float4 dl = make_float4(0.f, 0.f, 0.f, 0.f);
            bool have = false;
            for (int q0 = 0; q0 < cnt_mine; q0 += 4) {
                float4 xv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)  // up to 4 independent row loads in flight, then the ordered adds
                    if (q0 + u < cnt_mine) xv[u] = *reinterpret_cast<const float4 *>(x + (size_t)sList[q0 + u] * d + j);
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (q0 + u < cnt_mine) {
                        const float4 v = make_float4(xv[u].x * lr32, xv[u].y * lr32, xv[u].z * lr32, xv[u].w * lr32);
                        dl = have ? make_float4(dl.x + v.x, dl.y + v.y, dl.z + v.z, dl.w + v.w) : v;
                        have = true;
                    }
            }
            const float4 nv = make_float4(c4.x * f + dl.x, c4.y * f + dl.y, c4.z * f + dl.z, c4.w * f + dl.w);
            *reinterpret_cast<float4 *>(crow + j) = nv;
            *reinterpret_cast<float4 *>(sRow + j) = nv;
        }
    } else {
        for (int j = tid; j < d; j += blockDim.x) {
            float delta = 0.f;
            bool have = false;
            for (int i = me; i < b; ++i) {
                if (sBest[i] != mine) continue;
                const float v = x[(size_t)i * d + j] * lr32;
                delta = have ? (delta + v) : v;
                have = true;
            }
            const float nv = crow[j] * f + delta;
            crow[j] = nv;
            sRow[j] = nv;
        }
    }
    __syncthreads();
    if (tid < 32) {
        float p = 0.f;
#pragma unroll 8
        for (int j = tid; j < d; j += 32) p = __builtin_fmaf(sRow[j], sRow[j], p);
        p = p + __shfl_xor(p, 1);
        p = p + __shfl_xor(p, 2);
        p = p + __shfl_xor(p, 4);
        p = p + __shfl_xor(p, 8);
        p = p + __shfl_xor(p, 16);
        if (tid == 0) cn[mine] = norm2_from_sumsq(p);
    }
}

// --------------------------------------------------------------------- k_train_persistent
// Many consecutive add() steps in ONE launch ("owner computes"): workgroup (cg, rg) = 4 waves that
// OWN centres [8cg, 8cg+8) -- resident in LDS for the whole launch, replicated over the row groups --
// and label batch rows [8rg, 8rg+8) of every step.  Per step:
//   1. the step's 8 rows are already in LDS (prefetched by LDS-DMA during the previous step);
//      issue the DMA of the next step's rows into the other buffer
//   2. wave w runs the canonical FMA chain of column block (= segment) w for all 64 (centre, row) pairs;
//      wave 0 folds the four partial sums left to right
//   3. exchange: wave 0 publishes ONE tagged 8-byte granule per (own centre group, own row) -- {step tag,
//      local centre, orderable distance} -- with a single device-scope store into a ring of TP_RING steps, then
//      sweeps the granules of all centre groups for the 32 rows with device-scope loads until every tag is this
//      step's (bounded; err flag instead of a hang) and takes the lexicographic (distance, centre) minimum
//   4. every replica applies the identical in-order update to the centres it owns (rows fetched from
//      global: read-only data), refreshes ||c||^2 (deferred under the next step's FMA phase) and its usage
//      counts, all in LDS
// No plain-store data crosses workgroups inside the launch: the granules are device-scope atomics (one writer
// each, tag and payload in the same 8 bytes), x is read-only, so no L2 write-back / L1 invalidate is needed
// (MI355X_MICROARCH.md, inter-workgroup visibility).  All workgroups must be co-resident: the host checks the
// grid against hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs, every spin is bounded, and on `err` the host
// re-runs the call on the per-step launch path from the saved state.
// PROF (ACAV_PROFILE_STEPS=1): per-phase shader-clock timers; compiled out of the default kernel.
constexpr int TP_NC = 8;
constexpr int TP_NR = 8;
constexpr int TP_DS = 1024;
constexpr int TP_MAXB = 32;
constexpr unsigned TP_SPIN_LIMIT = 1u << 24;
#ifndef ACAV_TP_FIRST_SLEEP
#define ACAV_TP_FIRST_SLEEP 32
#endif
constexpr int TP_FIRST_SWEEP_PAUSE = ACAV_TP_FIRST_SLEEP;  // x 64 clocks between a workgroup's publish and its first sweep
#ifndef ACAV_WIDE_PIN
// pinned MFMA / LDS-read order (dot_tile_mfma<true>) in the wide and column-split kernels too?  No: same-box A/B (profiles/r06_train_pin_ab.txt)
// K = d = 1024 9.34 vs 8.91 us per step, the pair 9.45 vs 8.99, d = 2048 / K = 512 10.53 vs 10.12 -- phase timers: fma 3.86k vs 3.47k cycles;
// in these kernels the compiler's own windows do better than the pinned ones, in k_train_persistent the pinned order wins (3.38k -> 2.98k)
#define ACAV_WIDE_PIN 0
#endif
#ifndef ACAV_SWEEP_REREAD_ALL
#define ACAV_SWEEP_REREAD_ALL 1
#endif
#ifndef ACAV_TPW_FIRST_SLEEP
#define ACAV_TPW_FIRST_SLEEP 0
#endif
#ifndef ACAV_TPW_PASS_SLEEP
#define ACAV_TPW_PASS_SLEEP 0
#endif

#if defined(ACAV_WIDE_NO_MFMA) && !defined(ACAV_EXPERIMENT_BUILD)
#error "ACAV_WIDE_NO_MFMA is an A/B switch of experiment builds (add -DACAV_EXPERIMENT_BUILD)"
#endif
#if defined(ACAV_WIDE_PROF) && !defined(ACAV_EXPERIMENT_BUILD)
#error "ACAV_WIDE_PROF is a diagnostic switch of experiment builds (add -DACAV_EXPERIMENT_BUILD)"
#endif
constexpr int TP_RING = 4;
// Where a granule lives.  Every workgroup's sweep reads EVERY centre group's granules of the step: 128-256 readers of the same 128
// lines at the same moment, and a sweep pass takes 2-2.6 us -- 8 x the idle latency of a device-scope load -- whatever the number of
// granules it re-reads.  Round 6 tested whether that is queueing at the few memory channels that own a contiguous 16 KB ring slot:
// with each 128-byte line of 16 granules (rows 16 h .. 16 h + 15 of one centre group) 4 KB + 128 B from the next one (another page
// AND another line offset: different channels for any power-of-two interleave between 128 B and 4 KB; -DACAV_TP_LSTRIDE=528) the
// K = 1024 forms got SLOWER (9.02 vs 8.66 us per step alone, the pair 9.12 vs 8.77), K = 256 the same (6.32 vs 6.39): not a channel
// hot spot -- 128 pages instead of 4 cost more than the spreading gains.  Contiguous (16) stays.
#ifndef ACAV_TP_LSTRIDE
#define ACAV_TP_LSTRIDE 16
#endif
constexpr int TP_LSTRIDE = ACAV_TP_LSTRIDE;  // granules from one 16-granule line to the next (16 = contiguous, rounds 1-5)
constexpr int TP_MAXCG = 128;                // centre groups a sweep can address (the launch conditions keep to 64)
static_assert(TP_LSTRIDE >= 16, "a line holds 16 granules");
__device__ __forceinline__ int tp_gran_index(int cg, int row) { return (cg * 2 + (row >> 4)) * TP_LSTRIDE + (row & 15); }
struct TrainCtl {
    unsigned err;          // 1 = a bounded spin gave up
    unsigned pad[3];
    unsigned long long prof[8];  // shader-clock cycles per phase, summed over the steps of workgroup (1,0)
    unsigned long long prof_wg[256][8];  // the same per workgroup (ACAV_PROFILE_STEPS diagnostics)
    // granules[ring][centre group][batch row]: {tag:16 | local centre:16 | orderable distance:32}, each
    // written by exactly one workgroup per synced step with ONE 8-byte device-scope store
    unsigned long long gran[TP_RING][TP_MAXCG * 2 * TP_LSTRIDE];
};

#ifndef ACAV_SWEEP_LEAN
#define ACAV_SWEEP_LEAN 1
#endif
#ifndef ACAV_LEAN_PASS_SLEEP
#define ACAV_LEAN_PASS_SLEEP 0
#endif
// Round 6, late: the instruction stream of the exchange, not the fabric, was half of its cost.  Phase timers inside wave 0 of the
// 16 x 16 form at K = d = 1024 (-DACAV_WIDE_PROF): keys epilogue 2.9k cycles (4 x nblk DEPENDENT LDS reads, an s_waitcnt between each),
// a sweep pass ~1.3k cycles to issue its 32 loads (one spilled scalar condition + branch per load) + the round trip + ~1.5k to
// check the tags (the granules sat in AGPRs: 64 v_accvgpr_read, then v_cmp_eq_u64 + mask arithmetic per granule), and 3.1k from the
// last pass to the closing barrier (32 lexicographic 64-bit minima) -- against ~2.1k per pass for the same loads written
// straight-line (tools/exp/exchange_bench.hip, mode 8).  tp_fold_parts / tp_sweep_lean are those three pieces rewritten:
//   * the segment sums are loaded four (x NE chains) at a time, then folded in the canonical left-to-right order;
//   * a pass is NU unconditional loads (NU = 4 .. 32, chosen once per launch; groups past ncg read unused slots of the ring),
//     one XOR + OR per granule and a single ballot: done <=> every needed tag is this step's;
//   * the minimum is a strict `<` scan on the 32-bit distance in ascending group order (the first group of equal distances wins =
//     the smaller index, as the 64-bit key compares), the two halves of a row merge through the 64-bit key as before.  A row whose
//     best distance is 0xFFFFFFFF (no candidate, or NaN rows) is re-done with the 64-bit form -- never on finite data.
// min(key, key of lane ^ 32 / ^ 16 / ^ 8) without the LDS: v_permlane32_swap / v_permlane16_swap (gfx950) put the lower and the upper
// half (the even and the odd 16-lane rows) of a register side by side in two registers of EVERY lane, a row rotation by 8 is a DPP
// modifier -- against two ds_bpermute + address arithmetic + an LDS round trip per 64-bit __shfl_xor, three to four times per step on
// the slowest publisher's chain (tools/exp/permlane_probe.hip checks the lane mapping against __shfl_xor on the device).
typedef unsigned tp_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned long long tp_min_xor32(unsigned long long key)
{
    const tp_u32x2 a = __builtin_amdgcn_permlane32_swap((unsigned)key, (unsigned)key, false, false);
    const tp_u32x2 b = __builtin_amdgcn_permlane32_swap((unsigned)(key >> 32), (unsigned)(key >> 32), false, false);
    const unsigned long long k0 = ((unsigned long long)b[0] << 32) | a[0], k1 = ((unsigned long long)b[1] << 32) | a[1];
    return k1 < k0 ? k1 : k0;  // k0 = the key of lane & 31, k1 = of lane | 32, in both lanes
}
__device__ __forceinline__ unsigned long long tp_min_xor16(unsigned long long key)
{
    const tp_u32x2 a = __builtin_amdgcn_permlane16_swap((unsigned)key, (unsigned)key, false, false);
    const tp_u32x2 b = __builtin_amdgcn_permlane16_swap((unsigned)(key >> 32), (unsigned)(key >> 32), false, false);
    const unsigned long long k0 = ((unsigned long long)b[0] << 32) | a[0], k1 = ((unsigned long long)b[1] << 32) | a[1];
    return k1 < k0 ? k1 : k0;  // the keys of the even and of the odd row of the pair
}
__device__ __forceinline__ unsigned long long tp_min_xor8(unsigned long long key)
{
    const unsigned lo = (unsigned)key, hi = (unsigned)(key >> 32);
    const unsigned plo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0x128, 0xf, 0xf, false);  // row_ror:8 = lane ^ 8 within the row
    const unsigned phi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hi, 0x128, 0xf, 0xf, false);
    const unsigned long long o = ((unsigned long long)phi << 32) | plo;
    return o < key ? o : key;
}

template <int NE>
__device__ __forceinline__ void tp_fold_parts(const float *p, int se, int sw, int nblk, float acc[NE])
{
#pragma unroll
    for (int e = 0; e < NE; ++e) acc[e] = p[e * se];
    for (int w0 = 1; w0 < nblk; w0 += 4) {  // uniform
        float v[NE][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int w = w0 + j < nblk ? w0 + j : nblk - 1;  // clamped: a valid slot, dropped below
#pragma unroll
            for (int e = 0; e < NE; ++e) v[e][j] = p[e * se + w * sw];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (w0 + j < nblk) {
#pragma unroll
                for (int e = 0; e < NE; ++e) acc[e] = acc[e] + v[e][j];
            }
    }
}

// wave-level: returns 0 if a bounded spin gave up (ctl->err set); *best = the row's label for lanes < b (both halves hold it)
template <int NU>
__device__ __forceinline__ unsigned tp_sweep_lean(const unsigned long long *ring, int ncg, int b, int ncw, unsigned tag16, int lane, TrainCtl *ctl,
                                                  int *best, unsigned *npass)
{
    const int srow = lane & 31, half = lane >> 5;
    const unsigned long long *p = ring + tp_gran_index(half, srow);  // granule u of this lane: group half + 2 u = p + 4 u TP_LSTRIDE
    const int nval = srow < b ? (ncg - half + 1) >> 1 : 0;         // this lane needs u < nval
    const bool allvalid = __all(nval == NU);
    const unsigned T = tag16 << 16;
    unsigned hi[NU], lo[NU];
    unsigned ok = 1;
    for (unsigned spins = 0;; ++spins) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const unsigned long long v = __hip_atomic_load(p + u * 4 * TP_LSTRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lo[u] = (unsigned)v, hi[u] = (unsigned)(v >> 32);
        }
        unsigned x = 0;
        if (allvalid) {
#pragma unroll
            for (int u = 0; u < NU; ++u) x |= hi[u] ^ T;
        } else {
#pragma unroll
            for (int u = 0; u < NU; ++u) x |= u < nval ? hi[u] ^ T : 0u;
        }
        *npass += 1;
        if (__all((x >> 16) == 0u)) break;
#if ACAV_LEAN_PASS_SLEEP > 0
        __builtin_amdgcn_s_sleep(ACAV_LEAN_PASS_SLEEP);  // experiment knob: x 64 clocks before a repeated pass
#endif
        if (spins > TP_SPIN_LIMIT || (spins & 1023) == 1023) {
            if (spins > TP_SPIN_LIMIT || __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                if (lane == 0) __hip_atomic_store(&ctl->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
    }
    unsigned bd = 0xFFFFFFFFu, bh = 0u, bu = 0u;
    if (allvalid) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const bool c = lo[u] < bd;
            bd = c ? lo[u] : bd, bh = c ? hi[u] : bh, bu = c ? (unsigned)u : bu;
        }
    } else {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const bool c = u < nval && lo[u] < bd;
            bd = c ? lo[u] : bd, bh = c ? hi[u] : bh, bu = c ? (unsigned)u : bu;
        }
    }
    unsigned long long key = ~0ull;
    if (bd != 0xFFFFFFFFu) key = ((unsigned long long)bd << 32) | (unsigned)((half + 2 * (int)bu) * ncw + (int)(bh & 0xFFFFu));
    if (__any(bd == 0xFFFFFFFFu && nval > 0)) {  // cold: the lexicographic form of rounds 1-5 (a granule without a candidate, distance bits all ones)
        key = ~0ull;
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if (u < nval) {
                const unsigned loc = hi[u] & 0xFFFFu;
                const unsigned long long cand = loc == 0xFFFFu ? ~0ull : (((unsigned long long)lo[u] << 32) | (unsigned)((half + 2 * u) * ncw + (int)loc));
                key = cand < key ? cand : key;
            }
    }
    if (!ok) key = ~0ull;
    key = tp_min_xor32(key);
    *best = (int)(key & 0xffffffffull);
    return ok;
}

__device__ __forceinline__ int tp_off(int row, int j)
{
    return row * TP_DS + (j & ~255) + (((((j >> 2) & 63) ^ (row & 7))) << 2) + (j & 3);
}

// DMA of column block `blk` of 8 rows (clamped duplicates for ragged groups are never used)
template <bool RAGGED>
__device__ __forceinline__ void tp_dma_block(float *lds, const float *__restrict__ src, int first_row, int nrows_valid,
                                             int d, int blk, int lane)
{
#pragma unroll
    for (int row = 0; row < 8; ++row) {
        const int grow = first_row + (row < nrows_valid ? row : 0);
        const int col = blk * 256 + ((lane ^ (row & 7)) << 2);
        const float *g = src + (size_t)grow * d + col;
        // ragged last block (d % 256 != 0): the slots past column d are never written and keep their initial zeros
        if (!RAGGED || col < d)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(lds + row * TP_DS + blk * 256), 16, 0, 0);
    }
}

// ||c||^2 refresh of the centres flagged in `pend`: half-wave (2*wave + lane>>5) takes centre c8 = that index
// (32 canonical chains + butterfly).  Called by every wave; only reads sC rows, writes sCn[c8].
__device__ __forceinline__ void tp_refresh_norms(unsigned pend, const float *sC, float *sCn, int wave, int lane, int d)
{
    const int c8 = 2 * wave + (lane >> 5), q = lane & 31;
    const bool mine = (pend >> c8) & 1u;
    float p = 0.f;
    if (mine) {
        // column j = q + 32u of row c8 sits at  c8*DS + (((q>>2) ^ (c8&7)) << 2) + (q&3) + 32u : base + constant stride
        const float *base = sC + c8 * TP_DS + ((((q >> 2) ^ (c8 & 7))) << 2) + (q & 3);
        const int nu = (d + 31) >> 5;  // columns past d read the zero padding
#pragma unroll 32
        for (int u = 0; u < nu; ++u) {
            const float v = base[u * 32];
            p = __builtin_fmaf(v, v, p);
        }
    }
    p = p + __shfl_xor(p, 1);
    p = p + __shfl_xor(p, 2);
    p = p + __shfl_xor(p, 4);
    p = p + __shfl_xor(p, 8);
    p = p + __shfl_xor(p, 16);
    if (q == 0 && mine) sCn[c8] = norm2_from_sumsq(p);
}

// k_step_dist_dma for LARGE batches (the DDP path steps global batches of 32 W rows; d <= 1024): one workgroup keeps its 8
// centres in LDS and walks RG groups of 8 batch rows against them, the next group's rows in flight (LDS-DMA, double
// buffered) while the current one runs its FMA chains.  The one-group-per-workgroup kernel re-loads the centres for every
// 8 rows and needs b / 8 x K / 8 workgroups (1024 at b = 256, K = 256: two rounds of ~8 us of mostly DMA latency); here
// the grid is K / 8 x b / (8 RG) and a round costs ~1.5 + 2 RG us.  Same arithmetic per (centre, row) pair: bit-identical.
template <int RG, bool RAGGED>
__global__ __launch_bounds__(256) void k_step_dist_dma_rg(const float *__restrict__ x, int b, int d, const float *__restrict__ centers,
                                                          const float *__restrict__ cn, const float *__restrict__ counts,
                                                          const float *__restrict__ xn, int K, float thr, float r,
                                                          unsigned long long *__restrict__ keys)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sdrg_smem[];
    float *sC = reinterpret_cast<float *>(sdrg_smem);        // [8][TP_DS]
    float *sX = sC + TP_NC * TP_DS;                           // [2][8][TP_DS]
    __shared__ float sPart[2][4][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);  // wave w owns column block w
    const int kk = lane >> 3, ii = lane & 7;
    const int kbase = blockIdx.x * TP_NC, rbase0 = blockIdx.y * (TP_NR * RG);
    const int nck = min(TP_NC, K - kbase);
    const int nblk = (d + 255) >> 8;
    const bool active = wave < nblk;
    if (RAGGED && active) {  // ragged last block: columns d .. 256 nblk - 1 must read as zero for good
        for (int row = 0; row < TP_NC; ++row) {
            *reinterpret_cast<float4 *>(sC + row * TP_DS + wave * 256 + (lane << 2)) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(sX + row * TP_DS + wave * 256 + (lane << 2)) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(sX + (TP_NR + row) * TP_DS + wave * 256 + (lane << 2)) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // before the DMA below writes the same rows
    }
    // what the epilogue needs is loaded BEFORE the DMA it would otherwise queue behind (vmcnt retires in order)
    float my_cn = 0.f;
    bool my_disc = false;
    if (wave == 0 && kk < nck) {
        my_cn = cn[kbase + kk];
        my_disc = counts[kbase + kk] < thr;
    }
    if (active) {
        tp_dma_block<RAGGED>(sC, centers, kbase, nck, d, wave, lane);
        tp_dma_block<RAGGED>(sX, x, rbase0, min(TP_NR, b - rbase0), d, wave, lane);
    }
#pragma unroll
    for (int gi = 0; gi < RG; ++gi) {
        const int rb = rbase0 + gi * TP_NR;
        if (rb >= b) break;  // uniform
        const int nrv = min(TP_NR, b - rb);
        const bool more = gi + 1 < RG && rb + TP_NR < b;
        float xn_r = 0.f;
        if (wave == 0 && ii < nrv) xn_r = xn[rb + ii];
        if (active) {
            // a wave re-fills only the block it alone reads, and it is done with the buffer the group before last used
            if (more) tp_dma_block<RAGGED>(sX + ((gi + 1) & 1) * TP_NR * TP_DS, x, rb + TP_NR, min(TP_NR, b - rb - TP_NR), d, wave, lane);
            if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // everything but the newest group (8 DMA instructions)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sPart[gi & 1][wave][lane] = dot_blocks<1>(sC + kk * TP_DS + wave * 256, sX + ((gi & 1) * TP_NR + ii) * TP_DS + wave * 256,
                                                      kk << 2, ii << 2, 0.f, true);
        }
        __syncthreads();
        if (wave == 0) {
            float acc = sPart[gi & 1][0][lane];
            for (int w = 1; w < nblk; ++w) acc = acc + sPart[gi & 1][w][lane];  // canonical left fold of the segments
            const int k = kbase + kk;
            unsigned long long key = ~0ull;
            if (kk < nck && ii < nrv) key = pack_key(dist_epilogue(acc, xn_r, my_cn, my_disc, r), k);
            unsigned long long o = __shfl_xor(key, 8);
            key = o < key ? o : key;
            o = __shfl_xor(key, 16);
            key = o < key ? o : key;
            o = __shfl_xor(key, 32);
            key = o < key ? o : key;
            if (lane < nrv) atomicMin(&keys[rb + lane], key);
        }
        // the next group writes the other sPart buffer
    }
}

// 4 waves per workgroup; wave w owns column block (= canonical segment) w of every LDS row: it DMAs it,
// runs the segment's FMA chain for all 64 (centre, row) pairs, and applies the centre update to it.
// One wave can only issue ~1 ds_read_b128 per 20+ cycles, so the 4 waves quadruple the LDS read rate;
// the 4 segment sums are folded in order ((s0+s1)+s2)+s3 -- exactly the canonical dot.
template <bool PIN>
__device__ __forceinline__ f32x4 dot_tile_mfma(const float *pc, const float *px, int sc, int sx);  // (below, with the 16-centre forms)

template <bool RAGGED, bool PROF>  // RAGGED: d % 256 != 0 (guarded DMA / update lanes, zero-padded last block)
__global__ __launch_bounds__(256) void k_train_persistent(
    const float *__restrict__ x, const float *__restrict__ xn, int b, int d, int K, float *__restrict__ centers,
    float *__restrict__ cn, float *__restrict__ counts, const float *__restrict__ thr, double lr0, float r,
    const int64_t *__restrict__ forced, int need, int T, TrainCtl *__restrict__ ctl, StepScalars *__restrict__ sc,
    int nwg)
{
    __shared__ __attribute__((aligned(16))) float sC[TP_NC * TP_DS];
    __shared__ __attribute__((aligned(16))) float sX[2][TP_NR * TP_DS];
    __shared__ float sCn[TP_NC];
    __shared__ float sCnt[TP_NC];
    __shared__ float sPart[4][64];
    __shared__ int sBest[TP_MAXB];
    __shared__ int sDead;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 3, ii = lane & 7;
    const int kbase = blockIdx.x * TP_NC, rbase = blockIdx.y * TP_NR;
    const int nck = min(TP_NC, K - kbase), nrv = min(TP_NR, b - rbase);
    const int nblk = (d + 255) >> 8;
    const int ncg = gridDim.x;
    const bool active = wave < nblk;  // this wave has a column block
    const bool col_ok = !RAGGED || wave * 256 + (lane << 2) < d;  // this lane's 4 columns of the block exist (d % 4 == 0)

    if (RAGGED && active) {  // ragged last block: columns d .. 256 nblk - 1 must read as zero for good
        for (int row = 0; row < TP_NC; ++row) {
            *reinterpret_cast<float4 *>(sC + row * TP_DS + wave * 256 + (lane << 2)) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(sX[0] + row * TP_DS + wave * 256 + (lane << 2)) = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(sX[1] + row * TP_DS + wave * 256 + (lane << 2)) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // before the DMA below writes the same rows
    }
    if (active) tp_dma_block<RAGGED>(sC, centers, kbase, nck, d, wave, lane);
    if (tid < TP_NC) {
        const int k = kbase + (tid < nck ? tid : 0);
        sCn[tid] = cn[k];
        sCnt[tid] = counts[k];
    }
    if (tid == 0) sDead = 0;
    if (need < T && active) tp_dma_block<RAGGED>(sX[need & 1], x + (size_t)need * b * d, rbase, nrv, d, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    unsigned nsync = 0;
    unsigned pend = 0;  // centres whose ||c||^2 is stale: refreshed under the next step's FMA phase
#define TP_CLK() (PROF ? (long long)clock64() : 0ll)
    long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < T; ++t) {
        const float *xb = x + (size_t)t * b * d;
        const long long c0 = TP_CLK();
        if (t < need) {
            __syncthreads();  // every wave has read the previous step's labels (an untouched workgroup has no other barrier)
            if (tid < b) sBest[tid] = (int)forced[(size_t)t * b + tid];
            __syncthreads();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my block of step t's rows landed (issued one step ago)
            float xn_t = 0.f, thr_t = 0.f;
            if (wave == 0) {  // in flight under the FMA chain
                xn_t = xn[(size_t)t * b + rbase + (ii < nrv ? ii : 0)];
                thr_t = thr[t];
            }
            if (t + 1 < T && active) tp_dma_block<RAGGED>(sX[(t + 1) & 1], x + (size_t)(t + 1) * b * d, rbase, nrv, d, wave, lane);
            const long long c1 = TP_CLK();
            pr[0] += c1 - c0;
#ifndef ACAV_WIDE_NO_MFMA
            // round 6: the block's 8 x 8 (centre, row) pairs as one quarter of a 16 x 16 tile on the f32 matrix core (dot_tile_mfma:
            // bit for bit the v_fma chain; tile rows / columns 8 .. 15 repeat 0 .. 7 and are dropped): 64 dependent MFMAs = 2.6k
            // cycles and 128 ds_read_b32 per wave against 3.3k cycles and 2 x 64 ds_read_b128 of the per-lane chain
            if (active) {
                const int i15 = lane & 15, kq = lane >> 4, r8 = i15 & 7;
                const f32x4 q4 = dot_tile_mfma<true>(sC + r8 * TP_DS + wave * 256 + kq, sX[t & 1] + r8 * TP_DS + wave * 256 + kq, r8, r8);
                if (kq < 2 && i15 < 8) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) sPart[wave][(4 * kq + e) * 8 + i15] = q4[e];  // slot (centre kk, row ii) = kk * 8 + ii
                }
            } else
                sPart[wave][lane] = 0.f;
            if (PROF) pr[5] += TP_CLK() - c1;  // the chain alone
#else
            float part = 0.f;
            if (active)
                part = dot_blocks<1>(sC + kk * TP_DS + wave * 256, sX[t & 1] + ii * TP_DS + wave * 256, kk << 2, ii << 2,
                                     0.f, true);
            if (PROF) pr[5] += TP_CLK() - c1;  // the chain alone
            sPart[wave][lane] = part;
#endif
            if (pend) {  // uniform: the previous update's norm refresh, off the update's critical path
                tp_refresh_norms(pend, sC, sCn, wave, lane, d);
                pend = 0;
            }
            __syncthreads();
            const long long c2 = TP_CLK();
            pr[1] += c2 - c1;
            if (wave == 0) {
                const float cn_k = sCn[kk], ct_k = sCnt[kk];  // (with the segment sums: one LDS round trip)
                float acc1[1];
                tp_fold_parts<1>(&sPart[0][lane], 0, 64, nblk, acc1);  // canonical left fold of the segments (loads four at a time)
                const float acc = acc1[0];
                const int k = kbase + kk;
                unsigned long long key = ~0ull;
                {
                    float tq = -2.0f * acc;  // dist_epilogue; the discount's division only in steps where some lane needs it
                    tq = tq + xn_t;
                    tq = tq + cn_k;
                    const bool disc = ct_k < thr_t;
                    if (__any(disc)) {
                        const float q = tq / r;
                        tq = disc ? q : tq;
                    }
                    if (kk < nck && ii < nrv) key = pack_key(tq, k);
                }
                key = tp_min_xor32(tp_min_xor16(tp_min_xor8(key)));  // (the minimum over the 8 centres kk of row ii: lanes ^ 8, ^ 16, ^ 32)
                unsigned long long o;
                (void)o;
                // publish: one tagged granule per (my centre group, my row), a single 8-byte device-scope store
                const unsigned long long tag = (unsigned long long)((nsync % 65535u) + 1u) << 48;
                unsigned long long *ring = ctl->gran[nsync % TP_RING];
                if (lane < nrv) {
                    const unsigned long long local = (key == ~0ull) ? 0xFFFFull : ((key & 0xffffffffull) - (unsigned)kbase);
                    __hip_atomic_store(&ring[tp_gran_index(blockIdx.x, rbase + lane)], tag | (local << 32) | (key >> 32), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
#if ACAV_SWEEP_LEAN
                {  // gather: tp_sweep_lean (above); the pause before the first pass as tuned in round 2 (comment in the #else branch)
                    if (ncg >= 16) __builtin_amdgcn_s_sleep(TP_FIRST_SWEEP_PAUSE);
                    const unsigned tag16 = (nsync % 65535u) + 1u;
                    const int nu = (ncg + 1) >> 1;
                    int bl = -1;
                    unsigned np = 0, ok;
                    if (nu <= 4) ok = tp_sweep_lean<4>(ring, ncg, b, TP_NC, tag16, lane, ctl, &bl, &np);
                    else if (nu <= 8) ok = tp_sweep_lean<8>(ring, ncg, b, TP_NC, tag16, lane, ctl, &bl, &np);
                    else ok = tp_sweep_lean<16>(ring, ncg, b, TP_NC, tag16, lane, ctl, &bl, &np);
                    if (PROF) pr[7] += np;  // sweep passes (diagnostics)
                    if (lane < b) sBest[lane] = bl;
                    if (!ok && lane == 0) sDead = 1;
                }
#else
                // gather: lane (row = l & 31, half = l >> 5) sweeps the granules of its row from half of the
                // centre groups until every tag is this step's; then the lexicographic minimum
                const int srow = lane & 31, half = lane >> 5;
                unsigned long long bestkey = ~0ull;
                unsigned ok = 1;
                constexpr int TP_SW = 16;  // granules per lane: centre groups half, half+2, ... (ncg <= 32 per sweep set)
                unsigned long long g[TP_SW];
                unsigned need = 0;  // bit u: granule u of this lane not yet seen with this step's tag
#pragma unroll
                for (int u = 0; u < TP_SW; ++u)
                    if (srow < b && half + 2 * u < ncg) need |= 1u << u;
                // the first sweep leaves ~0.9 us after the publish: sent at once it nearly always comes back incomplete (the
                // other workgroups' stores are still on their way) and every pass costs a full round trip of ~1.7 us --
                // 2.5 passes per step without the pause, 1.7 with it (7.20 -> 7.00 us per step at K = 256, d = 1024; 24 / 28 /
                // 32 / 36 / 40 / 48 x 64 clocks: 7.08 / 7.01 / 7.00 / 7.02 / 7.08 / 7.24).  With few centre groups the first
                // pass is more often complete and the pause costs more than it saves (K = 64: 7.34 -> 7.43)
                if (ncg >= 16) __builtin_amdgcn_s_sleep(TP_FIRST_SWEEP_PAUSE);
                for (unsigned spins = 0;; ++spins) {
                    // round 6: EVERY pass re-reads every granule of the existing centre groups (a wave-uniform condition).  Re-reading
                    // only the lanes' missing granules -- rounds 1-5 -- put an exec-mask save / branch / restore around each of the
                    // loads and sent nearly as many (sparse) memory instructions: tools/exp/exchange_bench.hip, 128 workgroups,
                    // 3 500 cycles of work per round: 4.85-5.05 us per round against 3.75-3.85 with full re-reads; in this kernel
                    // K = 256 / d = 1024 5.98 -> 5.87 us per step, K = 64 / d = 512 6.58 -> 6.34
#pragma unroll
                    for (int u = 0; u < TP_SW; ++u)
                        if (ACAV_SWEEP_REREAD_ALL ? 2 * u < ncg : (int)((need >> u) & 1u))
                            g[u] = __hip_atomic_load(&ring[tp_gran_index(half + 2 * u, srow)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int u = 0; u < TP_SW; ++u)
                        if (((need >> u) & 1u) && (g[u] >> 48) == (tag >> 48)) need &= ~(1u << u);
                    if (PROF) pr[7] += 1;  // sweep passes (diagnostics)
                    if (__all(need == 0)) break;
                    if (spins > TP_SPIN_LIMIT || (spins & 1023) == 1023) {
                        if (spins > TP_SPIN_LIMIT || __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            if (lane == 0) __hip_atomic_store(&ctl->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = 0;
                            break;
                        }
                    }
                }
                bestkey = ~0ull;
#pragma unroll
                for (int u = 0; u < TP_SW; ++u) {
                    const int cg = half + 2 * u;
                    if (srow < b && cg < ncg && ok) {
                        const unsigned loc = (unsigned)(g[u] >> 32) & 0xFFFFu;
                        const unsigned long long cand =
                            loc == 0xFFFFu ? ~0ull : (((g[u] & 0xffffffffull) << 32) | (unsigned)(cg * TP_NC + loc));
                        bestkey = cand < bestkey ? cand : bestkey;
                    }
                }
                o = __shfl_xor(bestkey, 32);
                bestkey = o < bestkey ? o : bestkey;
                if (lane < b) sBest[lane] = (int)(bestkey & 0xffffffffull);
                if (!ok && lane == 0) sDead = 1;
#endif
            }
            ++nsync;
            __syncthreads();
            pr[2] += TP_CLK() - c2;
            if (sDead) break;  // uniform
        }
        // ---- update: every replica of a centre group does the same arithmetic; wave w owns column block w
        const long long c3 = TP_CLK();
        const int best = (lane < b) ? sBest[lane] : -1;
        double lr = lr0;
        bool fell = false;
        if ((double)b * lr0 >= 1.0) {  // lr fallback (:116-119) possible at all?  (never at the defaults 32 * 0.01)
            int cmaxi = 0;
#pragma unroll
            for (int i = 0; i < TP_MAXB; ++i) {
                const int li = __builtin_amdgcn_readlane(best, i);  // scalar; -1 for rows >= b
                const int c = __popcll(__ballot(lane < b && best == li));
                cmaxi = (i < b && c > cmaxi) ? c : cmaxi;
            }
            if ((double)(float)cmaxi * lr >= 1.0) {
                lr = 0.5 / (double)(float)cmaxi;
                fell = true;
            }
        }
        const float lr32 = (float)lr;
        if (fell && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) sc->fallback += 1;
        // COMPACT update (round 4): the centres that were hit, up to four at a time -- first the first row of each (all loads in
        // flight together: one round trip), then per centre its further rows (rare), the fold and the centre row.  The fully
        // unrolled form (8 centres x their row loops, ~800 straight-line instructions) ran once per step in two thirds of the
        // workgroups out of a cold instruction cache, on the chain of the next step's slowest publisher (the wide kernel's
        // update went from 4.8k to 1.7k cycles per touched step with the same change).  Same arithmetic: rows ascending within a
        // centre, v = x * lr, dl = v0 (+ v1 ...), c = c * f + dl; the order of the centres does not matter.
        const int li = best - kbase;  // this lane's batch row -> local centre (valid: lane < b, 0 <= li < nck)
        unsigned long long rem = __ballot(lane < b && li >= 0 && li < nck);
        if (rem) {  // uniform over the workgroup
            unsigned touched = 0;
            while (rem) {
                int cs[4];
                unsigned long long ms[4];
                float4 x0[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    cs[q] = 0, ms[q] = 0ull, x0[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (rem) {
                        cs[q] = __builtin_amdgcn_readlane(li, __ffsll((long long)rem) - 1);
                        ms[q] = __ballot(lane < b && li == cs[q]);
                        rem &= ~ms[q];
                        const int i0 = __ffsll((long long)ms[q]) - 1;
                        if (active && col_ok) x0[q] = *reinterpret_cast<const float4 *>(xb + (size_t)i0 * d + wave * 256 + (lane << 2));
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (ms[q]) {  // uniform
                        const int c = cs[q], cnt = __popcll(ms[q]);
                        touched |= 1u << c;
                        if (active) {
                            float4 dl = make_float4(x0[q].x * lr32, x0[q].y * lr32, x0[q].z * lr32, x0[q].w * lr32);
                            unsigned long long m = ms[q] & (ms[q] - 1);
                            while (m) {  // further rows of the batch with this label, ascending (torch_scatter's CPU order)
                                const int i = __ffsll((long long)m) - 1;
                                m &= m - 1;
                                const float4 x4 = col_ok ? *reinterpret_cast<const float4 *>(xb + (size_t)i * d + wave * 256 + (lane << 2))
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
                                const float4 v = make_float4(x4.x * lr32, x4.y * lr32, x4.z * lr32, x4.w * lr32);
                                dl = make_float4(dl.x + v.x, dl.y + v.y, dl.z + v.z, dl.w + v.w);
                            }
                            if (col_ok) {
                                const float f = 1.0f - (float)cnt * lr32;
                                float4 *pc4 = reinterpret_cast<float4 *>(sC + c * TP_DS + wave * 256 + ((lane ^ (c & 7)) << 2));
                                const float4 c4 = *pc4;
                                *pc4 = make_float4(c4.x * f + dl.x, c4.y * f + dl.y, c4.z * f + dl.z, c4.w * f + dl.w);
                            }
                        }
                        if (tid == 0) sCnt[c] = sCnt[c] + (float)cnt;
                    }
                }
            }
            __syncthreads();
            pr[6] += TP_CLK() - c3;
            pend |= touched;  // ||c||^2 of these centres is refreshed under the next FMA phase (or at the end)
        }
        pr[3] += TP_CLK() - c3;
        pr[4] += TP_CLK() - c0;
    }
    if (pend) {  // uniform
        tp_refresh_norms(pend, sC, sCn, wave, lane, d);
        __syncthreads();
    }
#undef TP_CLK
    if (PROF && tid == 0) {
        const int w = blockIdx.y * gridDim.x + blockIdx.x;
        if (w == 1 % (int)(gridDim.x * gridDim.y))
            for (int q = 0; q < 8; ++q) ctl->prof[q] = (unsigned long long)pr[q];
        if (w < 256)
            for (int q = 0; q < 8; ++q) ctl->prof_wg[w][q] = (unsigned long long)pr[q];
    }
    // ---- write the owned state back (one replica per centre group)
    if (blockIdx.y == 0 && !sDead && active) {
        for (int c8 = 0; c8 < nck; ++c8) {
            const float4 v = *reinterpret_cast<const float4 *>(sC + c8 * TP_DS + wave * 256 + ((lane ^ (c8 & 7)) << 2));
            if (col_ok) *reinterpret_cast<float4 *>(centers + (size_t)(kbase + c8) * d + wave * 256 + (lane << 2)) = v;
        }
    }
    if (blockIdx.y == 0 && !sDead && tid < nck) {
        cn[kbase + tid] = sCn[tid];
        counts[kbase + tid] = sCnt[tid];
    }
}

// ---------------------------------------------------------------- k_train_persistent_wide
// The same owner-computes epoch for shapes whose 8-centre groups outnumber the CUs (K = 1024: 128 groups x 4 row groups):
// a workgroup owns NCP x 8 centres (NCP "centre passes").  LDS rows have the run-time stride ds = d rounded up to 256
// columns (a 128-d view keeps 64 centres in 64 KB), the work items of the FMA phase are (centre pass, column block)
// pairs dealt round-robin over the 4 waves, wave 0 folds each pass's segments left to right and keeps the best key over
// the passes, and the sweep reads up to 64 centre groups (32 granules per lane).  The update walks the passes; a step
// touches ~b / (K / (8 NCP)) centres of a workgroup, almost always none or one.  Same exchange, same arithmetic, same
// bit-exact result as k_train_persistent -- which stays the kernel of every shape it can hold (K d <= 256 x 1024).
constexpr int TPW_SW = 32;  // granules per lane in the sweep: up to 64 centre groups

template <bool RAGGED>
__device__ __forceinline__ void tpw_dma_block(float *lds, const float *__restrict__ src, int first_row, int nrows_valid, int d,
                                              int ds, int blk, int lane)
{
#pragma unroll
    for (int row = 0; row < 8; ++row) {
        const int grow = first_row + (row < nrows_valid ? row : 0);
        const int col = blk * 256 + ((lane ^ (row & 7)) << 2);
        const float *g = src + (size_t)grow * d + col;
        if (!RAGGED || col < d)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(lds + row * ds + blk * 256), 16, 0, 0);
    }
}

// ||c||^2 of the centres flagged in pend8 among rows [row0, row0 + 8) of sC (half-wave per centre, canonical chains)
__device__ __forceinline__ void tpw_refresh_norms(unsigned pend8, const float *sC, float *sCn, int row0, int ds, int wave, int lane, int d)
{
    const int c8 = 2 * wave + (lane >> 5), q = lane & 31;
    const bool mine = (pend8 >> c8) & 1u;
    float p = 0.f;
    if (mine) {
        const float *base = sC + (row0 + c8) * ds + ((((q >> 2) ^ (c8 & 7))) << 2) + (q & 3);
        const int nu = (d + 31) >> 5;  // columns past d read the zero padding
#pragma unroll 8
        for (int u = 0; u < nu; ++u) {
            const float v = base[u * 32];
            p = __builtin_fmaf(v, v, p);
        }
    }
    p = p + __shfl_xor(p, 1);
    p = p + __shfl_xor(p, 2);
    p = p + __shfl_xor(p, 4);
    p = p + __shfl_xor(p, 8);
    p = p + __shfl_xor(p, 16);
    if (q == 0 && mine) sCn[row0 + c8] = norm2_from_sumsq(p);
}

// ONE_X (round 4, rows wider than 1024 columns): ONE batch-row buffer instead of two -- the rows of step t + 1 are fetched as
// soon as the FMA phase of step t has read the buffer (they land under the exchange) -- so that 8 centres + 8 rows of up to
// ~2400 columns fit a CU's LDS.  A wave owns the column blocks wave, wave + 4, wave + 8, ... (DMA, dot items, update).
//
// NRP = 2 (round 4, late; NCP = 2, ONE_X, ds = 1024): TWO row passes -- the workgroup labels 16 batch rows against its 16 centres,
// the four (centre half, row half) quadrants of a column block as four interleaved FMA chains (dot_quad, as the column-split
// kernel).  K = 1024 at 768 < d <= 1024 then takes 64 x 2 = 128 workgroups of 132 KB instead of 64 x 4 = 256: the centres are
// replicated twice instead of four times and TWO clusterings (the audio and the visual view of cfg5) train side by side.
__device__ __forceinline__ void dot_quad(const float *pc, const float *px, int scz, int sxz, float out[4]);

// MF (round 6): the FMA phase of the 16-centre forms on the f32 matrix core.  v_mfma_f32_16x16x4_f32 is, bit for bit, the ascending
// v_fma_f32 chain (k = 0 .. 3 in order, one rounding per multiply-add: MI355X_MICROARCH.md; k_step_dist_mfma has relied on it since
// round 1), so ONE wave computes the canonical 256-column segment sum of all 16 x 16 (centre, row) pairs of a block as 64 dependent
// MFMAs (40 cycles each: ~2.6k cycles) from 2 x 64 ds_read_b32 -- where the v_fma form read 8 bytes of LDS per multiply-add (every
// lane its own pair: 1 MB per step and workgroup at K = d = 1024) and took 8.6k cycles of a 24k-cycle step, LDS-bound.
// Lane l feeds A[i = l & 15][k = l >> 4] = centre i, column 4 t + k of the block, and B[k][j = l & 15] = batch row j, same column;
// it receives D[4 (l >> 4) + e][l & 15], e = 0 .. 3.  pc / px: the lane's centre / batch row of the block, + k; sc / sx: row & 7
// (chunk tt of 4 columns sits at float offset ((tt ^ s) << 2), as in dot_blocks).
template <bool PIN = true>
__device__ __forceinline__ f32x4 dot_tile_mfma(const float *pc, const float *px, int sc, int sx)
{
    if constexpr (!PIN) {  // the compiler's own schedule: windows of 16 k-steps, lgkmcnt(0) before each
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 64; ++t) {
            const float a = pc[(((t & 7) ^ sc) << 2) + ((t >> 3) << 5)];
            const float bq = px[(((t & 7) ^ sx) << 2) + ((t >> 3) << 5)];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq, acc0, 0, 0, 0);
        }
        return acc0;
    }
    // Two windows of 16 k-steps (32 operand registers each).  Left to itself the compiler loads a window, waits for ALL of it
    // (lgkmcnt(0)) and multiplies -- the last reads are issued right before the wait, ~120 cycles of LDS latency per window on a
    // 40-cycle-per-instruction chain (3.4k cycles per block instead of 2.6k, ACAV_PROFILE_STEPS).  Pinned order: the 16 ds_read2 of
    // window w + 1 go out between the FIRST eight MFMAs of window w, the last eight MFMAs cover their latency.  Same chain, same
    // order of the k-steps: bit-identical.
    const float *pcu[8], *pxu[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        pcu[u] = pc + ((u ^ sc) << 2);
        pxu[u] = px + ((u ^ sx) << 2);
    }
    float A[2][16], B[2][16];
#define ACAV_LDW(slot, w)                                              \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                    \
        A[slot][u] = pcu[u][(2 * (w)) << 5];                           \
        A[slot][u + 8] = pcu[u][(2 * (w) + 1) << 5];                   \
        B[slot][u] = pxu[u][(2 * (w)) << 5];                           \
        B[slot][u + 8] = pxu[u][(2 * (w) + 1) << 5];                   \
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    ACAV_LDW(0, 0)
    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int s = w & 1;
        if (w + 1 < 4) { ACAV_LDW(s ^ 1, w + 1) }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[s][i], B[s][i], acc, 0, 0, 0);
        if (w + 1 < 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA ...
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // ... two ds_read2 of the next window
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        } else
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    }
#undef ACAV_LDW
    return acc;
}

template <bool RAGGED, int NCP, bool ONE_X = false, int NRP = 1>
__global__ __launch_bounds__(256) void k_train_persistent_wide(
    const float *__restrict__ x, const float *__restrict__ xn, int b, int d, int ds, int K, float *__restrict__ centers,
    float *__restrict__ cn, float *__restrict__ counts, const float *__restrict__ thr, double lr0, float r,
    const int64_t *__restrict__ forced, int need, int T, TrainCtl *__restrict__ ctl, StepScalars *__restrict__ sc)
{
    constexpr int NCW = 8 * NCP;  // centres per workgroup
    constexpr int NRW = 8 * NRP;  // batch rows per workgroup
    static_assert(NRP == 1 || (NRP == 2 && NCP == 2 && ONE_X), "two row passes: 16 centres x 16 rows, one row buffer");
#ifdef ACAV_WIDE_NO_MFMA  // experiment builds: the v_fma chains of rounds 2-5 in the 16-centre forms (A/B)
    constexpr bool MF = false;
#else
    constexpr bool MF = NCP == 2;  // 16 centres: one 16 x 16 matrix-core tile per column block (dot_tile_mfma)
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char tpw_smem[];
    float *sC = reinterpret_cast<float *>(tpw_smem);  // [NCW][ds]
    const int nblk = (d + 255) >> 8;
    float *sX0 = sC + NCW * ds;                        // [2 or 1][8][ds]
    float *sCn = sX0 + (ONE_X ? NRP : 2) * 8 * ds;     // [NCW]
    float *sCnt = sCn + NCW;                           // [NCW]
    float *sPart = sCnt + NCW;                         // [NCP * NRP][nblk][64] (NRP = 2: quadrant 2 cp + rp); MF: [nblk][4][64]
    int *sBest = reinterpret_cast<int *>(sPart + (MF ? 4 : NCP * NRP) * nblk * 64);  // [32]
    __shared__ int sDead;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 3, ii = lane & 7;
    const int kbase = blockIdx.x * NCW, rbase = blockIdx.y * NRW;
    const int nck = min(NCW, K - kbase), nrv = min(NRW, b - rbase);
    const int ncg = gridDim.x;
    auto sX = [&](int par) { return ONE_X ? sX0 : sX0 + par * 8 * ds; };
    auto dma_rows = [&](int t) {  // this wave's share of step t's batch rows
        if (ONE_X) {  // issued while wave 0 exchanges: waves 1 .. 3 fetch everything (a barrier follows the wait at the step start)
            if (wave > 0)
                for (int it = wave - 1; it < nblk * NRP; it += 3) {
                    const int rp = NRP == 1 ? 0 : it / nblk, blk = NRP == 1 ? it : it - rp * nblk;
                    if (nrv > rp * 8)  // (a row pass without a valid row is never read into a key)
                        tpw_dma_block<RAGGED>(sX0 + rp * 8 * ds, x + (size_t)t * b * d, rbase + rp * 8, min(8, nrv - rp * 8), d, ds, blk, lane);
                }
        } else
            for (int blk = wave; blk < nblk; blk += 4) tpw_dma_block<RAGGED>(sX(t & 1), x + (size_t)t * b * d, rbase, nrv, d, ds, blk, lane);
    };

    if (RAGGED && wave == ((nblk - 1) & 3)) {  // ragged last block: columns d .. 256 nblk - 1 must read as zero for good
        const int blk = nblk - 1;
        for (int row = 0; row < NCW; ++row)
            *reinterpret_cast<float4 *>(sC + row * ds + blk * 256 + (lane << 2)) = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int row = 0; row < (ONE_X ? NRW : 16); ++row)
            *reinterpret_cast<float4 *>(sX0 + row * ds + blk * 256 + (lane << 2)) = make_float4(0.f, 0.f, 0.f, 0.f);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // before the DMA below writes the same rows
    }
    if (RAGGED && ONE_X) __syncthreads();  // (one row buffer: its blocks are fetched by other waves than the one that zeroed them)
    for (int blk = wave; blk < nblk; blk += 4)
        for (int cp = 0; cp < NCP; ++cp)
            if (cp * 8 < nck) tpw_dma_block<RAGGED>(sC + cp * 8 * ds, centers, kbase + cp * 8, min(8, nck - cp * 8), d, ds, blk, lane);
    if (tid < NCW) {
        const int k = kbase + (tid < nck ? tid : 0);
        sCn[tid] = cn[k];
        sCnt[tid] = counts[k];
    }
    if (tid == 0) sDead = 0;
    if (need < T) dma_rows(need);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    unsigned nsync = 0;
    unsigned long long pend = 0;  // centres whose ||c||^2 is stale: refreshed under the next step's FMA phase
#ifdef ACAV_WIDE_PROF  // experiment builds: shader-clock cycles per phase of workgroup (1, 0), as k_train_persistent's PROF
#define TPW_CLK() ((long long)clock64())
    long long wpr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long wup[4] = {0, 0, 0, 0};  // touched steps: lr / labels, row loads issued -> landed, the rest of the update, closing barrier
    long long wsw[4] = {0, 0, 0, 0};  // wave 0: keys epilogue, all sweep passes, the first pass, sweep end -> past the closing barrier
    long long wsw_end = 0;
#else
#define TPW_CLK() 0ll
#endif
    for (int t = 0; t < T; ++t) {
        const long long wc0 = TPW_CLK();
        long long wc1 = wc0, wc2 = wc0, wc3 = wc0;
        (void)wc1, (void)wc2, (void)wc3;
        const float *xb = x + (size_t)t * b * d;
        if (t < need) {
            __syncthreads();  // every wave has read the previous step's labels (an untouched workgroup has no other barrier)
            if (tid < b) sBest[tid] = (int)forced[(size_t)t * b + tid];
            __syncthreads();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my block of step t's rows landed (issued one step ago)
            if (ONE_X) __syncthreads();  // ... and everybody else's: the blocks a wave multiplies were fetched by waves 1 .. 3
            wc1 = TPW_CLK();
            float xn_t[NRP], thr_t = 0.f;
#pragma unroll
            for (int rp = 0; rp < NRP; ++rp) xn_t[rp] = 0.f;
            if (wave == 0) {  // in flight under the FMA chain
#pragma unroll
                for (int rp = 0; rp < NRP; ++rp) xn_t[rp] = xn[(size_t)t * b + rbase + (rp * 8 + ii < nrv ? rp * 8 + ii : 0)];
                if (MF) xn_t[0] = xn[(size_t)t * b + rbase + ((lane & 15) < nrv ? (lane & 15) : 0)];  // the lane's tile column
                thr_t = thr[t];
            }
            if (!ONE_X && t + 1 < T) dma_rows(t + 1);
            const float *xs = sX(t & 1);
            // (centre pass, column block) pairs: a wave multiplies the blocks it fetched (wave, wave + 4, ...) for every pass
            if constexpr (MF) {  // one 16 x 16 tile per block on the matrix core (NRP = 1: batch rows 8 .. 15 of the tile repeat 0 .. 7, unread)
                const int i15 = lane & 15, kq = lane >> 4, jx = NRP == 2 ? i15 : (i15 & 7);
                for (int blk = wave; blk < nblk; blk += 4) {
                    const f32x4 q4 = dot_tile_mfma<(ACAV_WIDE_PIN != 0)>(sC + i15 * ds + blk * 256 + kq, xs + jx * ds + blk * 256 + kq, i15 & 7, jx & 7);
#pragma unroll
                    for (int e = 0; e < 4; ++e) sPart[(blk * 4 + e) * 64 + lane] = q4[e];
                }
            } else if constexpr (NRP == 2) {  // ds == TS_COLS: the four quadrants of a block as four chains
                for (int blk = wave; blk < nblk; blk += 4) {
                    float q4[4];
                    dot_quad(sC + kk * ds + blk * 256, xs + ii * ds + blk * 256, kk << 2, ii << 2, q4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) sPart[(q * nblk + blk) * 64 + lane] = q4[q];
                }
            }
#ifndef ACAV_WIDE_NO_MFMA
            else if constexpr (NCP == 1) {  // 8 centres x 8 rows (the tall forms, d > 1024): a quarter of a matrix-core tile per block, as k_train_persistent
                const int i15 = lane & 15, kq = lane >> 4, r8 = i15 & 7;
                for (int blk = wave; blk < nblk; blk += 4) {
                    const f32x4 q4 = dot_tile_mfma<(ACAV_WIDE_PIN != 0)>(sC + r8 * ds + blk * 256 + kq, xs + r8 * ds + blk * 256 + kq, r8, r8);
                    if (kq < 2 && i15 < 8) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) sPart[blk * 64 + (4 * kq + e) * 8 + i15] = q4[e];  // slot (kk, ii) = kk * 8 + ii
                    }
                }
            }
#endif
            else
                for (int blk = wave; blk < nblk; blk += 4)
                    for (int cp = 0; cp < NCP; ++cp)
                        sPart[(cp * nblk + blk) * 64 + lane] =
                            dot_blocks<1>(sC + (cp * 8 + kk) * ds + blk * 256, xs + ii * ds + blk * 256, kk << 2, ii << 2, 0.f, true);
            if (pend) {  // uniform: the previous update's norm refresh, off the update's critical path
                for (int cp = 0; cp < NCP; ++cp) {
                    const unsigned p8 = (unsigned)(pend >> (cp * 8)) & 0xFFu;
                    if (p8) tpw_refresh_norms(p8, sC, sCn, cp * 8, ds, wave, lane, d);
                }
                pend = 0;
            }
            __syncthreads();
            wc2 = TPW_CLK();
            if (ONE_X && t + 1 < T) dma_rows(t + 1);  // the one row buffer is free: the next step's rows land under the exchange
            if (wave == 0) {
                unsigned long long keys[NRP];
                unsigned long long o;
                unsigned long long key;
                if constexpr (MF) {  // lane l: centres 4 (l >> 4) + e of batch row l & 15
                    const int i15 = lane & 15, kq = lane >> 4;
                    key = ~0ull;
                    // every operand of the four epilogues first (the centre norms / counts sit in LDS too: one round trip, not five),
                    // the epilogues of all lanes unconditionally (an invalid lane's result is dropped by the select at the end), and
                    // the discount's division only in steps where some lane of the wave needs it (wave-uniform branch)
                    float cn4[4], ct4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) cn4[e] = sCn[4 * kq + e], ct4[e] = sCnt[4 * kq + e];
                    float acc4[4];
                    tp_fold_parts<4>(sPart + lane, 64, 256, nblk, acc4);  // canonical left fold per chain, loads in batches
                    float t4[4];
                    bool disc[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = -2.0f * acc4[e];  // dist_epilogue, the discount apart
                        t = t + xn_t[0];
                        t4[e] = t + cn4[e];
                        disc[e] = ct4[e] < thr_t;
                    }
                    if (__any(disc[0] || disc[1] || disc[2] || disc[3])) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float q = t4[e] / r;
                            t4[e] = disc[e] ? q : t4[e];
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int lc = 4 * kq + e;
                        const unsigned long long kc = (lc < nck && i15 < nrv) ? pack_key(t4[e], kbase + lc) : ~0ull;
                        key = kc < key ? kc : key;
                    }
                    key = tp_min_xor32(tp_min_xor16(key));  // lane l < 16 holds the key of row l
                    (void)o;
                    (void)keys;
                } else {
#pragma unroll
                for (int rp = 0; rp < NRP; ++rp) {
                    unsigned long long key = ~0ull;
                    for (int cp = 0; cp < NCP; ++cp) {
                        const int q = NRP == 1 ? cp : 2 * cp + rp;
                        float acc1[1];
                        tp_fold_parts<1>(sPart + (q * nblk) * 64 + lane, 0, 64, nblk, acc1);  // canonical left fold
                        const float acc = acc1[0];
                        const int lc = cp * 8 + kk;
                        if (lc < nck && rp * 8 + ii < nrv) {
                            const unsigned long long kc = pack_key(dist_epilogue(acc, xn_t[rp], sCn[lc], sCnt[lc] < thr_t, r), kbase + lc);
                            key = kc < key ? kc : key;
                        }
                    }
                    key = tp_min_xor32(tp_min_xor16(tp_min_xor8(key)));
                    (void)o;
                    keys[rp] = key;
                }
                // lane l < 8 NRP holds the key of row l: its ii is l & 7 and every lane of an ii column holds that row's minimum
                key = (NRP == 2 && (lane >> 3) == 1) ? keys[NRP - 1] : keys[0];
                }
#ifdef ACAV_WIDE_PROF
                if (t >= need) wsw[0] += TPW_CLK() - wc2;
                const long long wsw0 = TPW_CLK();
#endif
                const unsigned long long tag = (unsigned long long)((nsync % 65535u) + 1u) << 48;
                unsigned long long *ring = ctl->gran[nsync % TP_RING];
                if (lane < nrv) {
                    const unsigned long long local = (key == ~0ull) ? 0xFFFFull : ((key & 0xffffffffull) - (unsigned)kbase);
                    __hip_atomic_store(&ring[tp_gran_index(blockIdx.x, rbase + lane)], tag | (local << 32) | (key >> 32), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
#if ACAV_SWEEP_LEAN
                {  // (the per-lane form of rounds 1-5 below: -DACAV_SWEEP_LEAN=0)
#if ACAV_TPW_FIRST_SLEEP > 0
                    __builtin_amdgcn_s_sleep(ACAV_TPW_FIRST_SLEEP);
#endif
                    const unsigned tag16 = (nsync % 65535u) + 1u;
                    const int nu = (ncg + 1) >> 1;
                    int bl = -1;
                    unsigned np = 0, okl;
                    if (nu <= 16) okl = tp_sweep_lean<16>(ring, ncg, b, NCW, tag16, lane, ctl, &bl, &np);
                    else okl = tp_sweep_lean<TPW_SW>(ring, ncg, b, NCW, tag16, lane, ctl, &bl, &np);
#ifdef ACAV_WIDE_PROF
                    wpr[7] += np;
                    wsw_end = TPW_CLK();
                    if (t >= need) wsw[1] += wsw_end - wsw0;
#endif
                    if (lane < b) sBest[lane] = bl;
                    if (!okl && lane == 0) sDead = 1;
                }
#else
                {
                const int srow = lane & 31, half = lane >> 5;
                unsigned long long bestkey = ~0ull;
                unsigned ok = 1;
                unsigned long long g[TPW_SW];
                unsigned needm = 0;  // bit u: granule u of this lane not yet seen with this step's tag
                const bool reread_all = gridDim.x * gridDim.y <= 128u;
#pragma unroll
                for (int u = 0; u < TPW_SW; ++u)
                    if (srow < b && half + 2 * u < ncg) needm |= 1u << u;
#if ACAV_TPW_FIRST_SLEEP > 0
                __builtin_amdgcn_s_sleep(ACAV_TPW_FIRST_SLEEP);  // experiment knob: x 64 clocks between the publish and the first sweep
#endif
                for (unsigned spins = 0;; ++spins) {
#if ACAV_TPW_PASS_SLEEP > 0
                    if (spins) __builtin_amdgcn_s_sleep(ACAV_TPW_PASS_SLEEP);  // experiment knob: pause before a repeated pass
#endif
#ifdef ACAV_WIDE_PROF
                    wpr[7] += 1;  // sweep passes
                    const long long wpass0 = TPW_CLK();
#endif
                    // up to 128 workgroups (the 16 x 16 form at K = 1024): every pass re-reads everything under a wave-uniform
                    // condition (k_train_persistent); more pollers than that (16 x 8 forms: 256) and the volume of full re-reads costs
                    // more than the per-lane conditions -- d = 128 / K = 1024 on 256 workgroups 9.05 vs 8.38 us per step
                    if (ACAV_SWEEP_REREAD_ALL && reread_all) {
#pragma unroll
                        for (int u = 0; u < TPW_SW; ++u)
                            if (2 * u < ncg)
                                g[u] = __hip_atomic_load(&ring[tp_gran_index(half + 2 * u, srow)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
#pragma unroll
                        for (int u = 0; u < TPW_SW; ++u)
                            if ((needm >> u) & 1u)
                                g[u] = __hip_atomic_load(&ring[tp_gran_index(half + 2 * u, srow)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int u = 0; u < TPW_SW; ++u)
                        if (((needm >> u) & 1u) && (g[u] >> 48) == (tag >> 48)) needm &= ~(1u << u);
#ifdef ACAV_WIDE_PROF
                    if (t >= need && spins == 0) wsw[2] += TPW_CLK() - wpass0;
#endif
                    if (__all(needm == 0)) break;
                    if (spins > TP_SPIN_LIMIT || (spins & 1023) == 1023) {
                        if (spins > TP_SPIN_LIMIT || __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            if (lane == 0) __hip_atomic_store(&ctl->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = 0;
                            break;
                        }
                    }
                }
#ifdef ACAV_WIDE_PROF
                wsw_end = TPW_CLK();
                if (t >= need) wsw[1] += wsw_end - wsw0;
#endif
#pragma unroll
                for (int u = 0; u < TPW_SW; ++u) {
                    const int cg = half + 2 * u;
                    if (srow < b && cg < ncg && ok) {
                        const unsigned loc = (unsigned)(g[u] >> 32) & 0xFFFFu;
                        const unsigned long long cand =
                            loc == 0xFFFFu ? ~0ull : (((g[u] & 0xffffffffull) << 32) | (unsigned)(cg * NCW + loc));
                        bestkey = cand < bestkey ? cand : bestkey;
                    }
                }
                o = __shfl_xor(bestkey, 32);
                bestkey = o < bestkey ? o : bestkey;
                if (lane < b) sBest[lane] = (int)(bestkey & 0xffffffffull);
                if (!ok && lane == 0) sDead = 1;
                }
#endif
            }
            ++nsync;
            __syncthreads();
            wc3 = TPW_CLK();
#ifdef ACAV_WIDE_PROF
            if (t >= need && wave == 0) wsw[3] += wc3 - wsw_end;
#endif
            if (sDead) break;  // uniform
        }
        // ---- update: every replica of a centre group does the same arithmetic; wave w owns column block w
        const int best = (lane < b) ? sBest[lane] : -1;
        double lr = lr0;
        bool fell = false;
        if ((double)b * lr0 >= 1.0) {  // lr fallback (:116-119) possible at all?  (never at the defaults 32 * 0.01)
            int cmaxi = 0;
#pragma unroll
            for (int i = 0; i < TP_MAXB; ++i) {
                const int li = __builtin_amdgcn_readlane(best, i);  // scalar; -1 for rows >= b
                const int c = __popcll(__ballot(lane < b && best == li));
                cmaxi = (i < b && c > cmaxi) ? c : cmaxi;
            }
            if ((double)(float)cmaxi * lr >= 1.0) {
                lr = 0.5 / (double)(float)cmaxi;
                fell = true;
            }
        }
        const float lr32 = (float)lr;
        if (fell && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) sc->fallback += 1;
#ifdef ACAV_WIDE_PROF
        if (t >= need) wpr[0] += wc1 - wc0, wpr[1] += wc2 - wc1, wpr[2] += wc3 - wc2, wpr[5] += 1;
#endif
        if (__ballot(lane < b && best >= kbase && best < kbase + nck) == 0ull) {  // uniform: nothing of mine was hit
#ifdef ACAV_WIDE_PROF
            if (t >= need) wpr[3] += TPW_CLK() - wc3, wpr[4] += TPW_CLK() - wc0;
#endif
            continue;
        }
#ifdef ACAV_WIDE_PROF
        const long long wu0 = TPW_CLK();
        const long long wacc1 = 0;  // (the row round trip is no longer timed apart: the compact loop overlaps it)
#endif
        // A COMPACT loop over the centres that were hit, four at a time (round 4).  The form it replaced walked every centre pass with
        // 8 centres x 3 column blocks unrolled: ~1000 straight-line instructions that ran once per step in a third of the workgroups,
        // i.e. out of a cold instruction cache -- 4.8k cycles per touched step for ~300 cycles of work (ACAV_WIDE_PROF), on the chain
        // of the next step's slowest publisher.  Same arithmetic: rows ascending within a centre, v = x * lr, dl = v0 (+ v1 ...),
        // c = c * f + dl; the order of the CENTRES does not matter (disjoint rows of sC, sCnt, bits of pend).
        {
            const int li = best - kbase;  // this lane's batch row -> local centre (valid: lane < b and 0 <= li < nck)
            unsigned long long rem = __ballot(lane < b && li >= 0 && li < nck);
            // a wave's column blocks: wave, wave + 4, wave + 8 (d <= 3072); the loads of a row for all of them are in flight together
            bool okb[3];
            int colb[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                colb[u] = (wave + 4 * u) * 256 + (lane << 2);
                okb[u] = wave + 4 * u < nblk && (!RAGGED || colb[u] < d);
            }
            const bool own = wave < nblk;
            while (rem) {  // uniform: up to four centres per trip, the first row of each in flight together (one round trip)
                int cs[4];
                unsigned long long ms[4];
                float4 x0[4][3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    cs[q] = 0, ms[q] = 0ull;
#pragma unroll
                    for (int u = 0; u < 3; ++u) x0[q][u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (rem) {
                        cs[q] = __builtin_amdgcn_readlane(li, __ffsll((long long)rem) - 1);
                        ms[q] = __ballot(lane < b && li == cs[q]);
                        rem &= ~ms[q];
                        const int i0 = __ffsll((long long)ms[q]) - 1;
#pragma unroll
                        for (int u = 0; u < 3; ++u)
                            if (okb[u]) x0[q][u] = *reinterpret_cast<const float4 *>(xb + (size_t)i0 * d + colb[u]);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (ms[q]) {  // uniform
                        const int c = cs[q], cnt = __popcll(ms[q]);
                        if (own) {
                            float4 dl[3];
#pragma unroll
                            for (int u = 0; u < 3; ++u) dl[u] = make_float4(x0[q][u].x * lr32, x0[q][u].y * lr32, x0[q][u].z * lr32, x0[q][u].w * lr32);
                            unsigned long long m = ms[q] & (ms[q] - 1);
                            while (m) {  // further rows of the batch with this label, ascending (torch_scatter's CPU order)
                                const int i = __ffsll((long long)m) - 1;
                                m &= m - 1;
                                float4 x4[3];
#pragma unroll
                                for (int u = 0; u < 3; ++u)
                                    x4[u] = okb[u] ? *reinterpret_cast<const float4 *>(xb + (size_t)i * d + colb[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                                for (int u = 0; u < 3; ++u) {
                                    const float4 v = make_float4(x4[u].x * lr32, x4[u].y * lr32, x4[u].z * lr32, x4[u].w * lr32);
                                    dl[u] = make_float4(dl[u].x + v.x, dl[u].y + v.y, dl[u].z + v.z, dl[u].w + v.w);
                                }
                            }
                            const float f = 1.0f - (float)cnt * lr32;
#pragma unroll
                            for (int u = 0; u < 3; ++u)
                                if (okb[u]) {
                                    float4 *pc4 = reinterpret_cast<float4 *>(sC + c * ds + (wave + 4 * u) * 256 + ((lane ^ (c & 7)) << 2));
                                    const float4 c4 = *pc4;
                                    *pc4 = make_float4(c4.x * f + dl[u].x, c4.y * f + dl[u].y, c4.z * f + dl[u].z, c4.w * f + dl[u].w);
                                }
                        }
                        if (tid == 0) sCnt[c] = sCnt[c] + (float)cnt;
                        pend |= 1ull << c;
                    }
                }
            }
        }
#ifdef ACAV_WIDE_PROF
        const long long wu3 = TPW_CLK();
#endif
        __syncthreads();  // the updated centres and counts are in place before the next step reads them
#ifdef ACAV_WIDE_PROF
        if (t >= need) {
            wpr[3] += TPW_CLK() - wc3, wpr[4] += TPW_CLK() - wc0, wpr[6] += 1;
            wup[0] += wu0 - wc3, wup[1] += wacc1, wup[2] += wu3 - wu0 - wacc1, wup[3] += TPW_CLK() - wu3;
        }
#endif
    }
#ifdef ACAV_WIDE_PROF
    if (blockIdx.x == 1 && blockIdx.y == 0 && tid == 0)
    {
        for (int q = 0; q < 8; ++q) ctl->prof[q] = (unsigned long long)wpr[q];
        for (int q = 0; q < 4; ++q) ctl->prof_wg[0][q] = (unsigned long long)wup[q];
        for (int q = 0; q < 4; ++q) ctl->prof_wg[0][4 + q] = (unsigned long long)wsw[q];
    }
#endif
#undef TPW_CLK
    if (pend) {  // uniform
        for (int cp = 0; cp < NCP; ++cp) {
            const unsigned p8 = (unsigned)(pend >> (cp * 8)) & 0xFFu;
            if (p8) tpw_refresh_norms(p8, sC, sCn, cp * 8, ds, wave, lane, d);
        }
        __syncthreads();
    }
    // ---- write the owned state back (one replica per centre group)
    if (blockIdx.y == 0 && !sDead)
        for (int blk = wave; blk < nblk; blk += 4) {
            const bool col_ok = !RAGGED || blk * 256 + (lane << 2) < d;
            for (int c = 0; c < nck; ++c) {
                const float4 v = *reinterpret_cast<const float4 *>(sC + c * ds + blk * 256 + ((lane ^ (c & 7)) << 2));
                if (col_ok) *reinterpret_cast<float4 *>(centers + (size_t)(kbase + c) * d + blk * 256 + (lane << 2)) = v;
            }
        }
    if (blockIdx.y == 0 && !sDead && tid < nck) {
        cn[kbase + tid] = sCn[tid];
        counts[kbase + tid] = sCnt[tid];
    }
}

// ---------------------------------------------------------------- k_train_persistent_split
// The owner-computes epoch for 1024 < d <= 2048 (cfg4's 2048-d visual view; round 3).  K = 1024 centres x 2048 columns are
// 8 MB: no workgroup count <= 256 holds them in LDS next to whole batch rows, so the COLUMNS are split: workgroup
// (cg, rg, ch) owns centres [16 cg, 16 cg + 16) x columns [1024 ch, 1024 ch + 1024) (64 KB) and labels rows
// [16 rg, 16 rg + 16) of every step (one 64 KB row buffer: the rows of step t + 1 are fetched by LDS-DMA as soon as the
// FMA phase of step t has read the buffer, under the exchange).  K = 1024, b = 32: 64 x 2 x 2 = 256 workgroups, one per CU.
//   * canonical dot: the fold over 256-column segments is left to right, ((((s0 + s1) + s2) + s3) + s4) + ... -- the ch = 0
//     workgroup folds its four segments into t0 and hands it to its ch = 1 partner (same cg, rg) as one tagged 8-byte
//     granule per (centre, row) pair (256 per pair of workgroups and step); the partner adds ITS segments in order and owns
//     the distances, the argmin over its 16 centres and the label granule of the existing exchange.  Same bits as one chain.
//   * canonical ||c||^2 (32 interleaved chains over ALL columns, then the fixed tree): after an update the ch = 0 workgroup
//     runs the chains over its half and hands the 32 partial sums of each touched centre to the partner, which continues
//     them over its half.  Only ch = 1 needs the norms.
//   * both halves sweep the label granules and apply the identical update to their columns.
// Hand-off slots are single (no ring): ch = 0 cannot reach its next publish before it has seen the label granules of the
// step, which its partner only writes after consuming the slot.  Tags are the step number + 1 (the rings are zeroed
// before every launch).  Bounded spins + error flag + re-run on the per-step path, as k_train_persistent.
constexpr int TS_NC = 16;      // centres per workgroup
constexpr int TS_NR = 16;      // batch rows per workgroup
constexpr int TS_COLS = 1024;  // columns per workgroup (= TP_DS: the LDS row stride of tp_dma_block / dot_blocks)
constexpr size_t TS_SMEM = sizeof(float) * (size_t)(2 * 16 * TS_COLS + 2 * TS_NC + 4 * 4 * 64 + 32) + 8 * 64 + 64;

__device__ __forceinline__ unsigned long long ts_granule(float v, unsigned tag)
{
    return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}
// spin until the slot carries `tag`; false = gave up (another workgroup is not resident / error flag raised)
__device__ __forceinline__ bool ts_poll(const unsigned long long *slot, unsigned tag, TrainCtl *ctl, float *out)
{
    for (unsigned spins = 0;; ++spins) {
        const unsigned long long g = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(g >> 32) == tag) {
            *out = __uint_as_float((unsigned)g);
            return true;
        }
        if (spins > TP_SPIN_LIMIT || (spins & 1023) == 1023) {
            if (spins > TP_SPIN_LIMIT || __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&ctl->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}

// Four canonical segment sums at once: the (centre, row) pairs (kk, ii), (kk, ii + 8), (kk + 8, ii), (kk + 8, ii + 8) of one
// 256-column block -- four independent v_fma chains per lane (a lone chain is latency-bound: 4 cycles per column) fed by two
// centre and two row chunks per step instead of one of each per chain.  Ascending columns within each chain, as dot_blocks.
__device__ __forceinline__ void dot_quad(const float *pc, const float *px, int scz, int sxz, float out[4])
{
    // chunk tt (4 columns) sits at float offset ((tt ^ s) << 2): 8 per-lane base pointers + compile-time offsets, and the
    // loads of chunk tt + 1 in flight under the FMAs of chunk tt (as dot_blocks)
    const float *pcu[8], *pxu[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        pcu[u] = pc + ((u << 2) ^ scz);
        pxu[u] = px + ((u << 2) ^ sxz);
    }
    float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
    float4 cA[2], xA[2], cB[2], xB[2];
#define TS_LD(Cq, Xq, tt)                                                                       \
    Cq[0] = *reinterpret_cast<const float4 *>(pcu[(tt) & 7] + (((tt) >> 3) << 5));                \
    Cq[1] = *reinterpret_cast<const float4 *>(pcu[(tt) & 7] + 8 * TS_COLS + (((tt) >> 3) << 5));  \
    Xq[0] = *reinterpret_cast<const float4 *>(pxu[(tt) & 7] + (((tt) >> 3) << 5));                \
    Xq[1] = *reinterpret_cast<const float4 *>(pxu[(tt) & 7] + 8 * TS_COLS + (((tt) >> 3) << 5));
#define TS_FMA4(e)                                                                                   \
    a00 = __builtin_fmaf(Cq[0].e, Xq[0].e, a00), a01 = __builtin_fmaf(Cq[0].e, Xq[1].e, a01),       \
    a10 = __builtin_fmaf(Cq[1].e, Xq[0].e, a10), a11 = __builtin_fmaf(Cq[1].e, Xq[1].e, a11);
#define TS_FMA(Cq_, Xq_)            \
    {                               \
        const float4 *Cq = Cq_, *Xq = Xq_; \
        TS_FMA4(x) TS_FMA4(y) TS_FMA4(z) TS_FMA4(w) \
    }
    TS_LD(cA, xA, 0)
#pragma unroll
    for (int tt = 0; tt < 64; tt += 2) {
        TS_LD(cB, xB, tt + 1)
        TS_FMA(cA, xA)
        if (tt + 2 < 64) { TS_LD(cA, xA, tt + 2) }
        TS_FMA(cB, xB)
    }
#undef TS_LD
#undef TS_FMA4
#undef TS_FMA
    out[0] = a00, out[1] = a01, out[2] = a10, out[3] = a11;  // quadrant q = 2 (centre half) + (row half)
}

template <bool PROF>
__global__ __launch_bounds__(256) void k_train_persistent_split(
    const float *__restrict__ x, const float *__restrict__ xn, int b, int d, int K, float *__restrict__ centers,
    float *__restrict__ cn, float *__restrict__ counts, const float *__restrict__ thr, double lr0, float r,
    const int64_t *__restrict__ forced, int need, int T, TrainCtl *__restrict__ ctl, StepScalars *__restrict__ sc,
    unsigned long long *__restrict__ t0ring, unsigned long long *__restrict__ nring)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ts_smem[];
    float *sC = reinterpret_cast<float *>(ts_smem);   // [16][1024]
    float *sX = sC + TS_NC * TS_COLS;                  // [16][1024]
    float *sCn = sX + TS_NR * TS_COLS;                 // [16]
    float *sCnt = sCn + TS_NC;                         // [16]
    float *sPart = sCnt + TS_NC;                       // [4 quadrants][4 column blocks][64]
    unsigned long long *sKey = reinterpret_cast<unsigned long long *>(sPart + 4 * 4 * 64);  // [4][16] (v_fma form: [4][8])
    int *sBest = reinterpret_cast<int *>(sKey + 64);   // [32]
    __shared__ int sDead;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kk = lane >> 3, ii = lane & 7;
    (void)kk, (void)ii;
    const int cg = blockIdx.x, rg = blockIdx.y, ch = blockIdx.z;
    const int ncg = gridDim.x;
    const int pair = cg * gridDim.y + rg;
    const int kbase = cg * TS_NC, rbase = rg * TS_NR, coff = ch * TS_COLS;
    const int nck = min(TS_NC, K - kbase), nrv = min(TS_NR, b - rbase);
    const int nblk = ch == 0 ? 4 : (d - TS_COLS) >> 8;  // 256-column blocks of this half (d % 256 == 0)
    const bool active = wave < nblk;                    // this wave has a column block (DMA, FMA chains, update)
    unsigned long long *my_t0 = t0ring + (size_t)pair * 256 + tid;  // slot of (quadrant = wave, lane)
    unsigned long long *my_nr = nring + (size_t)pair * (TS_NC * 32);

    if (active) {
        tp_dma_block<false>(sC, centers + coff, kbase, min(8, nck), d, wave, lane);
        tp_dma_block<false>(sC + 8 * TS_COLS, centers + coff, kbase + (nck > 8 ? 8 : 0), max(min(8, nck - 8), 1), d, wave, lane);
    }
    if (tid < TS_NC) {
        const int k = kbase + (tid < nck ? tid : 0);
        sCn[tid] = cn[k];
        sCnt[tid] = counts[k];
    }
    if (tid == 0) sDead = 0;
    auto dma_rows = [&](int t) {
        const float *src = x + (size_t)t * b * d + coff;
        tp_dma_block<false>(sX, src, rbase, min(8, nrv), d, wave, lane);
        tp_dma_block<false>(sX + 8 * TS_COLS, src, rbase + (nrv > 8 ? 8 : 0), max(min(8, nrv - 8), 1), d, wave, lane);
    };
    if (need < T && active) dma_rows(need);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ||c||^2 of the centres flagged in pend16: chains over this half (continued from the partner's partial sums when
    // ch = 1), half-wave per centre, 8 centres per pass; tag: the step the refresh runs in, + 1
    auto refresh_norms = [&](unsigned pend16, unsigned tag) -> bool {
        bool ok = true;
        for (int cp = 0; cp < 2; ++cp) {
            const unsigned p8 = (pend16 >> (cp * 8)) & 0xFFu;
            if (!p8) continue;
            const int c8 = 2 * wave + (lane >> 5), q = lane & 31, c = cp * 8 + c8;
            const bool mine = (p8 >> c8) & 1u;
            float p = 0.f;
            if (mine && ch == 1) ok = ts_poll(my_nr + c * 32 + q, tag, ctl, &p) && ok;
            if (mine) {
                const float *base = sC + c * TS_COLS + ((((q >> 2) ^ (c8 & 7))) << 2) + (q & 3);
                const int nu = nblk * 8;  // 32-column groups of this half
#pragma unroll 8
                for (int u = 0; u < nu; ++u) {
                    const float v = base[u * 32];
                    p = __builtin_fmaf(v, v, p);
                }
            }
            if (ch == 0) {
                if (mine) __hip_atomic_store(my_nr + c * 32 + q, ts_granule(p, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                p = p + __shfl_xor(p, 1);
                p = p + __shfl_xor(p, 2);
                p = p + __shfl_xor(p, 4);
                p = p + __shfl_xor(p, 8);
                p = p + __shfl_xor(p, 16);
                if (q == 0 && mine) sCn[c] = norm2_from_sumsq(p);
            }
        }
        return ok;
    };

    unsigned nsync = 0;
    unsigned pend = 0;  // centres whose ||c||^2 is stale: refreshed under the next step's FMA phase
#define TS_CLK() (PROF ? (long long)clock64() : 0ll)
    long long pr[6] = {0, 0, 0, 0, 0, 0};  // ACAV_PROFILE_STEPS: cycles in row wait / FMA+norms / hand-off+keys / sweep / update / step
    for (int t = 0; t < T; ++t) {
        const float *xb = x + (size_t)t * b * d;
        const long long c0 = TS_CLK();
        long long c4 = c0;
        if (t < need) {
            __syncthreads();  // every wave has read the previous step's labels
            if (tid < b) sBest[tid] = (int)forced[(size_t)t * b + tid];
            __syncthreads();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my blocks of step t's rows landed (a wave reads only its own block)
            const unsigned tag = (unsigned)t + 1u;
            const long long c1 = TS_CLK();
#ifndef ACAV_WIDE_NO_MFMA
            // round 6: the 16 x 16 (centre, row) tile of a column block on the f32 matrix core (dot_tile_mfma: bit for bit the v_fma
            // chain).  Lane l of the tile holds centres 4 (l >> 4) + e, e = 0 .. 3, of batch row l & 15; "quadrant" q below = e.
            const int i15 = lane & 15, kq = lane >> 4;
            float xn_t = 0.f, thr_t = 0.f;
            if (ch == 1) {  // in flight under the chains
                xn_t = xn[(size_t)t * b + rbase + (i15 < nrv ? i15 : 0)];
                thr_t = thr[t];
            }
            if (active) {
                const f32x4 seg = dot_tile_mfma<(ACAV_WIDE_PIN != 0)>(sC + i15 * TS_COLS + wave * 256 + kq, sX + i15 * TS_COLS + wave * 256 + kq, i15 & 7, i15 & 7);
#pragma unroll
                for (int q = 0; q < 4; ++q) sPart[(q * 4 + wave) * 64 + lane] = seg[q];
            }
#else
            // quadrant q of the 16 x 16 (centre, row) pairs: centres 8 (q >> 1) + kk, rows 8 (q & 1) + ii
            float xn_t = 0.f, thr_t = 0.f;
            if (ch == 1) {  // in flight under the FMA chains: wave = quadrant in the fold below
                const int i = (wave & 1) * 8 + ii;
                xn_t = xn[(size_t)t * b + rbase + (i < nrv ? i : 0)];
                thr_t = thr[t];
            }
            if (active) {
                float seg[4];
                dot_quad(sC + kk * TS_COLS + wave * 256, sX + ii * TS_COLS + wave * 256, kk << 2, ii << 2, seg);
#pragma unroll
                for (int q = 0; q < 4; ++q) sPart[(q * 4 + wave) * 64 + lane] = seg[q];
            }
#endif
            bool ok = true;
            if (pend) {  // uniform: the previous update's norm refresh, off the update's critical path
                ok = refresh_norms(pend, tag);
                pend = 0;
            }
            __syncthreads();  // partial sums in place, sX free
            const long long c2 = TS_CLK();
            {
                const int q = wave;  // this wave folds quadrant q
#ifndef ACAV_WIDE_NO_MFMA
                const int c = 4 * kq + q, i = i15;
#else
                const int c = (q >> 1) * 8 + kk, i = (q & 1) * 8 + ii;
#endif
                float acc = 0.f;
                if (ch == 0) {
                    acc = sPart[(q * 4) * 64 + lane];
#pragma unroll
                    for (int w = 1; w < 4; ++w) acc = acc + sPart[(q * 4 + w) * 64 + lane];  // canonical left fold
                    __hip_atomic_store(my_t0, ts_granule(acc, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
#ifndef ACAV_WIDE_NO_MFMA
                    const float cn_c = sCn[c], ct_c = sCnt[c];  // (c <= 15 always; in flight with the segment sums)
#else
                    const float cn_c = sCn[c < TS_NC ? c : 0], ct_c = sCnt[c < TS_NC ? c : 0];
#endif
                    float v4[4];  // this half's segment sums: in flight while the partner's partial sum is polled
#pragma unroll
                    for (int w = 0; w < 4; ++w) v4[w] = sPart[(q * 4 + (w < nblk ? w : nblk - 1)) * 64 + lane];
                    ok = ts_poll(my_t0, tag, ctl, &acc) && ok;
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        if (w < nblk) acc = acc + v4[w];  // ... continued over this half, left to right
                    unsigned long long key = ~0ull;
                    if (c < nck && i < nrv) key = pack_key(dist_epilogue(acc, xn_t, cn_c, ct_c < thr_t, r), kbase + c);
#ifndef ACAV_WIDE_NO_MFMA
                    key = tp_min_xor32(tp_min_xor16(key));
                    if (lane < 16) sKey[q * 16 + lane] = key;  // best of centres q, 4 + q, 8 + q, 12 + q for row `lane`
#else
                    unsigned long long o = __shfl_xor(key, 8);
                    key = o < key ? o : key;
                    o = __shfl_xor(key, 16);
                    key = o < key ? o : key;
                    o = __shfl_xor(key, 32);
                    key = o < key ? o : key;
                    if (lane < 8) sKey[q * 8 + lane] = key;  // best of this quadrant's 8 centres for row 8 (q & 1) + lane
#endif
                }
            }
            if (!__all(ok) ) {
                if (lane == 0) sDead = 1;
            }
            // the rows of the next step: 16 DMA instructions per wave, issued AFTER the hand-off left (they land under the sweep)
            if (t + 1 < T && active) dma_rows(t + 1);
            __syncthreads();
            const long long c3 = TS_CLK();
            if (wave == 0) {
                const unsigned long long tag16 = (unsigned long long)((nsync % 65535u) + 1u) << 48;
                unsigned long long *ring = ctl->gran[nsync % TP_RING];
                if (ch == 1 && lane < nrv) {
#ifndef ACAV_WIDE_NO_MFMA
                    unsigned long long key = sKey[lane];
#pragma unroll
                    for (int e = 1; e < 4; ++e) {
                        const unsigned long long ke = sKey[e * 16 + lane];
                        key = ke < key ? ke : key;
                    }
#else
                    const int hq = lane >> 3, i8 = lane & 7;  // row half, row in it: quadrants hq and hq + 2 cover its 16 centres
                    const unsigned long long k0 = sKey[hq * 8 + i8], k1 = sKey[(hq + 2) * 8 + i8];
                    const unsigned long long key = k1 < k0 ? k1 : k0;
#endif
                    const unsigned long long local = (key == ~0ull) ? 0xFFFFull : ((key & 0xffffffffull) - (unsigned)kbase);
                    __hip_atomic_store(&ring[tp_gran_index(cg, rbase + lane)], tag16 | (local << 32) | (key >> 32), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                }
#if ACAV_SWEEP_LEAN
                {
                    const int nu = (ncg + 1) >> 1;
                    int bl = -1;
                    unsigned np = 0, okl;
                    if (nu <= 16) okl = tp_sweep_lean<16>(ring, ncg, b, TS_NC, (nsync % 65535u) + 1u, lane, ctl, &bl, &np);
                    else okl = tp_sweep_lean<TPW_SW>(ring, ncg, b, TS_NC, (nsync % 65535u) + 1u, lane, ctl, &bl, &np);
                    if (lane < b) sBest[lane] = bl;
                    if (!okl && lane == 0) sDead = 1;
                }
#else
                const int srow = lane & 31, half = lane >> 5;
                unsigned long long bestkey = ~0ull;
                unsigned okw = 1;
                unsigned long long g[TPW_SW];
                unsigned needm = 0;  // bit u: granule u of this lane not yet seen with this step's tag
#pragma unroll
                for (int u = 0; u < TPW_SW; ++u)
                    if (srow < b && half + 2 * u < ncg) needm |= 1u << u;
                for (unsigned spins = 0;; ++spins) {
#pragma unroll
                    for (int u = 0; u < TPW_SW; ++u)
                        if ((needm >> u) & 1u)  // (256 workgroups: only the missing granules -- full re-reads measured 10.68 vs 10.50 us per step)
                            g[u] = __hip_atomic_load(&ring[tp_gran_index(half + 2 * u, srow)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int u = 0; u < TPW_SW; ++u)
                        if (((needm >> u) & 1u) && (g[u] >> 48) == (tag16 >> 48)) needm &= ~(1u << u);
                    if (__all(needm == 0)) break;
                    if (spins > TP_SPIN_LIMIT || (spins & 1023) == 1023) {
                        if (spins > TP_SPIN_LIMIT || __hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            if (lane == 0) __hip_atomic_store(&ctl->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            okw = 0;
                            break;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < TPW_SW; ++u) {
                    const int cgu = half + 2 * u;
                    if (srow < b && cgu < ncg && okw) {
                        const unsigned loc = (unsigned)(g[u] >> 32) & 0xFFFFu;
                        const unsigned long long cand =
                            loc == 0xFFFFu ? ~0ull : (((g[u] & 0xffffffffull) << 32) | (unsigned)(cgu * TS_NC + loc));
                        bestkey = cand < bestkey ? cand : bestkey;
                    }
                }
                const unsigned long long o = __shfl_xor(bestkey, 32);
                bestkey = o < bestkey ? o : bestkey;
                if (lane < b) sBest[lane] = (int)(bestkey & 0xffffffffull);
                if (!okw && lane == 0) sDead = 1;
#endif
            }
            ++nsync;
            __syncthreads();
            c4 = TS_CLK();
            if (PROF) pr[0] += c1 - c0, pr[1] += c2 - c1, pr[2] += c3 - c2, pr[3] += c4 - c3;
            if (sDead) break;  // uniform
        }
        // ---- update: every replica of a centre group does the same arithmetic on its columns; wave w owns block w
        const int best = (lane < b) ? sBest[lane] : -1;
        double lr = lr0;
        bool fell = false;
        if ((double)b * lr0 >= 1.0) {  // lr fallback (:116-119) possible at all?  (never at the defaults 32 * 0.01)
            int cmaxi = 0;
#pragma unroll
            for (int i = 0; i < TP_MAXB; ++i) {
                const int li = __builtin_amdgcn_readlane(best, i);  // scalar; -1 for rows >= b
                const int c = __popcll(__ballot(lane < b && best == li));
                cmaxi = (i < b && c > cmaxi) ? c : cmaxi;
            }
            if ((double)(float)cmaxi * lr >= 1.0) {
                lr = 0.5 / (double)(float)cmaxi;
                fell = true;
            }
        }
        const float lr32 = (float)lr;
        if (fell && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) sc->fallback += 1;
        if (__ballot(lane < b && best >= kbase && best < kbase + nck) == 0ull) {  // uniform: nothing of mine was hit
            if (PROF) pr[5] += TS_CLK() - c0;
            continue;
        }
        {  // COMPACT update (as k_train_persistent / k_train_persistent_wide): the hit centres four at a time, first rows in flight together
            const int li = best - kbase;  // this lane's batch row -> local centre (valid: lane < b, 0 <= li < nck)
            unsigned long long rem = __ballot(lane < b && li >= 0 && li < nck);
            while (rem) {  // uniform
                int cs[4];
                unsigned long long ms[4];
                float4 x0[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    cs[q] = 0, ms[q] = 0ull, x0[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (rem) {
                        cs[q] = __builtin_amdgcn_readlane(li, __ffsll((long long)rem) - 1);
                        ms[q] = __ballot(lane < b && li == cs[q]);
                        rem &= ~ms[q];
                        const int i0 = __ffsll((long long)ms[q]) - 1;
                        if (active) x0[q] = *reinterpret_cast<const float4 *>(xb + (size_t)i0 * d + coff + wave * 256 + (lane << 2));
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (ms[q]) {  // uniform
                        const int c = cs[q], cnt = __popcll(ms[q]);
                        if (active) {
                            float4 dl = make_float4(x0[q].x * lr32, x0[q].y * lr32, x0[q].z * lr32, x0[q].w * lr32);
                            unsigned long long m = ms[q] & (ms[q] - 1);
                            while (m) {  // further rows of the batch with this label, ascending (torch_scatter's CPU order)
                                const int i = __ffsll((long long)m) - 1;
                                m &= m - 1;
                                const float4 x4 = *reinterpret_cast<const float4 *>(xb + (size_t)i * d + coff + wave * 256 + (lane << 2));
                                const float4 v = make_float4(x4.x * lr32, x4.y * lr32, x4.z * lr32, x4.w * lr32);
                                dl = make_float4(dl.x + v.x, dl.y + v.y, dl.z + v.z, dl.w + v.w);
                            }
                            const float f = 1.0f - (float)cnt * lr32;
                            float4 *pc4 = reinterpret_cast<float4 *>(sC + c * TS_COLS + wave * 256 + ((lane ^ (c & 7)) << 2));
                            const float4 c4 = *pc4;
                            *pc4 = make_float4(c4.x * f + dl.x, c4.y * f + dl.y, c4.z * f + dl.z, c4.w * f + dl.w);
                        }
                        if (tid == 0) sCnt[c] = sCnt[c] + (float)cnt;
                        pend |= 1u << c;
                    }
                }
            }
        }
        __syncthreads();  // the updated centres and counts are in place before the next step reads them
        if (PROF) {
            const long long c5 = TS_CLK();
            pr[4] += c5 - c4, pr[5] += c5 - c0;
        }
    }
#undef TS_CLK
    if (PROF && tid == 0 && cg == 1 % ncg && rg == 0)  // one pair of workgroups reports: ch = 1 into prof[0..5], ch = 0 into prof_wg[0]
        for (int i = 0; i < 6; ++i) (ch ? ctl->prof : ctl->prof_wg[0])[i] = (unsigned long long)pr[i];
    if (pend && !sDead) {  // uniform
        if (!__all(refresh_norms(pend, (unsigned)T + 1u)) && lane == 0) sDead = 1;
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // a row fetch may still be in flight after an early exit
    __syncthreads();
    // ---- write the owned state back (one replica per centre group and column half)
    if (rg == 0 && !sDead) {
        if (active)
            for (int c = 0; c < nck; ++c) {
                const float4 v = *reinterpret_cast<const float4 *>(sC + c * TS_COLS + wave * 256 + ((lane ^ (c & 7)) << 2));
                *reinterpret_cast<float4 *>(centers + (size_t)(kbase + c) * d + coff + wave * 256 + (lane << 2)) = v;
            }
        if (ch == 1 && tid < nck) {
            cn[kbase + tid] = sCn[tid];
            counts[kbase + tid] = sCnt[tid];
        }
    }
}

__global__ void k_fill_u64(unsigned long long *p, int n, unsigned long long v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace

int acav_kmeans::refresh_cn()
{
    hipLaunchKernelGGL(k_row_norm2, dim3((K + 7) / 8), dim3(256), 0, ctx.stream, centers.as<float>(), K, d,
                       cn.as<float>());
    ACAV_HIP_TRY(hipGetLastError());
    return ACAV_OK;
}

ACAV_EXPORT int acav_kmeans_create(acav_kmeans **out, int device, int k, int d, const float *centers0,
                                   void *stream)
{
    ACAV_REQUIRE(out && centers0, ACAV_EINVAL, "NULL argument");
    ACAV_REQUIRE(k > 0 && d > 0, ACAV_EINVAL, "k and d must be positive (k=%d d=%d)", k, d);
    ACAV_REQUIRE(d <= 16384, ACAV_EINVAL, "d=%d exceeds the supported 16384", d);
    // (the assign kernels address a centre row as a 32-bit element offset from the matrix: 4 GB of centres is far beyond any use)
    ACAV_REQUIRE((int64_t)k * d <= ((int64_t)1 << 30), ACAV_EINVAL, "k * d = %lld exceeds the supported 2^30 elements", (long long)k * d);
    acav_kmeans *km = new (std::nothrow) acav_kmeans;
    ACAV_REQUIRE(km, ACAV_ENOMEM, "out of host memory");
    int rc = km->ctx.init(device, stream);
    if (rc != ACAV_OK) {
        delete km;
        return rc;
    }
    km->K = k;
    km->d = d;
    km->bind_buffers();
    auto fail = [&](int code) {
        km->ctx.fini();
        delete km;
        return code;
    };
    if ((rc = km->centers.ensure(sizeof(float) * (size_t)k * d)) != ACAV_OK) return fail(rc);
    if ((rc = km->cn.ensure(sizeof(float) * k)) != ACAV_OK) return fail(rc);
    if ((rc = km->counts.ensure(sizeof(float) * k)) != ACAV_OK) return fail(rc);
    if ((rc = km->scalars.ensure(sizeof(StepScalars))) != ACAV_OK) return fail(rc);
    if ((rc = acav_kmeans_set_state(km, centers0, nullptr, 0, 0)) != ACAV_OK) return fail(rc);
    if (hipMemsetAsync(km->counts.p, 0, sizeof(float) * k, km->ctx.stream) != hipSuccess) return fail(ACAV_EHIP);
    *out = km;
    return ACAV_OK;
}

ACAV_EXPORT int acav_kmeans_destroy(acav_kmeans *km)
{
    if (!km) return ACAV_OK;
    (void)hipSetDevice(km->ctx.device);
    (void)hipStreamSynchronize(km->ctx.stream);
    if (km->ev_f0) (void)hipEventDestroy(km->ev_f0);
    if (km->ev_f1) (void)hipEventDestroy(km->ev_f1);
    km->ctx.fini();
    delete km;
    return ACAV_OK;
}

ACAV_EXPORT int acav_kmeans_shape(const acav_kmeans *km, int *k, int *d)
{
    ACAV_REQUIRE(km, ACAV_EINVAL, "handle is NULL");
    if (k) *k = km->K;
    if (d) *d = km->d;
    return ACAV_OK;
}
ACAV_EXPORT int acav_kmeans_stream(const acav_kmeans *km, void **stream)
{
    ACAV_REQUIRE(km && stream, ACAV_EINVAL, "NULL argument");
    *stream = km->ctx.stream;
    return ACAV_OK;
}
ACAV_EXPORT int acav_kmeans_sync(acav_kmeans *km)
{
    ACAV_REQUIRE(km, ACAV_EINVAL, "handle is NULL");
    ACAV_HIP_TRY(hipStreamSynchronize(km->ctx.stream));
    return ACAV_OK;
}
ACAV_EXPORT int acav_kmeans_timer_begin(acav_kmeans *km)
{
    ACAV_REQUIRE(km, ACAV_EINVAL, "handle is NULL");
    return km->ctx.timer_begin();
}
ACAV_EXPORT int acav_kmeans_timer_end(acav_kmeans *km, float *ms)
{
    ACAV_REQUIRE(km && ms, ACAV_EINVAL, "NULL argument");
    return km->ctx.timer_end(ms);
}
ACAV_EXPORT int acav_kmeans_stats(acav_kmeans *km, int64_t *assign_launches, int64_t *step_launches)
{
    ACAV_REQUIRE(km, ACAV_EINVAL, "handle is NULL");
    if (assign_launches) *assign_launches = km->n_assign_launches;
    if (step_launches) *step_launches = km->n_step_launches;
    return ACAV_OK;
}

ACAV_EXPORT int acav_kmeans_train_stats(acav_kmeans *km, int64_t *persistent_launches, int64_t *persistent_fallbacks)
{
    ACAV_REQUIRE(km, ACAV_EINVAL, "handle is NULL");
    if (persistent_launches) *persistent_launches = km->n_persistent_launches;
    if (persistent_fallbacks) *persistent_fallbacks = km->n_persistent_fallbacks;
    return ACAV_OK;
}


ACAV_EXPORT int acav_kmeans_set_hyper(acav_kmeans *km, int initial_rounds, double reinit_p, double reinit_r)
{
    ACAV_REQUIRE(km, ACAV_EINVAL, "handle is NULL");
    ACAV_REQUIRE(initial_rounds >= 0 && reinit_r != 0.0, ACAV_EINVAL, "bad hyper-parameters");
    km->initial_rounds = initial_rounds;
    km->reinit_p = reinit_p;
    km->reinit_r = reinit_r;
    return ACAV_OK;
}

ACAV_EXPORT int acav_kmeans_get_state(acav_kmeans *km, float *centers, float *counts, int64_t *count,
                                      int64_t *fallback)
{
    ACAV_REQUIRE(km, ACAV_EINVAL, "handle is NULL");
    if (!centers && !counts && !fallback) {  // `count` alone is host state: no device round trip, no synchronisation
        if (count) *count = km->count;
        return ACAV_OK;
    }
    ACAV_HIP_TRY(hipSetDevice(km->ctx.device));
    if (centers) ACAV_TRY(from_device(centers, km->centers.p, sizeof(float) * (size_t)km->K * km->d, km->ctx.stream));
    if (counts) ACAV_TRY(from_device(counts, km->counts.p, sizeof(float) * km->K, km->ctx.stream));
    StepScalars s{};
    if (fallback) ACAV_HIP_TRY(hipMemcpyAsync(&s, km->scalars.p, sizeof(s), hipMemcpyDeviceToHost, km->ctx.stream));
    ACAV_HIP_TRY(hipStreamSynchronize(km->ctx.stream));
    if (count) *count = km->count;
    if (fallback) *fallback = (int64_t)s.fallback;
    return ACAV_OK;
}

ACAV_EXPORT int acav_kmeans_set_state(acav_kmeans *km, const float *centers, const float *counts,
                                      int64_t count, int64_t fallback)
{
    ACAV_REQUIRE(km, ACAV_EINVAL, "handle is NULL");
    ACAV_REQUIRE(count >= 0, ACAV_EINVAL, "count must be >= 0");
    ACAV_HIP_TRY(hipSetDevice(km->ctx.device));
    hipStream_t st = km->ctx.stream;
    if (centers) {
        const size_t bytes = sizeof(float) * (size_t)km->K * km->d;
        ACAV_HIP_TRY(hipMemcpyAsync(km->centers.p, centers, bytes,
                                    is_device_ptr(centers) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
        ACAV_TRY(km->refresh_cn());
        km->cb16_valid = false;
    }
    if (counts) {
        ACAV_HIP_TRY(hipMemcpyAsync(km->counts.p, counts, sizeof(float) * km->K,
                                    is_device_ptr(counts) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, st));
    }
    StepScalars s{};
    s.fallback = fallback;
    ACAV_HIP_TRY(hipMemcpyAsync(km->scalars.p, &s, sizeof(s), hipMemcpyHostToDevice, st));
    ACAV_HIP_TRY(hipStreamSynchronize(st));  // host staging buffers may go away
    km->count = count;
    // the filter's centre copy depends on all three: centred or not is decided by whether any usage count is below the
    // threshold (count / K)^p -- a stale "centred" copy under a discount would drop a row constant that no longer cancels
    km->cb16_valid = false;
    return km->prepare_filter();  // no-op while the state is still in its warm-up phase or the shape does not take the filter
}


// one add() on device-resident x [b,d]; forced (device) optional; xn_dev: ||x||^2 of the b rows
// when the caller already has them (bulk training), else computed here.
static int step_device(acav_kmeans *km, const float *dx, int64_t b, double lr, const int64_t *dforced,
                       const float *xn_dev)
{
    ACAV_REQUIRE(b > 0 && b <= SU_MAXB, ACAV_EINVAL, "batch size %lld outside 1..%d", (long long)b, SU_MAXB);
    ACAV_REQUIRE(km->K <= SU_MAXK, ACAV_EINVAL, "k=%d exceeds the %d supported by the SGD step kernels", km->K, SU_MAXK);
    hipStream_t st = km->ctx.stream;
    if (!km->keys.p) {
        ACAV_TRY(km->keys.ensure(sizeof(unsigned long long) * 2 * SU_MAXB));
        hipLaunchKernelGGL(k_fill_u64, dim3((2 * SU_MAXB + 255) / 256), dim3(256), 0, st,
                           km->keys.as<unsigned long long>(), 2 * SU_MAXB, ~0ull);
        ACAV_HIP_TRY(hipGetLastError());
        km->key_phase = 0;
    }
    unsigned long long *kcur = km->keys.as<unsigned long long>() + (size_t)km->key_phase * SU_MAXB;
    unsigned long long *knext = km->keys.as<unsigned long long>() + (size_t)(km->key_phase ^ 1) * SU_MAXB;
    if (!dforced) {
        ACAV_REQUIRE(!km->warm(), ACAV_ESTATE, "warm-up step needs forced labels (acav_rng_warmup_best)");
        if (!xn_dev) {
            ACAV_TRY(km->xn.ensure(sizeof(float) * SU_MAXB));
            hipLaunchKernelGGL(k_row_norm2, dim3((unsigned)((b + 7) / 8)), dim3(256), 0, st, dx, (int)b, km->d,
                               km->xn.as<float>());
            xn_dev = km->xn.as<float>();
        }
        const bool dma_ok = (km->d & 3) == 0 && ((uintptr_t)dx & 15) == 0;
        // large batches (DDP global batches): row groups per workgroup so that the grid is about one round of workgroups
        int rg = 1;
        if (dma_ok && km->d <= TP_DS && b >= 128) {
            const int64_t cgs = (km->K + TP_NC - 1) / TP_NC;
            while (rg < 8 && cgs * ((b + TP_NR * rg - 1) / (TP_NR * rg)) > 256) rg *= 2;
        }
        if (rg > 1) {
            const bool ragged = (km->d & 255) != 0;
            auto kern = rg == 2 ? (ragged ? k_step_dist_dma_rg<2, true> : k_step_dist_dma_rg<2, false>)
                      : rg == 4 ? (ragged ? k_step_dist_dma_rg<4, true> : k_step_dist_dma_rg<4, false>)
                                : (ragged ? k_step_dist_dma_rg<8, true> : k_step_dist_dma_rg<8, false>);
            const int smem = (int)sizeof(float) * (TP_NC + 2 * TP_NR) * TP_DS;
            if (!km->rg_attr_set) {
                ACAV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_step_dist_dma_rg<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                ACAV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_step_dist_dma_rg<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                ACAV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_step_dist_dma_rg<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                ACAV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_step_dist_dma_rg<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                ACAV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_step_dist_dma_rg<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                ACAV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_step_dist_dma_rg<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
                km->rg_attr_set = true;
            }
            hipLaunchKernelGGL(kern, dim3((unsigned)((km->K + TP_NC - 1) / TP_NC), (unsigned)((b + TP_NR * rg - 1) / (TP_NR * rg))), dim3(256),
                               smem, st, dx, (int)b, km->d, km->centers.as<float>(), km->cn.as<float>(), km->counts.as<float>(), xn_dev,
                               km->K, km->threshold(), (float)km->reinit_r, kcur);
        } else if (dma_ok) {
            hipLaunchKernelGGL(k_step_dist_dma, dim3((km->K + SD_NC - 1) / SD_NC, (unsigned)((b + SD_NR - 1) / SD_NR)),
                               dim3(256), 0, st, dx, (int)b, km->d, km->centers.as<float>(), km->cn.as<float>(),
                               km->counts.as<float>(), xn_dev, km->K, km->threshold(), (float)km->reinit_r, kcur);
        } else {
            hipLaunchKernelGGL(k_step_dist_mfma, dim3((km->K + 31) / 32, (unsigned)((b + 31) / 32)), dim3(256), 0, st, dx,
                               (int)b, km->d, km->centers.as<float>(), km->cn.as<float>(), km->counts.as<float>(), xn_dev,
                               km->K, km->threshold(), (float)km->reinit_r, kcur);
        }
        ACAV_HIP_TRY(hipGetLastError());
    }
    const size_t smem = sizeof(int) * 2 * (size_t)b + sizeof(int) * 2 * (size_t)km->K + sizeof(float) * (size_t)km->d + 32;
    hipLaunchKernelGGL(k_step_update, dim3((unsigned)b), dim3(256), smem, st, dx, (int)b, km->d,
                       km->centers.as<float>(), km->cn.as<float>(), km->counts.as<float>(), km->K, kcur,
                       dforced ? (unsigned long long *)nullptr : knext, dforced, lr, km->scalars.as<StepScalars>());
    ACAV_HIP_TRY(hipGetLastError());
    if (!dforced) km->key_phase ^= 1;
    km->count += b;
    km->cb16_valid = false;
    km->n_step_launches += 1;
    return ACAV_OK;
}

ACAV_EXPORT int acav_kmeans_step(acav_kmeans *km, const float *x, int64_t b, double lr,
                                 const int64_t *forced_best, float *mean_dist)
{
    ACAV_REQUIRE(km && x, ACAV_EINVAL, "NULL argument");
    ACAV_HIP_TRY(hipSetDevice(km->ctx.device));
    hipStream_t st = km->ctx.stream;
    const void *dx = nullptr, *df = nullptr;
    ACAV_TRY(to_device(x, sizeof(float) * (size_t)b * km->d, km->stage_x, st, &dx));
    if (forced_best) {
        for (int64_t i = 0; !is_device_ptr(forced_best) && i < b; ++i)
            ACAV_REQUIRE(forced_best[i] >= 0 && forced_best[i] < km->K, ACAV_EINVAL, "label %lld out of range",
                         (long long)forced_best[i]);
        ACAV_TRY(to_device(forced_best, sizeof(int64_t) * (size_t)b, km->stage_forced, st, &df));
    }
    ACAV_TRY(step_device(km, static_cast<const float *>(dx), b, lr, static_cast<const int64_t *>(df), nullptr));
    if (mean_dist && !forced_best) {
        StepScalars s{};
        ACAV_HIP_TRY(hipMemcpyAsync(&s, km->scalars.p, sizeof(s), hipMemcpyDeviceToHost, st));
        ACAV_HIP_TRY(hipStreamSynchronize(st));
        *mean_dist = s.mean;
    } else if (dx != x || (forced_best && df != forced_best)) {
        ACAV_HIP_TRY(hipStreamSynchronize(st));
    }
    return ACAV_OK;
}

ACAV_EXPORT int acav_kmeans_apply_update(acav_kmeans *km, const float *x, int64_t b, const int64_t *best, double lr)
{
    ACAV_REQUIRE(best, ACAV_EINVAL, "best is NULL");
    return acav_kmeans_step(km, x, b, lr, best, nullptr);
}

// One training call in two halves, so that several clusterings can have their persistent kernels in flight together
// (acav_kmeans_train_multi): train_launch() stages the inputs and, when the call is eligible, enqueues the persistent
// kernel WITHOUT waiting; train_finish() waits, checks the kernel's error flag and -- if the launch gave up or was not
// eligible -- runs the per-step launch path.  `budget` = workgroups that may still become co-resident on the device.
struct TrainCall {
    int64_t n = 0, b = 0, steps = 0, need = 0;
    double lr = 0.0;
    const float *fx = nullptr;
    const int64_t *dw = nullptr;
    const void *x_user = nullptr, *w_user = nullptr;
    int nwg = 0;
    bool launched = false, prof = false, split_prof = false, active = false;
    bool is_split = false;  // the column-split kernel: one 130 KB workgroup on every CU, 256 registers per lane
    bool shared = false;    // launched BESIDE such a kernel, in the LDS it leaves (books no CUs of the budget)
    std::vector<float> thr;  // staging of the per-step thresholds: alive until the launch has been waited for
};

static int train_launch(acav_kmeans *km, TrainCall &tc, const float *x, int64_t n, int64_t b, double lr,
                        const int64_t *warm_best, int64_t n_warm, int *budget, int share_lds = 0)
{
    // a pending clustering of acav_kmeans_train_multi is tried again whenever CUs come back: its inputs are staged ONCE (device copy of
    // host rows, warm-up labels, the row norms over the whole epoch) -- a retry only re-evaluates the fit and enqueues the kernel
    const bool again = tc.active && !tc.launched && tc.x_user == x && tc.w_user == warm_best && tc.n == n && tc.b == b && tc.lr == lr;
    if (!again) tc = TrainCall();
    ACAV_REQUIRE(km, ACAV_EINVAL, "handle is NULL");
    ACAV_REQUIRE(n >= 0 && b > 0 && n_warm >= 0, ACAV_EINVAL, "bad sizes");
    ACAV_REQUIRE(b <= SU_MAXB, ACAV_EINVAL, "batch size %lld above the supported %d", (long long)b, SU_MAXB);
    if (n / b == 0) return ACAV_OK;  // drop_last: not even one full batch (run_clustering.py:204)
    ACAV_REQUIRE(x, ACAV_EINVAL, "NULL argument");
    ACAV_HIP_TRY(hipSetDevice(km->ctx.device));
    hipStream_t st = km->ctx.stream;
    const int64_t steps = n / b;  // drop_last=True (run_clustering.py:204)
    // warm-up steps needed from the current count
    int64_t need = 0;
    {
        const int64_t lim = (int64_t)km->initial_rounds * km->K;
        if (km->count < lim) need = (lim - km->count + b - 1) / b;
        if (need > steps) need = steps;
    }
    ACAV_REQUIRE(n_warm == need, ACAV_EINVAL, "need labels for %lld warm-up steps, got %lld", (long long)need,
                 (long long)n_warm);
    ACAV_REQUIRE(need == 0 || warm_best, ACAV_EINVAL, "warm_best is NULL");
    if (!again) {
        const void *dx = nullptr, *dw = nullptr;
        ACAV_TRY(to_device(x, sizeof(float) * (size_t)n * km->d, km->stage_x, st, &dx));
        if (need) ACAV_TRY(to_device(warm_best, sizeof(int64_t) * (size_t)need * b, km->stage_forced, st, &dw));
        tc.active = true;
        tc.n = n, tc.b = b, tc.steps = steps, tc.need = need, tc.lr = lr, tc.fx = static_cast<const float *>(dx), tc.dw = static_cast<const int64_t *>(dw);
        tc.x_user = x, tc.w_user = warm_best;
    }
    const float *fx = tc.fx;
    // ||x||^2 of every row once per call (it does not depend on the centres)
    if (steps > need && !again) {
        const int64_t rows = steps * b;
        ACAV_TRY(km->xn.ensure(sizeof(float) * (size_t)(rows > SU_MAXB ? rows : SU_MAXB)));
        hipLaunchKernelGGL(k_row_norm2, dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, st, fx, (int)rows, km->d,
                           km->xn.as<float>());
        ACAV_HIP_TRY(hipGetLastError());
    }
    // persistent path: the whole call in one launch, centres resident in LDS (k_train_persistent)
    const int nwg = ((km->K + TP_NC - 1) / TP_NC) * (int)((b + TP_NR - 1) / TP_NR);
    tc.nwg = nwg;
    const char *nop = getenv("ACAV_NO_PERSISTENT");
    const bool prof = getenv("ACAV_PROFILE_STEPS") != nullptr;
    tc.prof = prof;
    const bool ragged = (km->d & 255) != 0;
    auto tkern = ragged ? (prof ? k_train_persistent<true, true> : k_train_persistent<true, false>)
                        : (prof ? k_train_persistent<false, true> : k_train_persistent<false, false>);
    // every workgroup of the launch must be resident at once: the occupancy query gives the workgroups one CU can
    // hold (1: 97 KB of LDS each), times the CUs of the device, minus what other launches of this call already hold.
    // Other streams may still hold CUs: the kernel's spins are bounded and a launch that gave up is re-run by
    // train_finish() on the per-step path from the saved state.
    if (km->num_cus == 0) {
        hipDeviceProp_t prop;
        ACAV_HIP_TRY(hipGetDeviceProperties(&prop, km->ctx.device));
        km->num_cus = prop.multiProcessorCount;
    }
    int occ = 0;
    ACAV_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(tkern), 256, 0));
    const int room = budget ? *budget : occ * km->num_cus;
    const bool shape_ok = steps > 0 && !(nop && nop[0] == '1') && (km->d % 4) == 0 && km->d <= TP_DS && b <= TP_MAXB &&
                          ((uintptr_t)fx & 15) == 0;
    // the exchange sweep of k_train_persistent reads 2 x TP_SW = 32 centre groups per row: K <= 256; more groups go to the
    // wide kernel (64 groups of NCP x 8 centres)
    // (ACAV_FORCE_WIDE=1: experiments -- the 16-centre forms for shapes the narrow kernel would take)
    const char *vfw = getenv("ACAV_FORCE_WIDE");
    const bool narrow_ok = (km->K + TP_NC - 1) / TP_NC <= 32 && !(vfw && vfw[0] == '1');
    bool persistent = shape_ok && narrow_ok && nwg <= room && nwg <= occ * km->num_cus;
    // more 8-centre groups than CUs (K = 1024): NCP x 8 centres per workgroup (k_train_persistent_wide) -- the smallest
    // NCP whose grid fits 3/4 of the device, else the whole device, within the LDS of a CU
    int ncp = 1, nrp = 1, ds = 0, wide_wg = 0;
    size_t wide_smem = 0;
    bool wide = false, one_x = false;
    // rows wider than 1024 columns (round 4: the real SlowFast widths 1408 / 2304, and d = 2048 below K = 1024): the wide kernel
    // with ONE centre pass per workgroup (or two), its waves looping over the 256-column blocks, one batch-row buffer when two do
    // not fit -- no column split, no hand-off between workgroups.  ACAV_TALL=0 switches it off (A/B against the split kernel /
    // the per-step launches).
    const char *vtall = getenv("ACAV_TALL");
    if (!persistent && steps > 0 && !(nop && nop[0] == '1') && !(vtall && vtall[0] == '0') && (km->d % 4) == 0 && km->d > TP_DS &&
        b <= TP_MAXB && ((uintptr_t)fx & 15) == 0) {
        ds = ((km->d + 255) / 256) * 256;
        const int nblk_t = ds / 256, rgroups = (int)((b + TP_NR - 1) / TP_NR);
        for (int c : {1, 2}) {
            for (int xrows : {16, 8}) {
                const int groups = (km->K + 8 * c - 1) / (8 * c);
                const size_t smem = sizeof(float) * ((size_t)(8 * c + xrows) * ds + 2 * 8 * c + 64 * (c == 2 ? 4 : c) * nblk_t + 32);  // (16 centres: matrix-core tile sums, 4 per lane and block)
                if (!wide && groups <= 64 && smem <= 160 * 1024 - 1024 && groups * rgroups <= km->num_cus && groups * rgroups <= room) {
                    ncp = c, wide_wg = groups * rgroups, wide_smem = smem, one_x = xrows == 8;
                    wide = persistent = true;
                }
            }
        }
    }
    if (shape_ok && !persistent && (nwg > occ * km->num_cus || !narrow_ok)) {
        ds = ((km->d + 255) / 256) * 256;
        const int rgroups = (int)((b + TP_NR - 1) / TP_NR);
        int best_ncp = 0;
        const char *fncp = getenv("ACAV_WIDE_NCP");  // experiments: force the centres per workgroup (2, 4, 8)
        // a call for one clustering takes the smallest NCP that fits the device (K = 1024, d = 128: 9.8 us per step on 256
        // workgroups, 11.4 on 128); with several clusterings in one call (budget) the grids stay within 3/4 of it first,
        // so that two of them run side by side
        for (int pass = (fncp || !budget) ? 1 : 0; pass < 2 && !best_ncp; ++pass)
            for (int c : {2, 4, 8}) {
                if (fncp && atoi(fncp) != c) continue;
                const int groups = (km->K + 8 * c - 1) / (8 * c);
                const size_t smem = sizeof(float) * ((size_t)(8 * c + 16) * ds + 2 * 8 * c + 256 * (c == 2 ? 4 : c) + 32);
                const int lim = pass == 0 ? (3 * km->num_cus) / 4 : km->num_cus;
                if (groups <= 64 && smem <= 160 * 1024 - 1024 && groups * rgroups <= lim && groups * rgroups <= room) {
                    best_ncp = c, wide_wg = groups * rgroups, wide_smem = smem;
                    break;
                }
            }
        // K = 1024 at 768 < d <= 1024 (cfg5): NCP = 2 needs the whole device (64 centre groups x 4 row groups) and more centres
        // per workgroup do not fit next to two row buffers -- with SEVERAL clusterings in the call the two-row-pass form
        // (16 centres x 16 rows, one row buffer, 64 x 2 = 128 workgroups) lets two of them run side by side: 14.x us per step of
        // the PAIR instead of 2 x 13.1.  ACAV_WIDE_NRP=2 forces it for a single clustering, =1 switches it off (A/B).
        const char *fnrp = getenv("ACAV_WIDE_NRP");
        const bool nrp_forced = fnrp && fnrp[0] == '2', nrp_off = fnrp && fnrp[0] == '1';
        if (ds == TS_COLS && !nrp_off && !fncp && (nrp_forced || !best_ncp || wide_wg > (3 * km->num_cus) / 4)) {
            // (round 6: also for a LONE clustering -- with the tile on the matrix core the 16 x 16 form costs no more FMA time than 16 x 8 and
            // has half the workgroups in the exchange: K = d = 1024 alone 8.6 vs 9.2 us per step)
            const int groups = (km->K + 15) / 16, rg2 = (int)((b + 15) / 16);
            const size_t smem = sizeof(float) * ((size_t)(16 + 16) * ds + 2 * 16 + 4 * 64 * (ds / 256) + 32);
            if (groups <= 64 && smem <= 160 * 1024 - 1024 && groups * rg2 <= (3 * km->num_cus) / 4 && groups * rg2 <= room) {
                best_ncp = 2, nrp = 2, one_x = true, wide_wg = groups * rg2, wide_smem = smem;
            }
        }
        // Round 6 (ACAV_TRAIN_SHARE_CU=0 switches it off): no CUs left, but the ONE launch in flight is the column-split kernel (a 130 KB workgroup
        // on every CU, 252 + 4 registers per lane since the exchange rewrite) -- a 16-centre form with one row buffer fits the LDS it
        // leaves (26 KB at ds = 256) and the register file beside it (231-243 + 4: profiles/r06_train_regs.txt), one workgroup per CU:
        // cfg4's 2048-d and 128-d views train side by side instead of one after the other.
        if (!best_ncp && share_lds > 0 && ds <= 512) {
            const int groups = (km->K + 15) / 16;
            const size_t smem = sizeof(float) * ((size_t)(16 + 8) * ds + 2 * 16 + 4 * 64 * (size_t)(ds / 256) + 32);  // (tile sums: 4 x 64 per column block)
            if (groups <= 64 && (int)smem + 512 <= share_lds && groups * rgroups <= km->num_cus) {
                best_ncp = 2, nrp = 1, one_x = true, wide_wg = groups * rgroups, wide_smem = smem;
                tc.shared = true;
                if (getenv("ACAV_TRAIN_SHARE_DEBUG")) fprintf(stderr, "[acav] shared launch: d = %d, K = %d, %d workgroups of %zu B beside the split kernel (%d B free per CU)\n", km->d, km->K, wide_wg, smem, share_lds);
            }
        }
        if (best_ncp) ncp = best_ncp, persistent = wide = true;
    }
    // 1024 < d <= 2048 (cfg4's visual view): the columns are split over pairs of workgroups (k_train_persistent_split)
    bool split = false;
    int split_wg = 0;
    const int s_ncg = (km->K + TS_NC - 1) / TS_NC, s_nrg = (int)((b + TS_NR - 1) / TS_NR);
    // (worth it from K = 512 on: below that the per-step launches are as fast -- 14.5 us at K = 256 -- because the two
    // dependent hand-offs of a split step cost more than the launches they replace; ACAV_SPLIT_MINK overrides)
    const char *smk = getenv("ACAV_SPLIT_MINK");
    const int split_mink = smk ? atoi(smk) : 512;
    if (!persistent && steps > 0 && !(nop && nop[0] == '1') && km->d > TP_DS && km->d <= 2 * TP_DS && (km->d & 255) == 0 &&
        b <= TP_MAXB && ((uintptr_t)fx & 15) == 0 && s_ncg <= 2 * TPW_SW && km->K >= split_mink) {
        auto sk = prof ? k_train_persistent_split<true> : k_train_persistent_split<false>;
        ACAV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(sk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TS_SMEM));
        int occ2 = 0;
        ACAV_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ2, reinterpret_cast<const void *>(sk), 256, TS_SMEM));
        split_wg = s_ncg * s_nrg * 2;
        if (occ2 >= 1 && split_wg <= occ2 * km->num_cus && split_wg <= (budget ? *budget : occ2 * km->num_cus)) split = persistent = true;
    }
    if (!persistent) return ACAV_OK;
    tc.nwg = split ? split_wg : wide ? wide_wg : nwg;
    tc.is_split = split;
    if (budget && !tc.shared) *budget -= tc.nwg;
    tc.thr.resize((size_t)steps);
    for (int64_t t = 0; t < steps; ++t)
        tc.thr[(size_t)t] = (float)pow((double)(km->count + t * b) / (double)km->K, km->reinit_p);
    ACAV_TRY(km->thr.ensure(sizeof(float) * (size_t)steps));
    ACAV_HIP_TRY(hipMemcpyAsync(km->thr.p, tc.thr.data(), sizeof(float) * (size_t)steps, hipMemcpyHostToDevice, st));
    ACAV_TRY(km->ctl.ensure(sizeof(TrainCtl)));
    ACAV_HIP_TRY(hipMemsetAsync(km->ctl.p, 0, sizeof(TrainCtl), st));  // err = 0, every granule tag = 0 (never a live tag)
    // the state as it is now, in case the launch gives up (1 MB at K=256, d=1024: a few microseconds)
    const size_t cbytes = sizeof(float) * (size_t)km->K * km->d, kbytes = sizeof(float) * (size_t)km->K;
    ACAV_TRY(km->backup.ensure(cbytes + 2 * kbytes + sizeof(StepScalars)));
    char *bk = km->backup.as<char>();
    ACAV_HIP_TRY(hipMemcpyAsync(bk, km->centers.p, cbytes, hipMemcpyDeviceToDevice, st));
    ACAV_HIP_TRY(hipMemcpyAsync(bk + cbytes, km->cn.p, kbytes, hipMemcpyDeviceToDevice, st));
    ACAV_HIP_TRY(hipMemcpyAsync(bk + cbytes + kbytes, km->counts.p, kbytes, hipMemcpyDeviceToDevice, st));
    ACAV_HIP_TRY(hipMemcpyAsync(bk + cbytes + 2 * kbytes, km->scalars.p, sizeof(StepScalars), hipMemcpyDeviceToDevice, st));
    if (split) {
        tc.prof = false;
        tc.split_prof = prof;
        const size_t t0b = sizeof(unsigned long long) * (size_t)s_ncg * s_nrg * 256, nrb = sizeof(unsigned long long) * (size_t)s_ncg * s_nrg * TS_NC * 32;
        ACAV_TRY(km->split_rings.ensure(t0b + nrb));
        ACAV_HIP_TRY(hipMemsetAsync(km->split_rings.p, 0, t0b + nrb, st));  // every hand-off tag = 0 (never a live tag)
        unsigned long long *t0r = km->split_rings.as<unsigned long long>();
        hipLaunchKernelGGL((prof ? k_train_persistent_split<true> : k_train_persistent_split<false>), dim3((unsigned)s_ncg, (unsigned)s_nrg, 2), dim3(256), TS_SMEM, st, fx,
                           km->xn.as<float>(), (int)b, km->d, km->K, km->centers.as<float>(), km->cn.as<float>(),
                           km->counts.as<float>(), km->thr.as<float>(), lr, (float)km->reinit_r, tc.dw, (int)need, (int)steps,
                           km->ctl.as<TrainCtl>(), km->scalars.as<StepScalars>(), t0r, t0r + t0b / sizeof(unsigned long long));
    } else if (!wide) {
        hipLaunchKernelGGL(tkern, dim3((km->K + TP_NC - 1) / TP_NC, (unsigned)((b + TP_NR - 1) / TP_NR)),
                           dim3(256), 0, st, fx, km->xn.as<float>(), (int)b, km->d, km->K, km->centers.as<float>(),
                           km->cn.as<float>(), km->counts.as<float>(), km->thr.as<float>(), lr, (float)km->reinit_r,
                           tc.dw, (int)need, (int)steps, km->ctl.as<TrainCtl>(), km->scalars.as<StepScalars>(), nwg);
    } else {
        tc.prof = false;  // the wide kernel carries no phase timers
        using WideKernel = void (*)(const float *, const float *, int, int, int, int, float *, float *, float *, const float *,
                                    double, float, const int64_t *, int, int, TrainCtl *, StepScalars *);
        WideKernel wk = nullptr;
        if (km->d > TP_DS) {  // the tall forms: one or two centre passes, one or two row buffers
            if (ragged) wk = ncp == 1 ? (one_x ? k_train_persistent_wide<true, 1, true> : k_train_persistent_wide<true, 1, false>)
                                      : (one_x ? k_train_persistent_wide<true, 2, true> : k_train_persistent_wide<true, 2, false>);
            else wk = ncp == 1 ? (one_x ? k_train_persistent_wide<false, 1, true> : k_train_persistent_wide<false, 1, false>)
                               : (one_x ? k_train_persistent_wide<false, 2, true> : k_train_persistent_wide<false, 2, false>);
        } else if (nrp == 2) wk = ragged ? k_train_persistent_wide<true, 2, true, 2> : k_train_persistent_wide<false, 2, true, 2>;
        else if (one_x && ncp == 2) wk = ragged ? k_train_persistent_wide<true, 2, true> : k_train_persistent_wide<false, 2, true>;  // (beside the split kernel)
        else if (ragged) wk = ncp == 2 ? k_train_persistent_wide<true, 2> : ncp == 4 ? k_train_persistent_wide<true, 4> : k_train_persistent_wide<true, 8>;
        else wk = ncp == 2 ? k_train_persistent_wide<false, 2> : ncp == 4 ? k_train_persistent_wide<false, 4> : k_train_persistent_wide<false, 8>;
        ACAV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(wk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wide_smem));
        hipLaunchKernelGGL(wk, dim3((km->K + 8 * ncp - 1) / (8 * ncp), (unsigned)((b + 8 * nrp - 1) / (8 * nrp))), dim3(256), wide_smem, st,
                           fx, km->xn.as<float>(), (int)b, km->d, ds, km->K, km->centers.as<float>(), km->cn.as<float>(),
                           km->counts.as<float>(), km->thr.as<float>(), lr, (float)km->reinit_r, tc.dw, (int)need, (int)steps,
                           km->ctl.as<TrainCtl>(), km->scalars.as<StepScalars>());
    }
    ACAV_HIP_TRY(hipGetLastError());
    tc.launched = true;
    return ACAV_OK;
}

static int train_finish(acav_kmeans *km, TrainCall &tc)
{
    if (!tc.active) return ACAV_OK;
    ACAV_HIP_TRY(hipSetDevice(km->ctx.device));
    hipStream_t st = km->ctx.stream;
    const int64_t steps = tc.steps, need = tc.need, b = tc.b;
    if (tc.launched) {
        const bool prof = tc.prof;
        const int nwg = tc.nwg;
        struct { unsigned err, pad[3]; unsigned long long prof[8]; unsigned long long prof_wg[256][8]; } head{};
        ACAV_HIP_TRY(hipMemcpyAsync(&head, km->ctl.p, prof ? sizeof(head) : 16, hipMemcpyDeviceToHost, st));
        ACAV_HIP_TRY(hipStreamSynchronize(st));  // also covers the thr staging vector
        if (prof) {
            const double den = (double)(steps > need ? steps - need : 1);
            fprintf(stderr, "[acav] persistent epoch: %lld steps; cycles/step: wait+dma-issue %.0f, fma %.0f, exchange %.0f, "
                            "update %.0f (fma chain alone %.0f, rows+apply %.0f), total %.0f\n", (long long)steps, head.prof[0] / den,
                    head.prof[1] / den, head.prof[2] / den, head.prof[3] / den, head.prof[5] / den, head.prof[6] / den,
                    head.prof[4] / den);
            double mx[4] = {0, 0, 0, 0}, mn[4] = {1e30, 1e30, 1e30, 1e30};
            for (int w = 0; w < nwg && w < 256; ++w)
                for (int q = 0; q < 4; ++q) {
                    const double v = head.prof_wg[w][q] / den;
                    mx[q] = v > mx[q] ? v : mx[q];
                    mn[q] = v < mn[q] ? v : mn[q];
                }
            fprintf(stderr, "[acav]   over workgroups: wait %.0f..%.0f fma %.0f..%.0f exch %.0f..%.0f upd %.0f..%.0f; sweep passes/step %.2f\n", mn[0],
                    mx[0], mn[1], mx[1], mn[2], mx[2], mn[3], mx[3], head.prof[7] / den);
        }
#ifdef ACAV_WIDE_PROF
        {
            struct { unsigned err, pad[3]; unsigned long long prof[8]; unsigned long long up[8]; } hw{};
            ACAV_HIP_TRY(hipMemcpy(&hw, km->ctl.p, sizeof(hw), hipMemcpyDeviceToHost));
            const double den = (double)(hw.prof[5] ? hw.prof[5] : 1);
            if (hw.prof[5])
                fprintf(stderr, "[acav] wide epoch (%d workgroups): cycles/step of workgroup (1, 0): row wait %.0f, fma %.0f, keys + exchange %.0f, update %.0f, "
                                "total %.0f; sweep passes/step %.2f; steps with an update of mine %.3f\n", tc.nwg, hw.prof[0] / den, hw.prof[1] / den,
                        hw.prof[2] / den, hw.prof[3] / den, hw.prof[4] / den, hw.prof[7] / den, hw.prof[6] / den);
            if (hw.prof[5])
                fprintf(stderr, "[acav]   wave 0 inside keys + exchange: keys epilogue %.0f, publish + sweep %.0f (first pass %.0f), sweep end -> past the barrier %.0f\n",
                        hw.up[4] / den, hw.up[5] / den, hw.up[6] / den, hw.up[7] / den);
            if (hw.prof[6]) {
                const double dt = (double)hw.prof[6];
                fprintf(stderr, "[acav]   per step WITH an update of mine: lr / labels %.0f, row loads issued -> landed %.0f, ballots + accumulate + centre rows "
                                "rewritten %.0f, closing barrier %.0f\n", hw.up[0] / dt, hw.up[1] / dt, hw.up[2] / dt, hw.up[3] / dt);
            }
        }
#endif
        if (tc.split_prof) {
            struct { unsigned err, pad[3]; unsigned long long prof[8]; unsigned long long wg0[8]; } hs{};
            ACAV_HIP_TRY(hipMemcpy(&hs, km->ctl.p, sizeof(hs), hipMemcpyDeviceToHost));
            const double den = (double)(steps > need ? steps - need : 1);
            for (int h2 = 1; h2 >= 0; --h2) {
                const unsigned long long *v = h2 ? hs.prof : hs.wg0;
                fprintf(stderr, "[acav] split epoch, column half %d: cycles/step: row wait %.0f, fma+norms %.0f, hand-off+keys %.0f, sweep %.0f, "
                                "update %.0f, total %.0f\n", h2, v[0] / den, v[1] / den, v[2] / den, v[3] / den, v[4] / den, v[5] / den);
            }
        }
        if (head.err == 0) {
            km->count += steps * b;
            km->cb16_valid = false;
            km->n_step_launches += steps;
            km->n_persistent_launches += 1;
            tc.active = false;
            return ACAV_OK;
        }
        // the launch gave up at an exchange (not all workgroups resident: the GPU is shared): restore the state it
        // started from and take the per-step launch path for this call
        km->n_persistent_fallbacks += 1;
        const size_t cbytes = sizeof(float) * (size_t)km->K * km->d, kbytes = sizeof(float) * (size_t)km->K;
        char *bk = km->backup.as<char>();
        ACAV_HIP_TRY(hipMemcpyAsync(km->centers.p, bk, cbytes, hipMemcpyDeviceToDevice, st));
        ACAV_HIP_TRY(hipMemcpyAsync(km->cn.p, bk + cbytes, kbytes, hipMemcpyDeviceToDevice, st));
        ACAV_HIP_TRY(hipMemcpyAsync(km->counts.p, bk + cbytes + kbytes, kbytes, hipMemcpyDeviceToDevice, st));
        ACAV_HIP_TRY(hipMemcpyAsync(km->scalars.p, bk + cbytes + 2 * kbytes, sizeof(StepScalars), hipMemcpyDeviceToDevice, st));
    }
    for (int64_t t = 0; t < steps; ++t) {
        const int64_t *f = t < need ? tc.dw + t * b : nullptr;
        ACAV_TRY(step_device(km, tc.fx + (size_t)t * b * km->d, b, tc.lr, f, km->xn.as<float>() + t * b));
    }
    if ((const void *)tc.fx != tc.x_user || (need && (const void *)tc.dw != tc.w_user)) ACAV_HIP_TRY(hipStreamSynchronize(st));
    tc.active = false;
    return ACAV_OK;
}

ACAV_EXPORT int acav_kmeans_train(acav_kmeans *km, const float *x, int64_t n, int64_t b, double lr,
                                  const int64_t *warm_best, int64_t n_warm)
{
    TrainCall tc;
    ACAV_TRY(train_launch(km, tc, x, n, b, lr, warm_best, n_warm, nullptr));
    ACAV_TRY(train_finish(km, tc));
    return km->prepare_filter();  // the sweep that follows an epoch finds the filter's centre copy ready (stream-ordered, ~25 us)
}

// The same call for SEVERAL clusterings at once (independent handles on one device, e.g. the audio and visual views of
// one batch stream, or the 5 + 5 layers of the real pipeline -- run_clustering.py:229-241 steps every clustering per
// batch): their persistent kernels are enqueued on their own streams before any is waited for, as many at a time as
// fit on the device together (a step is latency-bound -- 128 of 256 CUs at K = 256 -- so two chains side by side run at
// the speed of one).  Results are exactly those of acav_kmeans_train called per handle.
ACAV_EXPORT int acav_kmeans_train_multi(acav_kmeans *const *kms, int count, const float *const *xs, const int64_t *ns, int64_t b,
                                        double lr, const int64_t *const *warm_best, const int64_t *n_warm)
{
    ACAV_REQUIRE(kms && xs && ns && n_warm && count >= 0, ACAV_EINVAL, "NULL argument");
    if (count == 0) return ACAV_OK;
    for (int i = 0; i < count; ++i) {
        ACAV_REQUIRE(kms[i], ACAV_EINVAL, "handle %d is NULL", i);
        ACAV_REQUIRE(kms[i]->ctx.device == kms[0]->ctx.device, ACAV_EINVAL, "handle %d lives on another device", i);
        for (int e = 0; e < i; ++e) ACAV_REQUIRE(kms[e] != kms[i], ACAV_EINVAL, "handle %d given twice", i);
    }
    std::vector<TrainCall> calls((size_t)count);
    hipDeviceProp_t prop;
    ACAV_HIP_TRY(hipGetDeviceProperties(&prop, kms[0]->ctx.device));
    // Round 5: a running schedule instead of fixed groups.  The launches that fit the device's CUs together go out (in call
    // order, a later one that fits may pass an earlier one that does not), and as soon as ANY of them has finished its epoch the
    // pending ones are tried again in the CUs it gave back -- round 4 waited for a whole group and then ran the first clustering
    // that had not fitted ALONE before forming the next group (the ten K = 256 clusterings of the real pipeline: two, one, two,
    // one, two, one, one = 53 us per step of the ten instead of ~36).  The results do not depend on the schedule: the chains
    // are independent.
    const int cus = prop.multiProcessorCount;
    int budget = cus;
    const char *vshare = getenv("ACAV_TRAIN_SHARE_CU");
    const bool share_cu = !(vshare && vshare[0] == '0');  // on by default; ACAV_TRAIN_SHARE_CU=0: the views of such a pair one after the other
    std::vector<int> pending, inflight;
    for (int i = 0; i < count; ++i) pending.push_back(i);
    int tried_at = -1;  // budget at the last round of attempts: a pending launch is only tried again once CUs have come back
    while (!pending.empty() || !inflight.empty()) {
        if (!pending.empty() && budget != tried_at) {
            for (size_t q = 0; q < pending.size();) {
                const int i = pending[q];
                TrainCall &tc = calls[(size_t)i];
                // (the last clustering of the call with nothing else in flight is a call for ONE clustering: it gets the form that
                // is fastest alone, not the one that leaves room for a neighbour)
                int *bp = (inflight.empty() && pending.size() == 1) ? nullptr : &budget;
                // LDS a second launch may use on every CU: only beside ONE in-flight column-split kernel
                int share_lds = 0;
                if (share_cu && bp != nullptr && inflight.size() == 1 && calls[(size_t)inflight[0]].is_split && !calls[(size_t)inflight[0]].shared)
                    share_lds = 160 * 1024 - (int)TS_SMEM - 512;  // (LDS is handed out in 512-byte units; both kernels' static __shared__ is a few bytes)
                ACAV_TRY(train_launch(kms[i], tc, xs[i], ns[i], b, lr, warm_best ? warm_best[i] : nullptr, n_warm[i], bp, share_lds));
                if (tc.active && !tc.launched && inflight.empty() && bp != nullptr && budget == cus) {
                    // does not fit beside others even on an empty device: as a call for one clustering (whole-device forms)
                    ACAV_TRY(train_launch(kms[i], tc, xs[i], ns[i], b, lr, warm_best ? warm_best[i] : nullptr, n_warm[i], nullptr));
                    if (tc.launched) budget -= tc.nwg < budget ? tc.nwg : budget;  // (train_launch only books a budget it was given)
                }
                if (!tc.active) {  // not even one batch: nothing to do
                    pending.erase(pending.begin() + (long)q);
                } else if (tc.launched) {
                    inflight.push_back(i);
                    pending.erase(pending.begin() + (long)q);
                } else if (inflight.empty()) {  // no persistent form for this shape: the per-step path, now
                    ACAV_TRY(train_finish(kms[i], tc));
                    ACAV_TRY(kms[i]->prepare_filter());
                    pending.erase(pending.begin() + (long)q);
                } else {
                    ++q;  // waits for CUs
                }
            }
            tried_at = budget;
        }
        if (inflight.empty()) continue;
        // wait for the first launch to finish (the epochs take 0.1-10 s: a 50 us poll costs nothing)
        size_t done = inflight.size();
        while (done == inflight.size()) {
            for (size_t q = 0; q < inflight.size() && done == inflight.size(); ++q) {
                const hipError_t e = hipStreamQuery(kms[inflight[q]]->ctx.stream);
                if (e == hipSuccess) done = q;
                else if (e != hipErrorNotReady) ACAV_HIP_TRY(e);
            }
            if (done == inflight.size()) {
                (void)hipGetLastError();  // hipErrorNotReady is sticky in the last-error slot
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        }
        (void)hipGetLastError();  // a hipErrorNotReady of this sweep must not surface in the next launch's error check
        const int i = inflight[done];
        inflight.erase(inflight.begin() + (long)done);
        const int gave = calls[(size_t)i].shared ? 0 : calls[(size_t)i].nwg;  // (a shared launch booked no CUs)
        ACAV_TRY(train_finish(kms[i], calls[(size_t)i]));  // a launch that gave up is re-run on the per-step path in here
        ACAV_TRY(kms[i]->prepare_filter());
        budget = budget + gave > cus ? cus : budget + gave;
    }
    return ACAV_OK;
}
