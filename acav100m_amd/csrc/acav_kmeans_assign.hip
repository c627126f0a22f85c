// acav_kmeans_assign.hip -- the assign sweep of the SGD k-means (reference: clustering/code/sgd_clustering.py:63-79,
// KMeans.calc_best over a row partition) on gfx950: the bf16-MFMA filter, the exact fp32 sweep / re-check, the
// centre preparation, and their C-ABI entry points.  Training lives in acav_kmeans.hip.
#include <algorithm>
#include <cstddef>
#include <mutex>
#include <set>
#include <tuple>

#include "acav_kmeans_shared.h"

namespace {

// --------------------------------------------------------------------------- k_assign_f32
constexpr int AS_CG = 256;    // centres per group: 8 MFMA tiles, 2 per wave
constexpr int AS_BK = 32;     // feature columns per LDS stage (= the 32 canonical sumsq classes)
constexpr int AS_LD = 36;     // padded LDS row (floats): 144 B = 9 x 16 B -> conflict-free ds_read_b128

// LDS image of one staged row: within each group of 8 columns the even columns come first
// (pos = g8*8 + (e&1)*4 + (e>>1)), so that lane (i, h = lane>>5) reads with ONE ds_read_b128 the
// four values j = 8*g8 + 2m + h, m = 0..3, it must feed to four consecutive 32x32x2 MFMAs --
// keeping the FMA chain in ascending j.
__device__ __forceinline__ void stage_row4(float *srow, int q, float4 v)
{
    const int base = (q >> 1) * 8 + (q & 1) * 2;
    *reinterpret_cast<float2 *>(srow + base) = make_float2(v.x, v.z);      // even columns (h = 0)
    *reinterpret_cast<float2 *>(srow + base + 4) = make_float2(v.y, v.w);  // odd columns  (h = 1)
}

// GUARD = false: d % 32 == 0 and 16-byte aligned rows -> unconditional float4 loads with CLAMPED row /
// centre indices (rows >= n and centres >= K compute on a duplicate and are discarded), so the
// compiler keeps all 10 prefetch loads in flight under the MFMAs.  GUARD = true: element-guarded loads
// for ragged d (hipcc serialises those behind vmcnt(0) -- correctness path only).
struct AssignCtl;
__device__ __forceinline__ void assign_ctl_finish(AssignCtl *ctl, unsigned nblocks);
__device__ __forceinline__ AssignCtl *ctl_of_f32_count(const unsigned *f32_count);

// (GUARD: the ragged / unaligned form needs ~280 registers; cut for two waves per SIMD it carried 112 B of scratch -- and a queue
// waits ~130 us for a scratch allocation the first time such a kernel runs on it.  One wave per SIMD: the spills live in AGPRs.)
template <bool GUARD>
__global__ __launch_bounds__(256, GUARD ? 1 : 2) void k_assign_f32(const float *__restrict__ x, int64_t n, int d,
                                                       const float *__restrict__ centers,
                                                       const float *__restrict__ cn,
                                                       const float *__restrict__ counts, int K, float thr,
                                                       float r, int64_t *__restrict__ labels,
                                                       float *__restrict__ minval_out,
                                                       double *__restrict__ wg_sum,
                                                       const int *__restrict__ row_idx,
                                                       const unsigned *__restrict__ row_cnt)
{
    // row_idx != NULL: exact re-check pass of the bf16 filter -- the rows to label are x[row_idx[0 .. *row_cnt)].
    // The list length is only known on the device: that pass is launched with a small fixed grid whose workgroups
    // stride over the row tiles (an empty list costs one tiny launch, not a worst-case grid of early exits).
    if (row_idx) n = (int64_t)*row_cnt;
    const int64_t ntiles = (n + AS_ROWS - 1) / AS_ROWS;
    __shared__ __attribute__((aligned(16))) float sC[AS_CG * AS_LD];
    __shared__ __attribute__((aligned(16))) float sX[AS_ROWS * AS_LD];
    __shared__ float sXn[AS_ROWS];
    __shared__ float sCn[AS_CG];    // ||c||^2 of the current centre group
    __shared__ int sDisc[AS_CG];    // 1 = under-used centre (distance / r), -1 = centre index >= K
    __shared__ float sMinV[4][AS_ROWS];
    __shared__ int sMinI[4][AS_ROWS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;
    const bool vec_ok = (d & 3) == 0;

    // staging roles: 8 threads per row (q = float4 index inside the 32-column stage)
    const int srow = tid >> 3;  // 0..31
    const int sq = tid & 7;
    const int nchunks = (d + AS_BK - 1) / AS_BK;
    const int ngroups = (K + AS_CG - 1) / AS_CG;

  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {  // one trip except in the re-check pass
    const int64_t row0 = tile * AS_ROWS;
    float ssq[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float gbv = INFINITY;  // running best of row (tid) across centre groups, threads 0..63
    int gbi = 0x7fffffff;

    for (int cg = 0; cg < ngroups; ++cg) {
        const int kbase = cg * AS_CG;
        f32x16 acc[2][2], tot[2][2];  // acc: the running 256-column segment chain; tot: folded segments
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    acc[a][b][e] = 0.f;
                    tot[a][b][e] = 0.f;
                }

        {   // epilogue operands of this centre group: one element per thread, read back from LDS later
            const int k = kbase + tid;
            sCn[tid] = k < K ? cn[k] : 0.f;
            sDisc[tid] = k < K ? (counts[k] < thr ? 1 : 0) : -1;
        }
        float4 xr[2], cr[8];
        // !GUARD: the centre loads as (uniform 64-bit base of the stage) + (32-bit element offset of the thread's centre row): K d < 2^30
        // elements (acav_kmeans_create), so the eight addresses cost eight registers instead of eight pairs -- with pairs the kernel
        // spilled 39 dwords, and a kernel with ANY scratch makes its queue wait ~130 us for a scratch allocation the first time
        unsigned coff[8];
        if (!GUARD) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int k = kbase + srow + 32 * m;
                coff[m] = (unsigned)(k < K ? k : K - 1) * (unsigned)d + (unsigned)(sq * 4);
            }
        }
        auto issue_loads = [&](int c) {
            const int j = c * AS_BK + sq * 4;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int64_t gr = row0 + srow + 32 * m;
                if (GUARD) {
                    xr[m] = ld4_guard(x + (size_t)gr * d, j, d, gr < n, vec_ok);
                } else {
                    int64_t src = gr < n ? gr : n - 1;
                    if (row_idx) src = row_idx[src];
                    xr[m] = *reinterpret_cast<const float4 *>(x + (size_t)src * d + j);
                }
            }
            const float *cstage = centers + c * AS_BK;  // uniform
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int k = kbase + srow + 32 * m;
                if (GUARD)
                    cr[m] = ld4_guard(centers + (size_t)k * d, j, d, k < K, vec_ok);
                else
                    cr[m] = *reinterpret_cast<const float4 *>(cstage + coff[m]);
            }
        };
        issue_loads(0);

        for (int c = 0; c < nchunks; ++c) {
            __syncthreads();  // everyone finished reading the previous stage
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                stage_row4(sX + (srow + 32 * m) * AS_LD, sq, xr[m]);
                if (cg == 0) {
                    ssq[m][0] = __builtin_fmaf(xr[m].x, xr[m].x, ssq[m][0]);
                    ssq[m][1] = __builtin_fmaf(xr[m].y, xr[m].y, ssq[m][1]);
                    ssq[m][2] = __builtin_fmaf(xr[m].z, xr[m].z, ssq[m][2]);
                    ssq[m][3] = __builtin_fmaf(xr[m].w, xr[m].w, ssq[m][3]);
                }
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) stage_row4(sC + (srow + 32 * m) * AS_LD, sq, cr[m]);
            __syncthreads();
            if (c + 1 < nchunks) issue_loads(c + 1);  // in flight under the MFMAs below

            const float *pa0 = sC + ((2 * wave) * 32 + l31) * AS_LD + h * 4;
            const float *pa1 = pa0 + 32 * AS_LD;
            const float *pb0 = sX + l31 * AS_LD + h * 4;
            const float *pb1 = pb0 + 32 * AS_LD;
#pragma unroll
            for (int g8 = 0; g8 < 4; ++g8) {
                const float4 a0 = *reinterpret_cast<const float4 *>(pa0 + g8 * 8);
                const float4 a1 = *reinterpret_cast<const float4 *>(pa1 + g8 * 8);
                const float4 b0 = *reinterpret_cast<const float4 *>(pb0 + g8 * 8);
                const float4 b1 = *reinterpret_cast<const float4 *>(pb1 + g8 * 8);
                const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
                const float bv0[4] = {b0.x, b0.y, b0.z, b0.w}, bv1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[m], bv0[m], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[m], bv1[m], acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[m], bv0[m], acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[m], bv1[m], acc[1][1], 0, 0, 0);
                }
            }
            if ((c & 7) == 7 || c + 1 == nchunks) {  // end of a 256-column segment: tot = (first) ? acc : tot + acc
                const bool first = c < 8;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            tot[a][b][e] = first ? acc[a][b][e] : tot[a][b][e] + acc[a][b][e];
                            acc[a][b][e] = 0.f;
                        }
            }
        }

        if (cg == 0) {
            // finish ||x||^2: (p0+p1)+(p2+p3) in the thread, then the 8 threads of the row
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float t = (ssq[m][0] + ssq[m][1]) + (ssq[m][2] + ssq[m][3]);
                t = t + __shfl_xor(t, 1);
                t = t + __shfl_xor(t, 2);
                t = t + __shfl_xor(t, 4);
                if (sq == 0) sXn[srow + 32 * m] = norm2_from_sumsq(t);
            }
        }
        __syncthreads();

        // epilogue: distances -> per-lane argmin over this wave's 64 centres, both row tiles
        // (lane coordinates re-derived from a laundered copy of the thread index: what the epilogue addresses with must not stay
        // live -- hoisted -- across the stage loop, where every register is taken)
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        const int l31e = tid_e & 31, he = (tid_e >> 5) & 1;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const float xn = sXn[rt * 32 + l31e];
            float bv = INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kl = (2 * wave + ct) * 32 + (e & 3) + 8 * (e >> 2) + 4 * he;
                    const int disc = sDisc[kl];
                    if (disc >= 0) {
                        const float t = dist_epilogue(tot[ct][rt][e], xn, sCn[kl], disc != 0, r);
                        lexmin(bv, bi, t, kbase + kl);
                    }
                }
            }
            const float ov = __shfl_xor(bv, 32);
            const int oi = __shfl_xor(bi, 32);
            lexmin(bv, bi, ov, oi);
            if (he == 0) {
                sMinV[wave][rt * 32 + l31e] = bv;
                sMinI[wave][rt * 32 + l31e] = bi;
            }
        }
        __syncthreads();
        if (tid_e < AS_ROWS) {
#pragma unroll
            for (int w = 0; w < 4; ++w) lexmin(gbv, gbi, sMinV[w][tid_e], sMinI[w][tid_e]);
        }
        // the next group's first __syncthreads orders these reads before sMin* is rewritten
    }

    if (tid < AS_ROWS) {  // wave 0
        const bool ok = row0 + tid < n;
        if (ok) {
            const int64_t dst = row_idx ? (int64_t)row_idx[row0 + tid] : row0 + tid;
            labels[dst] = (int64_t)gbi;
            if (minval_out) minval_out[dst] = gbv;
        }
        double s = ok ? (double)gbv : 0.0;
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (tid == 0) wg_sum[tile] = s;
    }
    __syncthreads();  // the next tile rewrites the epilogue scratch
  }
  // re-check pass = the last kernel of a filter sweep: the last workgroup to finish hands the sweep's counters to the
  // statistics copy and zeroes the control block for the next sweep (every workgroup has read the list length by then)
  // (row_cnt of a re-check pass is always the f32_count member of the sweep's AssignCtl)
  if (row_idx != nullptr && tid == 0) assign_ctl_finish(ctl_of_f32_count(row_cnt), gridDim.x);
}

// --------------------------------------------------------------------------- the filter (k_assign_f16_rw)
// HBM-bound calc_best: a reduced-precision MFMA FILTER followed by an exact fp32 re-check of the ambiguous rows.
// (Rounds 1-4: bf16 operands, kernel k_assign_bf16_rw.  Round 5: IEEE half -- see the ROUND 5 notes below; read "bf16" in the
// older remarks of this file as "the filter's 16-bit operand type".)
//   1. distances with centres and rows rounded to 16 bits (v_mfma_f32_32x32x16_f16, fp32 accumulate), the
//      same fused epilogue; per row the best (d1, k1) and the runner-up value d2 are tracked
//   2. |D~ - D| <= E_i for every centre (E_i: rigorous bound, see below), so d2 - d1 > 2 E_i proves that k1
//      is the argmin of the canonical fp32 distances -- first-index ties included, because the inequality
//      is strict; such rows are final
//   3. every other row is appended to a list and re-labelled by the exact kernel (k_assign_f32)
// The result is therefore bit-identical to the exact kernel for ANY input; only the speed depends on how
// well separated the clusters are.
// Error bound (per row i, any centre k):  dot~ uses c~ = c(1+a), x~ = x(1+b), |a|,|b| <= 2^-8 (RNE to bf16's 8
// significant bits: unit roundoff 2^-p = 2^-8; 1 + 2^-8 rounds to 1), products exact in fp32, accumulation error
// <= d 2^-24 sum|c~x~|; the canonical dot has error <= d 2^-24 sum|cx|; sum|cx| <= ||c|| ||x||.
// Hence |dot~ - dot| <= ||c|| ||x|| (2^-7 (1 + 2^-9) + 2.02 d 2^-24)
// and, through -2 dot + ||x||^2 + ||c||^2 (three fp32 roundings of magnitude <= (||x||+||c||)^2):
//   E_i = 2.02 (2^-7 1.002 + 2.02 d 2^-24) cmax ||x_i|| + 2^-17 (||x_i|| + cmax)^2,   cmax = max_k ||c_k||
// (ROUND 5 CORRECTION: rounds 1-4 charged 2^-9 per operand -- half of bf16's unit roundoff -- so the leading term of E
// was half of what the proof needs.  No test or stress run ever produced a wrong label: a real dot's rounding errors add
// up like a random walk, ~sqrt(d) below this worst case.  But the claim is a PROOF, and the window is now the proven one.)
// ROUND 5, HALF-PRECISION OPERANDS.  With the proven 2^-8 the bf16 window doubled (62 % undecided rows became 84 % on the hard
// data of bench.py's variants, 2.0 -> 3.7 ms), so the operands moved to IEEE half: 11 significant bits, unit roundoff 2^-11, the
// same MFMA rate and bytes:
//   |dot~ - dot| <= ||c'|| ||x|| (2^-10 (1 + 2^-12) + 2.02 d 2^-24) + the underflow terms of k_centers_scale (eA, eB)
// -- a window 8 x narrower than bf16's proven one (4 x narrower than what rounds 1-4 ran with): 27 % undecided, 1.40 ms.  On
// well separated data the kernel is 2-5 % SLOWER than with bf16 (0.81-0.84 -> 0.85-0.86 ms per 1M x 1024 on the same box, also
// with the legacy RTZ pack conversion: not the conversion).  Half's narrow exponent range is handled by exact power-of-two
// scales (CentersAux) and by filter_bound() refusing rows that could overflow.
// CENTRED CENTRES.  With c = c' + mu for ANY common vector mu, -2 x.c = -2 x.c' - 2 x.mu and the last term is the
// same for every centre of a row: it moves d1 and d2 alike.  The filter therefore multiplies by c' = fl(c - mu)
// (mu = the mean centre) and its bf16 error scales with cmax' = max_k ||c'_k|| -- the SPREAD of the centres -- instead
// of their norm; embeddings with a large common component (post-ReLU features) would otherwise send almost every
// row to the exact re-check.  Only the accumulation error of the canonical dot keeps the raw norm:
//   E_i = 2.02 [ (u2 1.002 + 1.01 d 2^-24) cmax' + (1.01 d 2^-24 + 2^-24) cmax ] ||x_i|| + 2^-17 (||x_i|| + cmax)^2
//   (u2 = both operands' unit roundoffs: 2^-7 for bf16, 2^-10 for half; + eA ||x_i|| + eB with half)
// (2^-24 ||x|| cmax' covers the rounding of c - mu; mu = 0 gives back the formula above).  A row constant does not
// survive the under-use division by r, so the centres are only centred when no centre is under-used.
// The second term also covers what the filter's epilogue does differently from the exact one: it multiplies by
// fl(1/r) where the exact path divides by r (< 2 ulp), and it overwrites the 5 low mantissa bits of a distance
// with the centre's position in the lane (< 2^-18 relative) -- together < 2^-17 (||x_i|| + cmax)^2 with room to
// spare.  (The under-use scaling by 1/r < 1 only shrinks both sides.)
// Operand type of the filter's MFMA.  Rounds 1-4: bf16 (8 significant bits, unit roundoff 2^-8).  Round 5: IEEE half (11 bits,
// 2^-11) at the same MFMA rate and the same bytes -- the acceptance window is 8 x narrower for the same data.  Half has a
// narrow exponent range, so both operands are scaled by exact powers of two first (below); bf16 needed no such care.
typedef _Float16 fl16;
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));  // (8 operand elements of one lane: the historical name stays)

constexpr int FB_ROWS = 128;  // rows per workgroup (4 MFMA row tiles)
constexpr int FILTER_NW_DEFAULT = 4, FILTER_SCHED_DEFAULT = 0;  // K <= 256 defaults of k_assign_f16_rw (see acav_kmeans_assign)

struct CentersAux {
    unsigned cmax_bits;   // bits of max_k ||c_k||^2 (non-negative floats order like unsigned)
    unsigned cmaxc_bits;  // bits of (an upper bound of) max_k ||c'_k||^2 of the copy the filter multiplies by
    unsigned any_disc;    // some centre is under-used (distance / r): no centring
    unsigned amax_bits;   // bits of max_kj |c_kj| of the RAW centres: sets the scale of the rows (rows look like centres)
    unsigned amaxc_bits;  // bits of max_kj |c'_kj| of what the filter multiplies by (centred or raw): sets the centres' scale
    float sx, sc;         // exact powers of two: the filter computes with fl16(sx x) and fl16(sc c') (k_centers_scale)
    float inv_ss;         // 1 / (sx sc), exact: the accumulated dot is scaled back in the epilogue's fma
    float eA, eB;         // underflow terms of the acceptance bound: E += eA ||x|| + eB (k_centers_scale)
};

// The acceptance bound of the filter (header of k_assign_f16_rw; derivation at CentersAux / k_centers_scale):
//   E = (e1c cmax' + e1r cmax + eA) ||x|| + eB + e2 (||x|| + cmax)^2
// NaN when an element of the scaled row could leave half's range (max_j |x_j| <= ||x||): RNE would turn it into +-inf and a
// single +inf product can make ONE distance -inf and the gap to the runner-up +inf -- "decided", wrongly.  Every comparison
// against a NaN bound fails, so such a row is undecided and takes the exact path.
__device__ __forceinline__ float filter_bound(const CentersAux *__restrict__ aux, float xnorm, float e1c, float e1r, float e2coef)
{
    const float cmax = __builtin_sqrtf(__uint_as_float(aux->cmax_bits));
    const float cmaxc = __builtin_sqrtf(__uint_as_float(aux->cmaxc_bits));
    const float s = xnorm + cmax;
    const float E = (e1c * cmaxc + e1r * cmax + aux->eA) * xnorm + aux->eB + e2coef * s * s;
    return xnorm * aux->sx < 60000.0f ? E : __builtin_nanf("");
}

// mu[j] = mean over the centres of column j (any vector would do, see the bound): 32 columns x 8 centre lanes per block
// Block 0 also scans the per-centre scalars: max ||c_k||^2 and whether any centre is under-used.
__global__ __launch_bounds__(256) void k_centers_mu(const float *__restrict__ c, const float *__restrict__ cn,
                                                    const float *__restrict__ counts, int K, int d, float thr,
                                                    float *__restrict__ mu, CentersAux *__restrict__ aux)
{
    __shared__ float sp[8][32];
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < K; k += blockDim.x) {
            atomicMax(&aux->cmax_bits, __float_as_uint(cn[k]));
            if (counts[k] < thr) atomicOr(&aux->any_disc, 1u);
        }
    const int cj = threadIdx.x & 31, ky = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + cj;
    float s = 0.f, am = 0.f;
    if (j < d)
        for (int k = ky; k < K; k += 8) {
            const float v = c[(size_t)k * d + j];
            s = s + v;
            am = fmaxf(am, fabsf(v));
        }
    if (am > 0.f) atomicMax(&aux->amax_bits, __float_as_uint(am));  // (a NaN centre element: fmaxf drops it; the bound's norms carry it)
    sp[ky][cj] = s;
    __syncthreads();
    if (ky == 0 && j < d) {
        float t = sp[0][cj];
#pragma unroll
        for (int q = 1; q < 8; ++q) t = t + sp[q][cj];
        mu[j] = t / (float)K;
    }
}

// max_kj |c'_kj| of what the filter multiplies by (c - mu when no centre is under-used, else c): one block per centre
__global__ __launch_bounds__(256) void k_centers_amaxc(const float *__restrict__ c, const float *__restrict__ mu, int d, int K,
                                                       CentersAux *__restrict__ aux)
{
    const bool centred = aux->any_disc == 0u;
    const size_t base = (size_t)blockIdx.x * d;
    float am = 0.f;
    for (int j = threadIdx.x; j < d; j += blockDim.x) am = fmaxf(am, fabsf(centred ? c[base + j] - mu[j] : c[base + j]));
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) am = fmaxf(am, __shfl_xor(am, dlt));
    if ((threadIdx.x & 63) == 0 && am > 0.f) atomicMax(&aux->amaxc_bits, __float_as_uint(am));
}

// The scales of the half-precision filter and the underflow terms of its bound (one thread).
//   sc = 2^(9 - floor(log2 amaxc)): the largest element of what the filter multiplies by lands in [2^9, 2^10) (known exactly: no
//        headroom needed);
//   sx = 1 while the largest RAW centre element (rows look like centres) sits in [2^-5, 2^6], else 2^(2 - floor(log2 amax)) -> amax sx
//        in [4, 8): a row element may be ~10^3..10^4 x larger than the largest centre element before sx x leaves half's range
//        (then filter_bound() refuses the row: noise-dominated rows over tiny centres are routine, tests/test_gpu_kmeans.py).
// With x~ = fl16(sx x), c~ = fl16(sc c'):  an element inside half's NORMAL range [2^-14, 65504] is rounded with relative error
// <= 2^-11; below 2^-14 the result is a subnormal with ABSOLUTE error <= usub = 2^-25 -- on hardware that keeps half subnormals in
// the conversions and in the MFMA's operands, which gfx950 does and half_subnormals_kept() verifies once per device with the
// very instructions of this file (tools/exp/f16_denorm_probe.hip was the first look); hardware that flushed them would be charged
// usub = 2^-14 (ACAV_FILTER_SUBNORMAL=0 forces that).  Hence, divided by sx sc,
//   |dot~ - dot| <= (2^-10 + 2^-22) ||c'|| ||x||                       both roundings (e1c on the host side, with the fp32 accumulation)
//                 + usub 1.001 (||c'||_1 / sx  +  ||x||_1 / sc)         one operand under the normal range, the other rounded
//                 + d usub^2 / (sx sc)                                  both
// and with ||v||_1 <= sqrt(d) ||v||, doubled for the -2 of the distance (2.02 as everywhere in the bound):
//   eA = 2.02 * usub * 1.001 * sqrt(d) / sc                                   (x ||x||)
//   eB = 2.02 * (usub * 1.001 * sqrt(d) * cmax' / sx  +  d usub^2 / (sx sc))
// Against the rounding term 2^-10 cmax' ||x|| the row term of eB is 2^-15 sqrt(d) / (sx ||x||): under 1 % for d <= 2304 wherever
// sx ||x|| >= 0.15, which both branches of sx guarantee for rows that look like the centres (||x|| >~ amax).  (Round 5's first
// version charged the flush value 2^-14 unconditionally: with sx = 1 that was 4 / ||x|| of the rounding term at d = 1024 -- 40 % on
// the candidate re-check test's data, 4 x the whole window for unit-norm rows.)
__global__ void k_centers_scale(CentersAux *__restrict__ aux, int d, float usub)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    auto pow2_for = [](unsigned bits, int target) -> float {
        const float a = __uint_as_float(bits);
        if (!(a > 0.f) || !(a < 3.0e38f)) return 1.0f;
        int e;
        (void)frexpf(a, &e);          // a = m 2^e, m in [0.5, 1): floor(log2 a) = e - 1
        int p = target - (e - 1);
        p = p > 100 ? 100 : (p < -100 ? -100 : p);
        return ldexpf(1.0f, p);
    };
    // rows: no scaling (sx = 1: the multiply costs the filter 4-6 %) while the largest RAW centre element sits in [2^-5, 2^6]
    const float amax = __uint_as_float(aux->amax_bits);
    const float sx = (amax >= 0.03125f && amax <= 64.0f) ? 1.0f : pow2_for(aux->amax_bits, 2);
    const float sc = pow2_for(aux->amaxc_bits, 9);
    const float cmaxc = __builtin_sqrtf(__uint_as_float(aux->cmaxc_bits));
    const float sd = __builtin_sqrtf((float)d) * 1.0001f;
    aux->sx = sx, aux->sc = sc;
    aux->inv_ss = (1.0f / sx) * (1.0f / sc);  // powers of two: exact
    aux->eA = 2.02f * usub * 1.002f * sd / sc;
    aux->eB = 2.02f * (usub * 1.002f * sd * cmaxc / sx + (float)d * usub * usub * 1.01f * (1.0f / sx) * (1.0f / sc));
}

// one block per centre slot: the half-precision copy of sc c_k (or of sc (c_k - mu)) and the largest squared norm of what was
// rounded (in UNSCALED units: cmax').
// LAYOUT (round 4): STAGE-MAJOR -- out[(j / 32) * Kp + k][j % 32], Kp = K rounded up to whole groups of 256 (the slots past K
// repeat the last centre; the filter gives them an unbeatable norm).  The filter's centre stage (32 columns of 256 centres) is
// then ONE contiguous 16 KB block: a DMA instruction covers 1 KB of whole 128-byte lines instead of 16 half lines 2 d bytes
// apart -- half the L2 requests of the centre stream (the path runs at ~100 G L2 requests/s whatever the bytes).
// Two passes over the slot's row: the norm first (k_centers_scale needs the max before anything is scaled), the copy in
// k_centers_f16 below.
__global__ __launch_bounds__(256) void k_centers_norm(const float *__restrict__ c, const float *__restrict__ mu, int d, int K,
                                                      CentersAux *__restrict__ aux)
{
    __shared__ float sred[4];
    const bool centred = aux->any_disc == 0u;
    const size_t base = (size_t)blockIdx.x * d;
    float ss = 0.f;
    for (int j = threadIdx.x; j < d; j += blockDim.x) {
        const float v = centred ? c[base + j] - mu[j] : c[base + j];
        ss = __builtin_fmaf(v, v, ss);
    }
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) ss = ss + __shfl_xor(ss, dlt);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = ((sred[0] + sred[1]) + (sred[2] + sred[3])) * (1.0f + 2.0f * (float)d * 5.9604645e-8f);  // >= exact
        atomicMax(&aux->cmaxc_bits, __float_as_uint(tot));
    }
}

__global__ __launch_bounds__(256) void k_centers_f16(const float *__restrict__ c, const float *__restrict__ mu, int d, int K,
                                                     int Kp, fl16 *__restrict__ out, const CentersAux *__restrict__ aux)
{
    const bool centred = aux->any_disc == 0u;
    const float sc = aux->sc;
    const int k = (int)blockIdx.x;
    const size_t base = (size_t)(k < K ? k : K - 1) * d;
    for (int j = threadIdx.x; j < d; j += blockDim.x) {
        const float v = centred ? c[base + j] - mu[j] : c[base + j];
        out[((size_t)(j >> 5) * Kp + k) * 32 + (j & 31)] = (fl16)(v * sc);  // v * sc exact (power of two), then RNE to half
    }
}

struct Top2 {
    float d1;
    int k1;
    float d2;
};
__device__ __forceinline__ Top2 top2_merge(Top2 a, Top2 b)
{
    const bool b_wins = b.d1 < a.d1 || (b.d1 == a.d1 && b.k1 < a.k1);
    Top2 m;
    m.d1 = b_wins ? b.d1 : a.d1;
    m.k1 = b_wins ? b.k1 : a.k1;
    m.d2 = fminf(b_wins ? b.d2 : a.d2, b_wins ? a.d1 : b.d1);
    return m;
}

// One step of the compare-free top-2 scan: (s1, s2) <- the two smallest of {s1, s2, v}, s1 <= s2 on entry and exit.
// new s2 = median(s1, s2, v), new s1 = min(s1, v): v_med3_f32 + v_min_f32.  Written as asm because fminf / fmaxf on a
// value that comes out of integer ops (the position tag) make the compiler put a NaN-canonicalising `v_max x, x, x` in
// front of every use -- the round-3 scan spent 5.5 VALU per distance where 3.5 suffice.  A NaN distance: v_min keeps s1,
// v_med3 returns min3 = s1 -> s2 collapses onto s1 and the row goes to the exact re-check, as before.
__device__ __forceinline__ void top2_push(float &s1, float &s2, float v)
{
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(s2) : "v"(s1), "v"(s2), "v"(v));
    asm("v_min_f32 %0, %1, %2" : "=v"(s1) : "v"(s1), "v"(v));
}
typedef float f32x2 __attribute__((ext_vector_type(2)));

// The filter kernel.  Built so the HBM stream never drains: the raw fp32 rows and the bf16 centres both arrive by
// LDS-DMA (global_load_lds_dwordx4) into rings -- no staging VGPRs -- and the fp32 -> bf16 rounding happens when a
// wave reads its B fragment.
//   * 4 "fat" waves (256 threads, 256 VGPRs), 128 rows x 256 centres per workgroup, TWO workgroups per CU (80 KB of
//     LDS each): one workgroup's epilogue / ring fill hides under the other's main loop.  Wave w owns centres
//     64 w .. 64 w + 63 against all 128 rows (2 x 4 accumulator tiles of 32x32).
//   * 32 columns per stage (2 MFMA k-steps); rows ring of 3 stages, centres ring of 2.  Every wave issues a quarter
//     of both: per stage 4 centre DMAs (stage c+1) THEN 4 row DMAs (stage c+2), so that the in-order
//     `s_waitcnt vmcnt(4)` at the top of a stage retires rows c and centres c and leaves rows c+1 in flight.
//   * one raw s_barrier per stage: after it every wave may read stage c, and the slots of stage c-1 are free for
//     the next DMA (a __syncthreads() would drain the DMA queue: vmcnt(0)).
//   * the DMAs are issued from inline asm: hipcc cannot prove that a ds_read does not alias an in-flight builtin
//     LDS-DMA and would put `s_waitcnt vmcnt(0)` in front of the first fragment read of every stage (measured:
//     the ring then never holds more than one stage).
//   * rows: 128-B row chunks, 16-B slots XOR-swizzled by ((row >> 1) & 7); centres: 64-B row chunks, slots
//     XOR-swizzled by ((row >> 2) & 3) -- both applied to the DMA source address, the LDS image stays
//     lane-linear; with ds_read_b128's 16-lane groups {0-3,12-15,20-27}/{4-11,16-19,28-31} both fragment reads
//     are conflict-free.
//   * canonical ||x||^2: wave w accumulates the 16 classes per lane of row tile w (both k-steps).
//   * the epilogue scratch aliases the row ring; it multiplies by 1/r where the exact path divides by r (inside
//     the e2 term of the acceptance bound).
constexpr int FD_BK = 32;
constexpr int FD_DX = 3;  // row ring depth (2 stages = 32 KB in flight per workgroup, two workgroups per CU)
constexpr int FD_DC = 2;  // centre ring depth (1 stage in flight: an L2 round trip is shorter than a stage)
constexpr int FD_SLOT = 16384;  // bytes per ring slot: 128 rows x 32 fp32 == 256 centres x 32 bf16
constexpr int FD_SMEM = (FD_DX + FD_DC) * FD_SLOT;  // 80 KB: two workgroups fill the CU's 160 KB

// LDS byte address of a __shared__ pointer (wave-uniform) and a 16-byte-per-lane LDS-DMA issued from inline asm:
// lane l's 16 bytes at gbase + voff(l) land at lds + 16 l.  The compiler does not see the pending LDS write, so the
// caller owns the ordering: counted `s_waitcnt vmcnt(N)` + barrier before any read of the destination.
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return __builtin_amdgcn_readfirstlane(
        (unsigned)(__SIZE_TYPE__)(const __attribute__((address_space(3))) void *)(p));
}
__device__ __forceinline__ void dma16_asm(const void *gbase_uniform, unsigned voff, unsigned lds)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(gbase_uniform), "s"(lds)
                 : "memory", "m0");
}
// the same with the non-temporal policy: for data that is read exactly once (the feature rows)
__device__ __forceinline__ void dma16_asm_nt(const void *gbase_uniform, unsigned voff, unsigned lds)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(voff), "s"(gbase_uniform), "s"(lds)
                 : "memory", "m0");
}

// the same with a full 64-bit address per lane (rows picked through a list: no common base within 4 GB)
__device__ __forceinline__ void dma16_asm_v64(const char *addr, unsigned lds)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(addr), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ void dma16_asm_v64_nt(const char *addr, unsigned lds)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(addr), "s"(lds) : "memory", "m0");
}

// eight row elements -> half (4 v_cvt_pk_f16_f32, RNE), scaled by the exact power of two sx first when sx != 1 (4 v_pk_mul_f32
// more: measured +4.5 % on the K = 256 filter, +6 % at K = 1024 -- k_centers_scale therefore keeps sx = 1 whenever the data's
// own scale sits well inside half's range, and the branch is wave-uniform)
__device__ __forceinline__ bf16x8 cvt_bf16x8(float4 lo, float4 hi, float sx, bool scaled)
{
    typedef float f32x2s __attribute__((ext_vector_type(2)));
    f32x2s p0 = {lo.x, lo.y}, p1 = {lo.z, lo.w}, p2 = {hi.x, hi.y}, p3 = {hi.z, hi.w};
    if (scaled) {
        const f32x2s s2 = {sx, sx};
        p0 = p0 * s2, p1 = p1 * s2, p2 = p2 * s2, p3 = p3 * s2;
    }
    // Four v_cvt_pk_f16_f32 (RNE).  Each float2 -> half2 pair is bit-cast to a dword AS A WHOLE: hipcc 7.2 lowers the vector conversion
    // correctly while the pair is consumed whole, but EXTRACTING element 1 of its result yields element 0 (seen twice:
    // tools/exp/f16_denorm_probe.hip's `v_cvt_pk_f16_f32 v2, s8, s8`, and the first version of k_half_subnormal_check storing
    // h[1] = h[0]).  (An inline-asm v_cvt_pk_f16_f32 instead was WRONG in the XS instantiations with 8 waves: a third of the labels
    // of a 3 000-row test, nondeterministically -- the hazard recogniser does not look into inline asm next to the packed multiplies;
    // found by tools/stress_parity.py once it varied the data's scale.  NOTES_r05 section 14.)
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
    u32x4s w;
    w.x = __builtin_bit_cast(unsigned, __builtin_convertvector(p0, h2));
    w.y = __builtin_bit_cast(unsigned, __builtin_convertvector(p1, h2));
    w.z = __builtin_bit_cast(unsigned, __builtin_convertvector(p2, h2));
    w.w = __builtin_bit_cast(unsigned, __builtin_convertvector(p3, h2));
    return __builtin_bit_cast(bf16x8, w);
}

// Does this device keep half-precision subnormals where the filter's bound assumes it (k_centers_scale: usub)?  One wave, the
// instructions of the product path: the pack conversion of cvt_bf16x8 and the scalar conversion of k_centers_f16 on multiples of a
// QUARTER of the subnormal spacing 2^-24 (every rounding case: exact, below / above half way, ties to even; the larger values
// cross into the normal range), then v_mfma_f32_32x32x16_f16 with those subnormals as the A operand against 1024.0, and as the B
// operand.  The host compares every half and every accumulator with the exact values.
__global__ __launch_bounds__(64) void k_half_subnormal_check(unsigned short *__restrict__ halves, unsigned short *__restrict__ halves_scalar,
                                                             float *__restrict__ acc_a, float *__restrict__ acc_b)
{
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
    const int l = threadIdx.x;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (float)((l * 8 + q) * 33) * 1.4901161e-8f;  // (33 idx / 4) 2^-24: exact in fp32
    const bf16x8 h = cvt_bf16x8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), 1.0f, false);
    const u32x4s hw = __builtin_bit_cast(u32x4s, h);
    bf16x8 big;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        halves[l * 8 + q] = (unsigned short)(hw[q >> 1] >> (16 * (q & 1)));
        halves_scalar[l * 8 + q] = __builtin_bit_cast(unsigned short, (fl16)v[q]);  // v_cvt_f16_f32: k_centers_f16's conversion
        big[q] = (fl16)1024.0f;
    }
    f32x16 a = {0}, b = {0};
    a = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, big, a, 0, 0, 0);  // out[i][j] = 1024 sum_k h(row i)[k]
    b = __builtin_amdgcn_mfma_f32_32x32x16_f16(big, h, b, 0, 0, 0);  // out[i][j] = 1024 sum_k h(column j)[k]
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_a[l * 16 + r] = a[r], acc_b[l * 16 + r] = b[r];
}

// -> 2^-25 when the device passed the check above (cached per device), 2^-14 (flush to zero charged) otherwise
static int half_underflow_unit(int device, hipStream_t st, float *usub)
{
    static std::mutex mu;
    static int state[64] = {0};  // 0 unknown, 1 kept, 2 not kept
    *usub = 6.1035156e-5f;       // 2^-14
    if (const char *v = getenv("ACAV_FILTER_SUBNORMAL"))
        if (v[0] == '0') return ACAV_OK;
    std::lock_guard<std::mutex> lock(mu);
    const int slot = device >= 0 && device < 64 ? device : 63;
    if (state[slot] == 0) {
        const bool say = getenv("ACAV_FILTER_SUBNORMAL_DEBUG") != nullptr;
        acav::DevBuf buf;
        ACAV_TRY(buf.ensure(2 * 512 * sizeof(unsigned short) + 2 * 1024 * sizeof(float)));
        unsigned short *hv = buf.as<unsigned short>(), *hs = hv + 512;
        float *aa = reinterpret_cast<float *>(hs + 512), *ab = aa + 1024;
        hipLaunchKernelGGL(k_half_subnormal_check, dim3(1), dim3(64), 0, st, hv, hs, aa, ab);
        ACAV_HIP_TRY(hipGetLastError());
        std::vector<unsigned short> h(1024);
        std::vector<float> acc(2048);
        ACAV_HIP_TRY(hipMemcpyAsync(h.data(), hv, 1024 * sizeof(unsigned short), hipMemcpyDeviceToHost, st));
        ACAV_HIP_TRY(hipMemcpyAsync(acc.data(), aa, 2048 * sizeof(float), hipMemcpyDeviceToHost, st));
        ACAV_HIP_TRY(hipStreamSynchronize(st));
        auto dec = [](unsigned short b) {  // a non-negative finite half
            const int e = (b >> 10) & 31, m = b & 1023;
            return e ? ldexp(1024.0 + m, e - 25) : ldexp((double)m, -24);
        };
        bool ok = true;
        for (int i = 0; i < 512 && ok; ++i) {
            // value = (33 i / 4) units of 2^-24.  RNE to a count of units of the result's spacing: one unit while the count fits
            // 11 bits (subnormals AND the first normal binade -- the encoding simply runs on), doubling beyond
            const unsigned num = 33u * (unsigned)i;  // quarter units
            unsigned sh = 2;                          // log2 (quarter units per unit of the result's spacing)
            while ((num >> sh) > 2047u) ++sh;
            unsigned q = num >> sh;
            const unsigned rem = num & ((1u << sh) - 1u), halfway = 1u << (sh - 1);
            if (rem > halfway || (rem == halfway && (q & 1u))) ++q;
            const double want = ldexp((double)q, (int)sh - 2 - 24);
            ok = dec(h[i]) == want && dec(h[512 + i]) == want && !(h[i] & 0x8000) && !(h[512 + i] & 0x8000);
            if (!ok && say) fprintf(stderr, "half check: %u/4 units of 2^-24 -> pack 0x%04x scalar 0x%04x, want %g\n", num, h[i], h[512 + i], want);
        }
        for (int l = 0; l < 64 && ok; ++l)
            for (int r = 0; r < 16 && ok; ++r) {
                const int row = 8 * (r / 4) + 4 * (l / 32) + (r % 4), col = l % 32;  // accumulator layout of the 32x32 MFMA
                double sa = 0, sb = 0;  // operand fragments: lanes i and i + 32 hold k = 0..7 and 8..15 of row / column i
                for (int q = 0; q < 8; ++q) {
                    sa += dec(h[row * 8 + q]) + dec(h[(row + 32) * 8 + q]);
                    sb += dec(h[col * 8 + q]) + dec(h[(col + 32) * 8 + q]);
                }
                // 16 exact products of <= 12-bit counts, at most two binades apart: the sums are exact in fp32 in any order
                ok = (double)acc[l * 16 + r] == 1024.0 * sa && (double)acc[1024 + l * 16 + r] == 1024.0 * sb;
                if (!ok && say) fprintf(stderr, "half check: mfma lane %d reg %d: %g / %g, want %g / %g\n", l, r, acc[l * 16 + r], acc[1024 + l * 16 + r], 1024.0 * sa, 1024.0 * sb);
            }
        if (say) fprintf(stderr, "half check: device %d %s half subnormals\n", device, ok ? "keeps" : "does NOT keep");
        state[slot] = ok ? 1 : 2;
    }
    if (state[slot] == 1) *usub = 2.9802322e-8f;  // 2^-25
    return ACAV_OK;
}

#ifdef ACAV_RW_PROF  // tools/exp/assign_bench.hip only: per-stage phase cycles of k_assign_f16_rw (s_memtime)
__device__ unsigned long long g_rw_prof[16];
#define RW_T(v) const long long v = clock64()
#else
#define RW_T(v)
#endif

// k_assign_f16_rw (k_assign_bf16_rw until round 5) -- "row waves": wave w owns ROWS 32 w .. 32 w + 31
// (round 1's layout -- wave = centre quarter, every wave re-converting all rows -- was kept as k_assign_bf16 for A/B runs until round 5)
// of the workgroup's 128 against ALL 256 centres of the group (8 accumulator tiles of 32x32).
//   * a wave reads and converts only its own rows' fp32 fragments: 8 v_cvt_pk_bf16_f32 per 32-column stage instead of
//     32 (in the centre-quarter layout every wave re-read and re-converted all 128 rows); the bf16 centre fragments,
//     which need no conversion, are the ones every wave reads
//   * a lane ends up with all 128 centres of its half for ITS row: the top-2 scan completes in the lane, the two
//     halves meet with one shfl_xor 32, and the running top-2 over centre groups stays in registers -- no cross-wave
//     merge through LDS, no per-group scratch for it
//   * the position tag needs 7 bits (128 distances per lane): 128 ulp = 2^-16 of the tagged distance.  That is charged
//     to the two distances the acceptance test compares (|d1| + |d2|) instead of to (||x|| + cmax)^2; what is left in the
//     e2 term are the 3 + 3 fp32 roundings of the two epilogues (< 6 x 2^-24 (||x|| + cmax)^2): e2 = 2^-20 here
// Rings, swizzles, DMA shares, counted waits and the one raw barrier per stage are those of round 1's kernel.
// Ablations of this kernel on one MI355X (1M x 1024, K = 256; wrong labels, timing only): full 0.93 ms; without the
// centre DMA 0.71 ms; DMA only (no fragment reads, no MFMA) 0.78 ms -- the bf16 centre stage (16 KB from L2 per 16 KB of
// rows from HBM) is the largest single cost.  A persistent 256-row workgroup (eight row waves sharing one centre stage,
// DMA cursors running across tile boundaries, one workgroup per CU) halves that traffic and was SLOWER (0.94 vs 0.90 ms):
// eight waves in lockstep on one barrier lose more than the centre traffic costs; two independent 128-row workgroups
// per CU interleave their phases.  (round 2, measured and rejected)
// Same-box ablations of the compute side (full 0.87-0.93 ms): half of the centre fragment reads 0.82, one centre tile's
// fragments only 0.81, half of the MFMAs 0.80, both halved 0.80 -- neither the LDS reads nor the matrix pipe is the floor:
// the two DMA streams are (rows from HBM + as many bytes of centres from L2: 8.2 GB through the LDS-DMA path per launch,
// DMA only 0.78 ms), against a chip that copies at 6.3 TB/s (0.65 ms for the rows alone, MI355X_MICROARCH.md).
// Halving the centre stream the other way -- a wave owning 64 rows (two accumulator sets = 256 registers, 512 with the
// fragments, 7 spills), 256 rows per workgroup against one centre stage, 128 KB of LDS, one wave per SIMD -- was correct
// and SLOWER too (0.97-0.98 vs 0.91-0.94 ms same box): with one wave per SIMD every barrier and LDS wait idles the SIMD.
//
// Template parameters (round 3):
//   NW   waves per workgroup = 32-row tiles per workgroup: 4 (128 rows, 80 KB of LDS, two workgroups per CU) or 8 (256 rows
//        against ONE centre stage, 128 KB, one workgroup per CU: half the centre bytes per row)
//   GS   "group split" for K > 256: one workgroup = one (row tile, group of 256 centres) PAIR instead of a loop over the
//        groups that re-streams the tile's rows from HBM once per group.  The pairs of a tile are consecutive in dispatch
//        order on ONE XCD (block b runs on XCD b % 8): the tile's rows come from HBM once and from that XCD's L2 for the
//        other groups.  Each pair leaves a (d1, k1, d2, ||x||^2) record per row; k_assign_merge folds the groups in
//        ascending order (the order of the loop) and applies the acceptance test.
//   DCR  centre ring depth.  2: a centre stage is issued one stage ahead (rows two).  With the rows L2-resident (GS) the
//        stage time is no longer set by HBM but by how long a DMA piece takes to land under load (~1 us, L2 hit or not)
//        over the stages of look-ahead: 3 gives the centre stream two stages like the rows (order per stage: centres
//        c+2 THEN rows c+2, counted wait leaves one stage of both in flight).
// ---- candidate-restricted exact re-check (round 4) ------------------------------------------------------------------
// A row the filter cannot decide does NOT need all K exact distances.  With E_i the row's error bound and c the tag slack
// (both as in the acceptance test), any centre k whose filter value v_k satisfies  v_k - d1 > 2 E_i + c (|d1| + |v_k|)  is
// strictly beaten by k1 in canonical fp32 arithmetic (the acceptance argument applied to the pair (k1, k)), so the canonical
// argmin -- and every centre that ties with it -- lies in  CAND_i = { k : that inequality FAILS }  (k1 is a member).  The
// filter's epilogue emits CAND_i for every undecided row (its 128 distances per lane are still in the accumulators) and
// k_assign_cand evaluates the canonical chain for those (row, centre) pairs only; lexmin over exact distances of a superset
// of the minimisers is the exact first-index argmin.  Rows with more than CAND_MAX candidates, or that do not fit the pair
// pool, go to the full exact sweep (k_assign_f32) as before -- never a wrong label, only a slower row.
// The emission test uses T_i >= every v that could fail the inequality (derivation at the use): a superset is always safe.
struct CandRow {        // one undecided row: its pairs are cpair[pair_base .. pair_base + cnt)
    int row;
    unsigned pair_base;
    unsigned cnt;
    float xn;           // canonical ||x||^2 of the row (the filter has it)
};
struct CandPair {
    unsigned slot;      // index into the CandRow array (slots and pair ranges are handed out by ONE 64-bit atomic, so both
    int k;              // are ordered alike: consecutive slots own consecutive pair ranges)
};
struct AssignCtl {                  // device-side control block of one assign sweep; the counters are zero between sweeps
    unsigned long long alloc;       // low 32 bits: candidate row slots handed out, high 32 bits: candidate pairs handed out
    unsigned f32_count;             // rows on the full exact re-check list
    unsigned ticket;                // workgroups of the sweep's last kernel that have finished (the last one resets the block)
    unsigned pool_over;             // rows that hold a slot but whose pairs did not fit the pool (they are on the f32 list too)
    unsigned und_count;             // rows the filter could not decide (the list the emission pass walks)
    unsigned last_und;
    unsigned long long last_alloc;  // copies for acav_kmeans_filter_stats, written by the sweep's last kernel
    unsigned last_f32, last_pool_over;
};
struct CandOut {  // where the emission writes: candidate rows / pairs, and the list of the full exact sweep
    CandRow *crow;
    CandPair *cpair;
    int *f32_list;
    float *und_T;  // K > 256: the candidate threshold T of every listed row (k_assign_merge knows the row's minimum over all groups)
    unsigned pair_cap;
};
// T: every filter value v > T satisfies v - d1 > W + c |v|, W = 2 E + c |d1|, c = 1.6e-5 (untagged v: its own tag is not charged,
// the slack stays).  With u = d1 + W:  v >= 0: v (1 - c) > u  <=  v > u / (1 - c) < u (1 + 2 c);  v < 0 (then u < 0): v (1 + c) > u
// <=  v > u (1 - 2 c).  Both are u + 2 c |u|; the last term covers the roundings of this evaluation itself (a few 2^-24 of
// |d1| + W) sixfold.  NaN / inf bounds give T = NaN: `!(v > T)` holds for every v -> everything is a candidate -> full exact sweep.
__device__ __forceinline__ float cand_threshold(float d1, float E)
{
#ifdef ACAV_DBG_HALF_CAND_WINDOW  // experiment builds: HALF the proven window -- tests/test_gpu_kmeans.py::test_candidate_threshold_... must FAIL
    const float W = 1.0f * E + 1.6e-5f * fabsf(d1);
#else
    const float W = 2.0f * E + 1.6e-5f * fabsf(d1);
#endif
    const float u = d1 + W;
    return u + 3.2e-5f * fabsf(u) + 1.6e-6f * (fabsf(d1) + W);
}
constexpr unsigned CAND_MAX = 16;  // candidates per row beyond which the row takes the full exact sweep
__device__ __forceinline__ AssignCtl *ctl_of_f32_count(const unsigned *f32_count)
{
    return reinterpret_cast<AssignCtl *>(reinterpret_cast<char *>(const_cast<unsigned *>(f32_count)) - offsetof(AssignCtl, f32_count));
}
__device__ __forceinline__ void assign_ctl_finish(AssignCtl *ctl, unsigned nblocks)
{
    // no fence: every workgroup read the list length long before its ticket, nobody reads the block after the reset inside
    // this kernel, and the kernel boundary publishes the stores (__threadfence() = buffer_wbl2 + buffer_inv: an L2 write-back
    // and an L1 invalidate under the other workgroups of the CU)
    if (atomicAdd(&ctl->ticket, 1u) == nblocks - 1u) {
        ctl->last_alloc = atomicAdd(&ctl->alloc, 0ull);  // read at the coherence point
        ctl->last_f32 = atomicAdd(&ctl->f32_count, 0u);
        ctl->last_pool_over = atomicAdd(&ctl->pool_over, 0u);
        ctl->last_und = atomicAdd(&ctl->und_count, 0u);
        ctl->und_count = 0u;
        ctl->alloc = 0ull;
        ctl->f32_count = 0u;
        ctl->pool_over = 0u;
        ctl->ticket = 0u;
    }
}

struct Top2Rec {
    float d1;
    int k1;
    float d2;
    float xn;
};
// XS (round 5): the rows are multiplied by the power of two aux->sx before their conversion to half (data whose own scale sits
// outside half's comfortable range: k_centers_scale); XS = false carries no trace of it -- the multiply and the registers behind
// it cost the K = 256 filter 4.5 % and pushed the product instantiation into scratch.
template <bool NT, int NW, bool GS, int DCR = FD_DC, int SCHED = 0, int EMIT = 0, bool XS = false>
// (second launch bound = waves per SIMD the register budget is cut for: 2 -> 256 registers, what two 128-row workgroups per CU need.
// The emission-pass instantiations (EMIT == 1: 64-bit row addresses through a list, the candidate emission) need ~40 registers more
// and took them from scratch; they only ever cover the undecided rows, so they get one wave per SIMD and the accumulation half of
// the file instead -- no instantiation the library launches uses scratch, and a queue never waits for a scratch allocation)
__global__ __launch_bounds__(NW * 64, EMIT == 1 ? 1 : 2) void k_assign_f16_rw(const float *__restrict__ x, int64_t n, int d,
                                                            const fl16 *__restrict__ cb, const float *__restrict__ cn,
                                                            const float *__restrict__ counts, int K, float thr, float r,
                                                            const CentersAux *__restrict__ aux, float e1c, float e1r, float e2coef,
                                                            int64_t *__restrict__ labels, int *__restrict__ recheck_list,
                                                            unsigned *__restrict__ recheck_count, AssignCtl *__restrict__ ctl,
                                                            Top2Rec *__restrict__ grec, CandOut out)
{
    // EMIT = 2 (the default for K <= 256): the filter's epilogue emits the candidate centres of every row it cannot decide, in
    // place -- its 128 distances per lane are still in the accumulators (header above).  The first version of this cost the
    // filter 6-18 % with NOT ONE row undecided (0.865 -> 0.94-1.035 ms per 1M x 1024) and was therefore split off into a second
    // pass (EMIT = 1); the cost was not the emission code but ONE register pair: the compiler computed &labels[row] in the
    // prologue, spilled it across the stage loop and re-loaded it from scratch (+ s_waitcnt vmcnt(0)) in every tile's epilogue.
    // With the row index laundered at its use (below) the kernel is scratch-free and as fast as the lean one (0.81-0.83 ms).
    // EMIT = 1 (ACAV_ASSIGN_EMIT=1): the EMISSION PASS over the rows a lean filter (EMIT = 0) listed: the rows are
    // x[recheck_list[0 .. *recheck_count)] (a fixed grid strides over their tiles), same main loop, same emission.
    constexpr bool LISTED = EMIT == 1;
    if (LISTED) n = (int64_t)*recheck_count;
    constexpr int XSLOT = NW * 4096;  // bytes per row-ring slot: NW x 32 rows x 32 fp32
    constexpr int CQ = 16 / NW;       // centre-stage DMA pieces (1 KB = 16 centres x 64 B) per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char fd_smem[];
    float *sXr = reinterpret_cast<float *>(fd_smem);                        // [FD_DX][NW * 32][32] fp32
    fl16 *sCb = reinterpret_cast<fl16 *>(fd_smem + FD_DX * XSLOT);          // [FD_DC][256][32] half
    float *sCn = reinterpret_cast<float *>(fd_smem);  // [256] epilogue scratch, aliases the (idle) row ring
    float *sSc = sCn + 256;                           // [256] 1, or 1/r for a discounted centre
    // emission pass only (EMIT = 1; 4.5 KB past the rings): every row's candidate list, collected over the centre groups
    unsigned *sCandN = reinterpret_cast<unsigned *>(fd_smem + FD_DX * XSLOT + DCR * FD_SLOT);  // [NW * 32]
    unsigned short *sCandK = reinterpret_cast<unsigned short *>(sCandN + NW * 32);              // [NW * 32][CAND_MAX]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave = row tile (32 rows)
    const int l31 = lane & 31, h = lane >> 5;
    const int nchunks = d / FD_BK;
    const int ngroups = (K + 255) / 256;
  for (int64_t tile_it = blockIdx.x;; tile_it += gridDim.x) {  // one trip, except in the emission pass
    if (LISTED && tile_it * (NW * 32) >= n) break;
    int64_t tile = tile_it;
    int cg0 = 0, cg1 = ngroups;
    if (GS) {
        const int j = blockIdx.x >> 3;  // sequence number on XCD (blockIdx.x & 7)
        tile = (int64_t)(j / ngroups) * 8 + (blockIdx.x & 7);
        cg0 = j % ngroups;
        cg1 = cg0 + 1;
        if (tile * (NW * 32) >= n) return;  // the grid is rounded up to whole rounds of 8 tiles
    }
    const int64_t row0 = tile * (NW * 32);
    const float inv_r = 1.0f / r;
    // Centred mode (no centre is under-used: the filter multiplies by c - mu): every distance of a row carries the same
    // constant ||x||^2 + M - 2 x.mu, M = any number.  It is left out of the compared values -- ||x||^2 is not added and
    // M = ||c_0||^2 is subtracted from every ||c_k||^2 (exact by Sterbenz when the norms are within a factor 2, else one
    // more rounding inside e2) -- so the position tag perturbs numbers of the size of the distance SPREAD, not of the
    // squared norms.  With an under-used centre the division by r does not commute with a row constant: full values.
    const bool centred = aux->any_disc == 0u;
    const float cn_shift = centred ? cn[0] : 0.0f;
    // half-precision operands: rows are multiplied by the exact power of two sx before the conversion, the centre copy holds
    // sc c'; the accumulated dot comes back through m2s = -2 / (sx sc) in the epilogue's fma (a power of two: exact)
    const float sxrow = XS ? aux->sx : 1.0f;

    unsigned voffx[4], voffc[CQ];
    const char *ax[4], *ax0[4];  // emission pass: the 64-bit source address of this lane's piece of row (wq * 4 + q) * 8 + (lane >> 3)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rr = (wq * 4 + q) * 8 + (lane >> 3);
        const int rc = row0 + rr < n ? rr : (int)(n - 1 - row0);  // ragged tail: re-read the last row
        voffx[q] = (unsigned)rc * (unsigned)d * 4u + (((lane & 7) ^ ((rr >> 1) & 7)) << 4);
        ax0[q] = nullptr;
        if (LISTED)
            ax0[q] = reinterpret_cast<const char *>(x + (size_t)recheck_list[row0 + rc] * d) + (((lane & 7) ^ ((rr >> 1) & 7)) << 4);
    }
    const unsigned xring = lds_addr(sXr) + wq * 4096, cring = lds_addr(sCb) + wq * (CQ * 1024);

    float ssq[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) ssq[e] = 0.f;
    Top2 run = {INFINITY, 0x7fffffff, INFINITY};
    float xn = 0.f;  // ||x||^2 of this lane's row
    if (LISTED && h == 0) sCandN[wq * 32 + l31] = 0u;  // (wave-private entries: ordered before the wave's own appends)
#ifdef ACAV_RW_PROF
    long long rwp[6] = {0, 0, 0, 0, 0, 0};
    const long long rw_tstart = clock64(), rw_wstart = wall_clock64();
#endif

    for (int cg = cg0; cg < cg1; ++cg) {
        const int kbase = cg * 256;
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
            const int rr = (wq * CQ + q) * 16 + (lane >> 2);
            voffc[q] = (unsigned)rr * 64u + (((lane & 3) ^ ((rr >> 2) & 3)) << 4);  // stage-major copy: 64 B per centre and stage
        }
        const char *gx = reinterpret_cast<const char *>(x + (size_t)row0 * d);
        const char *gc = reinterpret_cast<const char *>(cb + (size_t)kbase * 32);
        const size_t cstage = (size_t)ngroups * 256 * 64;  // bytes from one stage of the centre copy to the next
#pragma unroll
        for (int q = 0; q < 4; ++q) ax[q] = ax0[q];  // every centre group streams the rows from their first column
        int wx = 0, wc = 0;
        auto issue_x = [&]() {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#if defined(ACAV_ABL_NOXDMA_G123)  // timing-only bounds of a bf16-row hand-off between the pair workgroups of a tile:
                if (GS && cg > 0) continue;  // groups 1.. get their rows for free
#elif defined(ACAV_ABL_XHALF_G123)
                if (GS && cg > 0 && (q & 1)) continue;  // groups 1.. move half the row bytes (what bf16 rows would)
#endif
                if (LISTED) {
                    if (NT) dma16_asm_v64_nt(ax[q], xring + wx * XSLOT + q * 1024);
                    else dma16_asm_v64(ax[q], xring + wx * XSLOT + q * 1024);
                    ax[q] += FD_BK * 4;
                } else if (NT) dma16_asm_nt(gx, voffx[q], xring + wx * XSLOT + q * 1024);
                else dma16_asm(gx, voffx[q], xring + wx * XSLOT + q * 1024);
            }
            gx += FD_BK * 4;
            wx = wx + 1 == FD_DX ? 0 : wx + 1;
        };
        auto issue_c = [&]() {
#pragma unroll
            for (int q = 0; q < CQ; ++q) dma16_asm(gc, voffc[q], cring + wc * FD_SLOT + q * 1024);
            gc += cstage;
            wc = wc + 1 == DCR ? 0 : wc + 1;
        };

        f32x16 acc[8];
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;

        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave is done with the previous centre group's epilogue scratch
        if (DCR == 2) {
            issue_x();
            issue_c();
            if (nchunks > 1) issue_x();
        } else {
            issue_c();
            issue_x();
            if (nchunks > 1) {
                issue_c();
                issue_x();
            }
        }
        int rx = 0, rcs = 0;
        const int swz = (l31 >> 1) & 7, swa = (l31 >> 2) & 3;
        for (int c = 0; c < nchunks; ++c) {
            RW_T(t0);
#if defined(ACAV_ABL_NOXDMA)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
            // what stays in flight behind stage c: DCR 2 -- the 4 row pieces of stage c+1 (issued last, after the
            // centres of stage c); DCR 3 -- the CQ centre + 4 row pieces of stage c+1
#if defined(ACAV_ABL_NOXDMA_G123) || defined(ACAV_ABL_XHALF_G123)
            // (fewer row pieces in flight behind stage c for groups 1..: the counted wait follows)
            if (c + 1 >= nchunks) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (GS && cg > 0) {
#ifdef ACAV_ABL_NOXDMA_G123
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DCR == 2 ? 0 : CQ) : "memory");
#else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DCR == 2 ? 2 : CQ + 2) : "memory");
#endif
            } else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DCR == 2 ? 4 : CQ + 4) : "memory");
#else
            if (c + 1 < nchunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DCR == 2 ? 4 : CQ + 4) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            RW_T(t1);
            __builtin_amdgcn_s_barrier();
            RW_T(t2);
            const float *pxl = sXr + rx * (XSLOT / 4) + wq * 1024 + l31 * 32;  // this lane's row of the wave's own tile
            const fl16 *pcl = sCb + rcs * (256 * 32) + l31 * 32;              // row l31 of centre tile 0 (+ 1024 per tile)
            rx = rx + 1 == FD_DX ? 0 : rx + 1;
            rcs = rcs + 1 == DCR ? 0 : rcs + 1;
#define RW_LDB(ks, F0, F1)                                                                         \
    const float4 F0 = *reinterpret_cast<const float4 *>(pxl + (((4 * (ks) + 2 * h) ^ swz) << 2));  \
    const float4 F1 = *reinterpret_cast<const float4 *>(pxl + (((4 * (ks) + 2 * h + 1) ^ swz) << 2));
#define RW_LDA(ks, ct) (*reinterpret_cast<const bf16x8 *>(pcl + (ct) * 1024 + (((2 * (ks) + h) ^ swa) << 3)))
            // ACAV_ABL_*: timing-only ablations for tools/exp/assign_bench.hip (never defined in the product build)
#ifdef ACAV_ABL_NOAFRAG
#define RW_CT(ct) 0
#else
#define RW_CT(ct) (ct)
#endif
#ifdef ACAV_ABL_NOMFMA
#define RW_MFMA(a, b, cc) cc[0] += (float)(a)[0] + (float)(b)[0]
#else
#define RW_MFMA(a, b, cc) cc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, cc, 0, 0, 0)
#endif
            // SCHED: when a wave issues its CQ + 4 DMA pieces of the stage (an issue costs the wave 60-120 cycles while the
            // path is busy: tools/exp/assign_bench.hip -DACAV_RW_PROF).  0: all of them in one burst before the MFMAs.
            // 1 (512-thread workgroups): the second half of the waves shares its SIMDs with the first half and issues its
            // burst AFTER its MFMAs.  2: one piece after every second MFMA -- the DMA path sees a steady stream instead of
            // a burst per barrier, and while one wave of a SIMD sits in an issue the other one feeds the matrix pipe.
            const bool dma_first = SCHED == 0 || (SCHED == 1 && (NW == 4 || wq < 4));
            const bool dma_last = SCHED == 1 && !dma_first;
            constexpr int NP = CQ + 4, NP1 = NP / 2;
            auto piece = [&](int i) {
                if (i < CQ) {
#ifndef ACAV_ABL_NOCDMA
                    if (c + DCR - 1 < nchunks) dma16_asm(gc, voffc[i], cring + wc * FD_SLOT + i * 1024);
#endif
                    if (i == CQ - 1) {
                        gc += cstage;
                        wc = wc + 1 == DCR ? 0 : wc + 1;
                    }
                } else {
                    const int q = i - CQ;
#ifndef ACAV_ABL_NOXDMA
#if defined(ACAV_ABL_NOXDMA_G123)
                    if (c + 2 < nchunks && !(GS && cg > 0)) {
#elif defined(ACAV_ABL_XHALF_G123)
                    if (c + 2 < nchunks && !(GS && cg > 0 && (q & 1))) {
#else
                    if (c + 2 < nchunks) {
#endif
                        if (LISTED) {
                            if (NT) dma16_asm_v64_nt(ax[q], xring + wx * XSLOT + q * 1024);
                            else dma16_asm_v64(ax[q], xring + wx * XSLOT + q * 1024);
                            ax[q] += FD_BK * 4;
                        } else if (NT) dma16_asm_nt(gx, voffx[q], xring + wx * XSLOT + q * 1024);
                        else dma16_asm(gx, voffx[q], xring + wx * XSLOT + q * 1024);
                    }
#endif
                    if (q == 3) {
                        gx += FD_BK * 4;
                        wx = wx + 1 == FD_DX ? 0 : wx + 1;
                    }
                }
            };
            RW_LDB(0, p0, p1)
            bf16x8 a0[8], a1[8];
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) a0[ct] = RW_LDA(0, RW_CT(ct));
            RW_T(t3);
            if (dma_first) {  // into the slots stage c-1 just vacated
#pragma unroll
                for (int i = 0; i < NP; ++i) piece(i);
            }
            RW_T(t4);
            RW_LDB(1, q0, q1)
            const bf16x8 b0 = cvt_bf16x8(p0, p1, sxrow, XS);
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                RW_MFMA(a0[ct], b0, acc[ct]);
                a1[ct] = RW_LDA(1, RW_CT(ct));
                if (SCHED == 2 && (ct & 1) == 0 && ct / 2 < NP1) piece(ct / 2);
            }
            const bf16x8 b1 = cvt_bf16x8(q0, q1, sxrow, XS);
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
                RW_MFMA(a1[ct], b1, acc[ct]);
                if (SCHED == 2 && (ct & 1) == 0 && NP1 + ct / 2 < NP) piece(NP1 + ct / 2);
            }
            if (dma_last) {
#pragma unroll
                for (int i = 0; i < NP; ++i) piece(i);
            }
#undef RW_MFMA
#undef RW_CT
#ifdef ACAV_RW_PROF
            {
                RW_T(t5);
                rwp[0] += t1 - t0, rwp[1] += t2 - t1, rwp[2] += t3 - t2, rwp[3] += t4 - t3, rwp[4] += t5 - t4, rwp[5] += 1;
            }
#endif
            if (cg == cg0) {  // uniform: canonical ||x||^2 of the lane's row, classes 16 ks + 8 h + e
                ssq[0] = __builtin_fmaf(p0.x, p0.x, ssq[0]), ssq[1] = __builtin_fmaf(p0.y, p0.y, ssq[1]);
                ssq[2] = __builtin_fmaf(p0.z, p0.z, ssq[2]), ssq[3] = __builtin_fmaf(p0.w, p0.w, ssq[3]);
                ssq[4] = __builtin_fmaf(p1.x, p1.x, ssq[4]), ssq[5] = __builtin_fmaf(p1.y, p1.y, ssq[5]);
                ssq[6] = __builtin_fmaf(p1.z, p1.z, ssq[6]), ssq[7] = __builtin_fmaf(p1.w, p1.w, ssq[7]);
                ssq[8] = __builtin_fmaf(q0.x, q0.x, ssq[8]), ssq[9] = __builtin_fmaf(q0.y, q0.y, ssq[9]);
                ssq[10] = __builtin_fmaf(q0.z, q0.z, ssq[10]), ssq[11] = __builtin_fmaf(q0.w, q0.w, ssq[11]);
                ssq[12] = __builtin_fmaf(q1.x, q1.x, ssq[12]), ssq[13] = __builtin_fmaf(q1.y, q1.y, ssq[13]);
                ssq[14] = __builtin_fmaf(q1.z, q1.z, ssq[14]), ssq[15] = __builtin_fmaf(q1.w, q1.w, ssq[15]);
            }
#undef RW_LDB
#undef RW_LDA
        }
        // beyond K: a huge FINITE norm -- such a centre never wins and never becomes the runner-up.  (+inf would turn
        // into a NaN under the position tag, and fmaxf(s1, NaN) = s1 makes the runner-up collapse onto the minimum: every
        // lane that mixes real and padding centres would send its row to the re-check.)
        float my_cn = 3.0e38f, my_sc = 1.0f;
        if (kbase + tid < K && tid < 256) {
            my_cn = cn[kbase + tid] - cn_shift;
            my_sc = counts[kbase + tid] < thr ? inv_r : 1.0f;
        }
        __syncthreads();  // every wave has read its last fragments: the row ring becomes epilogue scratch
        if (NW == 4 || tid < 256) {
            sCn[tid] = my_cn;
            sSc[tid] = my_sc;
        }
        if (cg == cg0) {
            // canonical tree: (p0+p1)+(p2+p3) per group of 4 classes, then ((g0+g1)+(g2+g3)) + ((g4+g5)+(g6+g7)):
            // g0,g1 = (ks 0, h 0), g2,g3 = (ks 0, h 1), g4,g5 = (ks 1, h 0), g6,g7 = (ks 1, h 1); fp32 + commutes bitwise
            float ta = ((ssq[0] + ssq[1]) + (ssq[2] + ssq[3])) + ((ssq[4] + ssq[5]) + (ssq[6] + ssq[7]));
            float tb = ((ssq[8] + ssq[9]) + (ssq[10] + ssq[11])) + ((ssq[12] + ssq[13]) + (ssq[14] + ssq[15]));
            const float oa = __shfl_xor(ta, 32), ob = __shfl_xor(tb, 32);
            ta = h ? oa + ta : ta + oa;  // (h 0) + (h 1) in both halves
            tb = h ? ob + tb : tb + ob;
            xn = norm2_from_sumsq(ta + tb);
        }
        __syncthreads();
        // compare-free top-2 scan of this lane's 8 x 16 distances: the 7 low mantissa bits carry the position
        const float m2s = -2.0f * aux->inv_ss;  // (read here, not held across the stage loop: a scalar load per tile)
        float s1 = INFINITY, s2 = INFINITY;
        if (centred) {
            // no centre is under-used: every scale is 1 and ||x||^2 is left out -- v = fl(-2 dot + ||c||^2') is ONE fma (-2 dot is
            // exact), the same value the general form below produces through its three operations; no scale table read
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 cnv = *reinterpret_cast<const float4 *>(sCn + ct * 32 + 4 * h + 8 * g);
                    const float cn4[4] = {cnv.x, cnv.y, cnv.z, cnv.w};
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {  // two distances per v_pk_fma_f32 (each half is one IEEE fma)
                        const f32x2 a2 = {acc[ct][4 * g + j], acc[ct][4 * g + j + 1]}, c2 = {cn4[j], cn4[j + 1]}, m2 = {m2s, m2s};
                        const f32x2 v2 = __builtin_elementwise_fma(m2, a2, c2);
                        top2_push(s1, s2, __uint_as_float((__float_as_uint(v2.x) & 0xFFFFFF80u) | (unsigned)(ct * 16 + g * 4 + j)));
                        top2_push(s1, s2, __uint_as_float((__float_as_uint(v2.y) & 0xFFFFFF80u) | (unsigned)(ct * 16 + g * 4 + j + 1)));
                    }
                }
            }
        } else {
            const float xoff = xn;
#pragma unroll
            for (int ct = 0; ct < 8; ++ct) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int kl = ct * 32 + 4 * h + 8 * g;
                    const float4 cnv = *reinterpret_cast<const float4 *>(sCn + kl);
                    const float4 scv = *reinterpret_cast<const float4 *>(sSc + kl);
                    const float cn4[4] = {cnv.x, cnv.y, cnv.z, cnv.w};
                    const float sc4[4] = {scv.x, scv.y, scv.z, scv.w};
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {  // packed: each half is the same three IEEE operations as the scalar form
                        const f32x2 a2 = {acc[ct][4 * g + j], acc[ct][4 * g + j + 1]}, m2 = {m2s, m2s}, x2 = {xoff, xoff};
                        const f32x2 c2 = {cn4[j], cn4[j + 1]}, sc2 = {sc4[j], sc4[j + 1]};
                        f32x2 v2 = __builtin_elementwise_fma(m2, a2, x2);  // == (-2 dot) + xn
                        v2 = v2 + c2;
                        v2 = v2 * sc2;  // * (1/r) where the exact path divides by r
                        top2_push(s1, s2, __uint_as_float((__float_as_uint(v2.x) & 0xFFFFFF80u) | (unsigned)(ct * 16 + g * 4 + j)));
                        top2_push(s1, s2, __uint_as_float((__float_as_uint(v2.y) & 0xFFFFFF80u) | (unsigned)(ct * 16 + g * 4 + j + 1)));
                    }
                }
            }
        }
        const unsigned c7 = __float_as_uint(s1) & 127u;  // (ct, g, j) of the lane's minimum
        Top2 t = {s1, kbase + 4 * h + (int)((c7 >> 4) * 32 + ((c7 >> 2) & 3) * 8 + (c7 & 3)), s2}, o;
        o.d1 = __shfl_xor(t.d1, 32);
        o.k1 = __shfl_xor(t.k1, 32);
        o.d2 = __shfl_xor(t.d2, 32);
        run = top2_merge(run, top2_merge(t, o));

        if (EMIT != 0 && !GS && (ngroups == 1 || LISTED)) {
            // ---- acceptance test in BOTH half-lanes of a row, and for undecided rows the candidate emission (header above)
            // (l31 is laundered: the compiler otherwise computes &labels[row] in the prologue, keeps it across the stage loop in a
            // spilled register pair and reloads it from scratch in every tile's epilogue -- 6 % of the filter with not a row undecided)
            int l31e = l31;
            asm volatile("" : "+v"(l31e));
            const int64_t li = row0 + wq * 32 + l31e;  // the row itself, or its position in the list of undecided rows
            const int64_t row = li < n ? (LISTED ? (int64_t)recheck_list[li] : li) : -1;
            const float E = filter_bound(aux, __builtin_sqrtf(xn), e1c, e1r, e2coef);  // NaN: the row cannot be decided here
            const float tagged = 1.6e-5f * (fabsf(run.d1) + fabsf(run.d2));  // > (2^-16 + 2^-22) x 1.01
            // (every listed row fails the test again: same arithmetic as the filter's; the label is already the filter's k1.
            // K > 256, emission pass: the row is undecided by construction and its threshold comes from k_assign_merge, which
            // knows the minimum over ALL groups -- this group's own running minimum would give a wider window)
            const bool undecided = row >= 0 && (ngroups > 1 || !((run.d2 - run.d1) > 2.0f * E + tagged));
            if (EMIT == 2 && h == 0 && row >= 0) labels[row] = (int64_t)run.k1;
            if (__builtin_amdgcn_ballot_w64(undecided) != 0ull) {
                // compiler barrier: nothing of the rare block (its 128 + 128 LDS reads above all) may be hoisted in front of
                // the branch into the path every tile takes
                asm volatile("" ::: "memory");
                if (EMIT == 2 && undecided && h == 0) atomicAdd(&ctl->und_count, 1u);
                float T = ngroups > 1 ? out.und_T[li < n ? li : 0] : cand_threshold(run.d1, E);  // (derivation at cand_threshold)
                if (!undecided) T = -INFINITY;
                // bit (ct * 16 + g * 4 + j) of m: distance (ct, g, j) of this lane is not > T.  Branch-free: compare, 0 / 1, shift-or.
                unsigned m0 = 0u, m1 = 0u, m2 = 0u, m3 = 0u;
#define ACAV_CAND_MARK(v, idx)                                                   \
    {                                                                            \
        const unsigned hit_ = (v) > T ? 0u : 1u;                                 \
        if ((idx) < 32) m0 |= hit_ << ((idx) & 31);                              \
        else if ((idx) < 64) m1 |= hit_ << ((idx) & 31);                         \
        else if ((idx) < 96) m2 |= hit_ << ((idx) & 31);                         \
        else m3 |= hit_ << ((idx) & 31);                                         \
    }
                if (centred) {
#pragma unroll
                    for (int ct = 0; ct < 8; ++ct) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 cnv = *reinterpret_cast<const float4 *>(sCn + ct * 32 + 4 * h + 8 * g);
                            const float cn4[4] = {cnv.x, cnv.y, cnv.z, cnv.w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const float v = __builtin_fmaf(m2s, acc[ct][4 * g + j], cn4[j]);
                                ACAV_CAND_MARK(v, ct * 16 + g * 4 + j)
                            }
                        }
                    }
                } else {
                    const float xoff = xn;
#pragma unroll
                    for (int ct = 0; ct < 8; ++ct) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int kl = ct * 32 + 4 * h + 8 * g;
                            const float4 cnv = *reinterpret_cast<const float4 *>(sCn + kl);
                            const float4 scv = *reinterpret_cast<const float4 *>(sSc + kl);
                            const float cn4[4] = {cnv.x, cnv.y, cnv.z, cnv.w};
                            const float sc4[4] = {scv.x, scv.y, scv.z, scv.w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float v = __builtin_fmaf(m2s, acc[ct][4 * g + j], xoff);
                                v = v + cn4[j];
                                v = v * sc4[j];
                                ACAV_CAND_MARK(v, ct * 16 + g * 4 + j)
                            }
                        }
                    }
                }
#undef ACAV_CAND_MARK
                if (LISTED) {
                    // emission pass: the row's candidates of this group go to its list in LDS (a row may collect them over
                    // several groups); the row is settled after the last group, below
                    const int r = wq * 32 + l31e;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        unsigned bits = w == 0 ? m0 : w == 1 ? m1 : w == 2 ? m2 : m3;
                        while (bits) {
                            const int bpos = __builtin_ctz(bits);
                            bits &= bits - 1;
                            const unsigned idx = (unsigned)(w * 32 + bpos);  // (ct, g, j) -> centre, as for the minimum
                            const unsigned pos = atomicAdd(&sCandN[r], 1u);
                            if (pos < CAND_MAX)
                                sCandK[r * CAND_MAX + pos] = (unsigned short)(kbase + 4 * h + (int)((idx >> 4) * 32 + ((idx >> 2) & 3) * 8 + (idx & 3)));
                        }
                    }
                } else {
                const unsigned m[4] = {m0, m1, m2, m3};
                const unsigned mine = __builtin_popcount(m[0]) + __builtin_popcount(m[1]) + __builtin_popcount(m[2]) + __builtin_popcount(m[3]);
                const unsigned other = __shfl_xor(mine, 32);
                const unsigned total = mine + other;
                CandRow *const crow = out.crow;
                CandPair *const cpair = out.cpair;
                const unsigned pair_cap = out.pair_cap;
                unsigned long long got = 0ull;
                bool full = false;  // the row takes the full exact sweep instead
                if (undecided && h == 0) {
                    if (total > CAND_MAX) full = true;
                    else {
                        got = atomicAdd(&ctl->alloc, ((unsigned long long)total << 32) | 1ull);
                        if ((got >> 32) + total > (unsigned long long)pair_cap) {  // pool exhausted (the slot stays, empty)
                            full = true;
                            atomicAdd(&ctl->pool_over, 1u);
                        }
                    }
                    if (full) {
                        const unsigned fslot = atomicAdd(&ctl->f32_count, 1u);
                        out.f32_list[fslot] = (int)row;
                    }
                    if (total <= CAND_MAX) {
                        // (a row whose range runs past the pool's end -- pair_base + cnt > pair_cap -- is "lost": on the f32 list)
                        const CandRow cr = {(int)row, (unsigned)(got >> 32), total, xn};
                        crow[(unsigned)got] = cr;
                    }
                }
                const unsigned slot_lo = __shfl(( unsigned)got, l31), base_lo = __shfl((unsigned)(got >> 32), l31);
                const bool full_row = __shfl((int)full, l31) != 0;
                // (a row that lost its range to the pool's end still writes the pairs that fit: every entry below the cap is a
                // valid (slot, centre) -- k_assign_cand walks the pool as one array; that row's label comes from the full sweep)
                (void)full_row;
                if (undecided && total <= CAND_MAX) {
                    unsigned wpos = base_lo + (h ? other : 0u);  // the h = 0 lane's candidates first
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        unsigned bits = w == 0 ? m0 : w == 1 ? m1 : w == 2 ? m2 : m3;
                        while (bits) {
                            const int bpos = __builtin_ctz(bits);
                            bits &= bits - 1;
                            const unsigned idx = (unsigned)(w * 32 + bpos);  // (ct, g, j) -> centre, as for the minimum
                            const CandPair cp = {slot_lo, kbase + 4 * h + (int)((idx >> 4) * 32 + ((idx >> 2) & 3) * 8 + (idx & 3))};
                            if (wpos < pair_cap) cpair[wpos] = cp;
                            ++wpos;
                        }
                    }
                }
                }
            }
        }
    }
    if (LISTED) {
        // ---- emission pass: settle the tile's rows from their LDS lists (the two half-lanes of a row are in one wave: their
        // LDS appends are ordered before this read)
        const int r = wq * 32 + l31;
        const int64_t li = row0 + r;
        if (h == 0 && li < n) {
            const int row = recheck_list[li];
            const unsigned total = sCandN[r];
            bool full = total > CAND_MAX || total == 0u;
            unsigned long long got = 0ull;
            if (!full) {
                got = atomicAdd(&ctl->alloc, ((unsigned long long)total << 32) | 1ull);
                if ((got >> 32) + total > (unsigned long long)out.pair_cap) {  // pool exhausted (the slot stays: a "lost" row)
                    full = true;
                    atomicAdd(&ctl->pool_over, 1u);
                }
                const CandRow cr = {row, (unsigned)(got >> 32), total, xn};
                out.crow[(unsigned)got] = cr;
                for (unsigned q = 0; q < total; ++q) {
                    const unsigned wpos = (unsigned)(got >> 32) + q;
                    const CandPair cp = {(unsigned)got, (int)sCandK[r * CAND_MAX + q]};
                    if (wpos < out.pair_cap) out.cpair[wpos] = cp;
                }
            }
            if (full) {
                const unsigned fslot = atomicAdd(&ctl->f32_count, 1u);
                out.f32_list[fslot] = row;
            }
        }
    }
#ifdef ACAV_RW_PROF
    if (lane == 0 && (blockIdx.x & 15) == 3 && (wq == 0 || wq == NW - 1)) {
        const int o = wq == 0 ? 0 : 8;
        for (int i = 0; i < 6; ++i) atomicAdd(&g_rw_prof[o + i], (unsigned long long)rwp[i]);
        atomicAdd(&g_rw_prof[o + 6], (unsigned long long)(clock64() - rw_tstart));
        atomicAdd(&g_rw_prof[o + 7], (unsigned long long)(wall_clock64() - rw_wstart));
    }
#endif
    const int64_t row = row0 + wq * 32 + l31;
    if (GS) {
        if (h == 0 && row < n) {
            const Top2Rec rec = {run.d1, run.k1, run.d2, xn};
            grec[(size_t)cg0 * (size_t)n + (size_t)row] = rec;
        }
        return;
    }
    if (LISTED) continue;  // emitted inside the group loop (the accumulators live there); next tile of the list
    if (EMIT == 2) break;
    if (h == 0 && row < n) {
        const float E = filter_bound(aux, __builtin_sqrtf(xn), e1c, e1r, e2coef);
        labels[row] = (int64_t)run.k1;
        // the position tag perturbs a distance by < 2^-16 of ITS OWN magnitude (and fl(1/r) by 2 ulp more): charged to the
        // two distances that are compared instead of to (||x|| + cmax)^2 -- with a large common component the distances
        // are orders of magnitude smaller than the norms
        const float tagged = 1.6e-5f * (fabsf(run.d1) + fabsf(run.d2));  // > (2^-16 + 2^-22) x 1.01
        if (!((run.d2 - run.d1) > 2.0f * E + tagged)) {  // also catches NaN / inf
            const unsigned slot = atomicAdd(recheck_count, 1u);
            recheck_list[slot] = (int)row;
        }
    }
    break;  // not the emission pass: one tile per workgroup
  }
}

// Folds the per-group records of the group-split filter (ascending group order = the order of the single-workgroup loop;
// top2_merge breaks distance ties towards the lower centre index, so the fold is order-independent anyway) and applies
// k_assign_f16_rw's acceptance test.  One thread per row; 16-byte records, coalesced.
__global__ __launch_bounds__(256) void k_assign_merge(const Top2Rec *__restrict__ grec, int ngroups, int64_t n,
                                                      const CentersAux *__restrict__ aux, float e1c, float e1r, float e2coef,
                                                      int64_t *__restrict__ labels, int *__restrict__ recheck_list,
                                                      unsigned *__restrict__ recheck_count, float *__restrict__ und_T)
{
    // und_T != NULL: the listed rows go to the emission pass; their candidate threshold (which needs the minimum over ALL
    // groups: only this kernel has it) travels with the list
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    Top2Rec r0 = grec[row];
    Top2 run = {r0.d1, r0.k1, r0.d2};
    for (int g = 1; g < ngroups; ++g) {
        const Top2Rec rg = grec[(size_t)g * (size_t)n + (size_t)row];
        const Top2 o = {rg.d1, rg.k1, rg.d2};
        run = top2_merge(run, o);
    }
    const float E = filter_bound(aux, __builtin_sqrtf(r0.xn), e1c, e1r, e2coef);
    labels[row] = (int64_t)run.k1;
    const float tagged = 1.6e-5f * (fabsf(run.d1) + fabsf(run.d2));
    if (!((run.d2 - run.d1) > 2.0f * E + tagged)) {
        const unsigned slot = atomicAdd(recheck_count, 1u);
        recheck_list[slot] = (int)row;
        if (und_T) und_T[slot] = cand_threshold(run.d1, E);
    }
}

// ------------------------------------------------------------------------- k_assign_cand
// Exact canonical distances of the (undecided row, candidate centre) pairs the filter emitted; the row's label = their
// first-index lexmin.  Waves are independent.  Work item = S consecutive row slots = one contiguous pair range (slots and
// pair ranges come from one atomic: CandPair); S adapts to the load (few rows: small items over many waves; many rows: 128
// slots per item), item i goes to wave i mod W.  A wave walks its item in passes of up to 64 consecutive pairs that END AT A
// ROW BOUNDARY and span at most 32 rows: every row is settled inside one pass -- no atomics, no fences.
// (A first version cut the pool into fixed 64-pair ranges and folded the rows cut by a range boundary through global
// atomics + __threadfence(): the fence is buffer_wbl2 + buffer_inv -- an L2 write-back and an L1 invalidate per cut row, 33 k
// of them per launch at 62 % undecided rows: 2.4 ms instead of 1.05, TA busy 100 cycles per load instead of 22.)
//   * lane = pair: the canonical dot is a sequential fp32 FMA chain per 256-column segment, folded left to right, so a
//     pair's d columns are one dependent chain in one lane -- 32 FMAs per 32-column stage, 64 pairs side by side.
//   * per stage the wave brings its 64 centre chunks and its <= 32 row chunks (128 bytes each) in as coalesced 16-byte pieces
//     (8 lanes per chunk: every fetched line is used whole), one stage ahead in registers, and transposes them through its
//     own 12 KB of LDS -- 16-byte slots XOR-swizzled by ((chunk >> 1) & 7): lane = chunk reads are conflict-free.  No
//     workgroup barrier: LDS operations of one wave execute in order.
//     (Per-lane streaming of the lane's own two lines, no LDS, measured 2.6x slower: 12 waves x 82 lines thrash the 32 KB L1
//     between the eight 16-byte reads of a line.)
//   * epilogue = dist_epilogue() of the exact sweep; segmented lexmin over the adjacent lanes of a row (<= CAND_MAX = 16).
constexpr int CAND_XROWS = 32;    // distinct rows per pass

__device__ __forceinline__ unsigned long long cand_key(float v, int k)
{
    unsigned u = __float_as_uint(v);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;  // total order of the floats as unsigned
    return ((unsigned long long)u << 32) | (unsigned)k;
}

__global__ __launch_bounds__(256, 3) void k_assign_cand(const float *__restrict__ x, int d, const float *__restrict__ centers,
                                                         const float *__restrict__ cn, const float *__restrict__ counts, float thr,
                                                         float r, const AssignCtl *__restrict__ ctl, const CandRow *__restrict__ crow,
                                                         const CandPair *__restrict__ cpair, unsigned pair_cap,
                                                         int64_t *__restrict__ labels)
{
    __shared__ __attribute__((aligned(16))) float sC[4][64 * 32];          // [wave][pair chunk of 32 columns], swizzled
    __shared__ __attribute__((aligned(16))) float sX[4][CAND_XROWS * 32];  // [wave][row chunk]
    const unsigned long long alloc = ctl->alloc;
    const unsigned nslots = (unsigned)alloc;
    unsigned npairs = (unsigned)(alloc >> 32);
    if (npairs > pair_cap) npairs = pair_cap;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned gw = blockIdx.x * 4u + (unsigned)wave, nw = gridDim.x * 4u;
    unsigned S = (nslots + nw - 1u) / nw;  // slots per item: about one item per wave, within [8, 128]
    S = S < 8u ? 8u : (S > 128u ? 128u : S);
    const unsigned nitems = (nslots + S - 1u) / S;
    const int nchunks = d / 32;
    float *wc = sC[wave], *wx = sX[wave];
    int *wrow = reinterpret_cast<int *>(wx);
    const int sub = lane >> 3, piece = lane & 7;  // staging role: 16-byte piece `piece` of chunk 8 q + sub

    for (unsigned item = gw; item < nitems; item += nw) {
        const unsigned s0 = item * S, s1 = s0 + S < nslots ? s0 + S : nslots;
        unsigned p0 = crow[s0].pair_base;
        unsigned pe = s1 < nslots ? crow[s1].pair_base : npairs;
        if (pe > npairs) pe = npairs;  // the pool's end: pairs past it were never written (their rows are "lost": f32 list)
        while (p0 < pe) {  // wave-uniform
            const unsigned p = p0 + (unsigned)lane;
            const CandPair cp = cpair[p < pe ? p : pe - 1u];  // lanes past the item shadow its last pair
            const CandRow cr = crow[cp.slot];
            const unsigned xs = cp.slot - (unsigned)__shfl((int)cp.slot, 0);  // row chunk of this lane's row; ascends with the lane
            const unsigned wend = p0 + 64u < pe ? p0 + 64u : pe;
            const unsigned row_end = cr.pair_base + cr.cnt;                   // one past the row's last pair
            const bool lost = row_end > pair_cap;                              // only the pool's last rows
            // taken = a prefix of the lanes: rows that end inside the window (the first row always does: <= 16 pairs), at most 32
            const bool taken = p < pe && xs < (unsigned)CAND_XROWS && (row_end <= wend || lost);
            const unsigned ntaken = (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(taken));
            const bool live = taken && !lost;
            const unsigned xsc = xs < (unsigned)CAND_XROWS ? xs : (unsigned)CAND_XROWS - 1u;
            const unsigned nrows_w = (unsigned)__shfl((int)xsc, 63) + 1u;
            // the row of every row chunk travels through LDS (lanes of one row write the same value)
            __builtin_amdgcn_wave_barrier();
            if (xs < (unsigned)CAND_XROWS) wrow[xs] = cr.row;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // staging sources: centre chunk 8 q + sub = the centre of pair lane 8 q + sub; row chunk 8 q + sub (q < 4)
            const float *cb0, *cb1, *cb2, *cb3, *cb4, *cb5, *cb6, *cb7, *xb0, *xb1, *xb2, *xb3;
#define ACAV_CAND_CB(q) cb##q = centers + (size_t)__shfl(cp.k, 8 * q + sub) * d + piece * 4;
#define ACAV_CAND_XB(q)                                                                      \
    {                                                                                        \
        const unsigned want = (unsigned)(8 * q + sub);                                       \
        xb##q = x + (size_t)wrow[want < nrows_w ? want : 0u] * d + piece * 4;                \
    }
            ACAV_CAND_CB(0) ACAV_CAND_CB(1) ACAV_CAND_CB(2) ACAV_CAND_CB(3) ACAV_CAND_CB(4) ACAV_CAND_CB(5) ACAV_CAND_CB(6) ACAV_CAND_CB(7)
            ACAV_CAND_XB(0) ACAV_CAND_XB(1) ACAV_CAND_XB(2) ACAV_CAND_XB(3)
#undef ACAV_CAND_CB
#undef ACAV_CAND_XB
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // wrow is dead: the first stage overwrites it
            const bool x1 = nrows_w > 8u, x2 = nrows_w > 16u, x3 = nrows_w > 24u;  // wave-uniform: row-chunk instructions needed

            f32x4 rc0, rc1, rc2, rc3, rc4, rc5, rc6, rc7, rx0, rx1, rx2, rx3;
            rx1 = rx2 = rx3 = f32x4{0.f, 0.f, 0.f, 0.f};
#define ACAV_CAND_LD4(p) (*reinterpret_cast<const f32x4 *>(p))
#define ACAV_CAND_LOAD(c)                          \
    rc0 = ACAV_CAND_LD4(cb0 + (c) * 32);           \
    rc1 = ACAV_CAND_LD4(cb1 + (c) * 32);           \
    rc2 = ACAV_CAND_LD4(cb2 + (c) * 32);           \
    rc3 = ACAV_CAND_LD4(cb3 + (c) * 32);           \
    rc4 = ACAV_CAND_LD4(cb4 + (c) * 32);           \
    rc5 = ACAV_CAND_LD4(cb5 + (c) * 32);           \
    rc6 = ACAV_CAND_LD4(cb6 + (c) * 32);           \
    rc7 = ACAV_CAND_LD4(cb7 + (c) * 32);           \
    rx0 = ACAV_CAND_LD4(xb0 + (c) * 32);           \
    if (x1) rx1 = ACAV_CAND_LD4(xb1 + (c) * 32);   \
    if (x2) rx2 = ACAV_CAND_LD4(xb2 + (c) * 32);   \
    if (x3) rx3 = ACAV_CAND_LD4(xb3 + (c) * 32);
#define ACAV_CAND_SLOT(q) ((8 * q + sub) * 32 + ((piece ^ (((8 * q + sub) >> 1) & 7)) << 2))
            ACAV_CAND_LOAD(0)
            float acc = 0.f, tot = 0.f;
            const int swl = (lane >> 1) & 7, swx = ((int)xsc >> 1) & 7;
            const float *pc = wc + lane * 32, *px = wx + xsc * 32;
            for (int c = 0; c < nchunks; ++c) {
                __builtin_amdgcn_wave_barrier();  // the wave's reads of the previous stage are behind (LDS ops of a wave are in order)
                *reinterpret_cast<f32x4 *>(wc + ACAV_CAND_SLOT(0)) = rc0;
                *reinterpret_cast<f32x4 *>(wc + ACAV_CAND_SLOT(1)) = rc1;
                *reinterpret_cast<f32x4 *>(wc + ACAV_CAND_SLOT(2)) = rc2;
                *reinterpret_cast<f32x4 *>(wc + ACAV_CAND_SLOT(3)) = rc3;
                *reinterpret_cast<f32x4 *>(wc + ACAV_CAND_SLOT(4)) = rc4;
                *reinterpret_cast<f32x4 *>(wc + ACAV_CAND_SLOT(5)) = rc5;
                *reinterpret_cast<f32x4 *>(wc + ACAV_CAND_SLOT(6)) = rc6;
                *reinterpret_cast<f32x4 *>(wc + ACAV_CAND_SLOT(7)) = rc7;
                *reinterpret_cast<f32x4 *>(wx + ACAV_CAND_SLOT(0)) = rx0;
                if (x1) *reinterpret_cast<f32x4 *>(wx + ACAV_CAND_SLOT(1)) = rx1;
                if (x2) *reinterpret_cast<f32x4 *>(wx + ACAV_CAND_SLOT(2)) = rx2;
                if (x3) *reinterpret_cast<f32x4 *>(wx + ACAV_CAND_SLOT(3)) = rx3;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int cn_ = c + 1 < nchunks ? c + 1 : c;  // the last stage re-loads itself (discarded): no branch around the loads
                ACAV_CAND_LOAD(cn_)                           // in flight under the chains below
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 a = *reinterpret_cast<const float4 *>(pc + ((j ^ swl) << 2));
                    const float4 b = *reinterpret_cast<const float4 *>(px + ((j ^ swx) << 2));
                    acc = __builtin_fmaf(a.x, b.x, acc);
                    acc = __builtin_fmaf(a.y, b.y, acc);
                    acc = __builtin_fmaf(a.z, b.z, acc);
                    acc = __builtin_fmaf(a.w, b.w, acc);
                }
                if ((c & 7) == 7 || c + 1 == nchunks) {  // end of a 256-column segment: tot = (first) ? acc : tot + acc
                    tot = c < 8 ? acc : tot + acc;
                    acc = 0.f;
                }
            }
#undef ACAV_CAND_LOAD
#undef ACAV_CAND_LD4
#undef ACAV_CAND_SLOT
            const bool disc = counts[cp.k] < thr;
            const float dist = dist_epilogue(tot, cr.xn, cn[cp.k], disc, r);
            unsigned long long key = cand_key(dist, cp.k);
            // segmented lexmin over the run of lanes with this lane's slot (inclusive scan towards higher lanes)
            const unsigned slot = live ? cp.slot : 0xFFFFFFFFu - (unsigned)lane;  // other lanes: unique slots, no merging
#pragma unroll
            for (int off = 1; off < (int)CAND_MAX; off <<= 1) {
                const unsigned long long ko = __shfl_up(key, off);
                const unsigned so = __shfl_up(slot, off);
                if (lane >= off && so == slot) key = ko < key ? ko : key;
            }
            const unsigned snext = __shfl_down(slot, 1);
            if (live && (lane == 63 || snext != slot))  // last lane of the row's run: the row sat in this pass whole
                labels[cr.row] = (int64_t)(unsigned)(key & 0xFFFFFFFFull);
            p0 += ntaken;
        }
    }
}

// out [n][dp] <- in [n][d], columns d .. dp-1 zero: a view whose width is not a multiple of the filter's 32-column stage
// (SlowFast's 88-wide layer, clustering/code/models/slowfast.py:31) takes the filter on a padded copy.  Zero columns change
// nothing in the canonical arithmetic: fmaf(0, 0, s) == s in every dot and norm chain, for the filter AND for the exact
// re-check of the rows it cannot decide (both run on the padded rows and centres).
__global__ __launch_bounds__(256) void k_pad_rows(const float *__restrict__ in, float *__restrict__ out, int64_t n, int d, int dp)
{
    const int64_t total = n * dp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / dp;
        const int j = (int)(i - r * dp);
        out[i] = j < d ? in[r * d + j] : 0.0f;
    }
}

}  // namespace

// The filter's copy of the centres: bf16 of c (or of c - mu when no centre is under-used), the mean centre, and the
// scalars of the acceptance bound.  Stream-ordered on the handle's stream; a no-op while the copy matches the state.
// Called by the assign sweep, and by the training entry points / set_state right after they change the state, so that a
// sweep that follows training finds the copy ready (the preparation used to sit inside every first sweep: ~25 us).
int acav_kmeans::prepare_filter()
{
    if (cb16_valid) return ACAV_OK;
    if (K < 2 || warm()) return ACAV_OK;  // the filter does not run on this shape / state
    hipStream_t st = ctx.stream;
    const int Kp = (K + 255) / 256 * 256;  // whole groups of 256 centre slots (stage-major copy: k_centers_f16)
    // a width that is not a multiple of the 32-column stage: the filter (and its exact re-check) run on zero-padded copies
    const int dp = filter_d();
    const float *fc = centers.as<float>();
    if (dp != d) {
        ACAV_TRY(cpad.ensure(sizeof(float) * (size_t)K * dp));
        hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)std::min<int64_t>(((int64_t)K * dp + 255) / 256, 4096)), dim3(256), 0, st,
                           centers.as<float>(), cpad.as<float>(), (int64_t)K, d, dp);
        fc = cpad.as<float>();
    }
    ACAV_TRY(cb16.ensure(sizeof(unsigned short) * (size_t)Kp * dp));
    ACAV_TRY(caux.ensure(sizeof(CentersAux)));
    ACAV_HIP_TRY(hipMemsetAsync(caux.p, 0, sizeof(CentersAux), st));
    ACAV_TRY(cmu.ensure(sizeof(float) * (size_t)dp));
    hipLaunchKernelGGL(k_centers_mu, dim3((unsigned)((dp + 31) / 32)), dim3(256), 0, st, fc, cn.as<float>(),
                       counts.as<float>(), K, dp, threshold(), cmu.as<float>(), caux.as<CentersAux>());
    hipLaunchKernelGGL(k_centers_amaxc, dim3((unsigned)K), dim3(256), 0, st, fc, cmu.as<float>(), dp, K, caux.as<CentersAux>());
    hipLaunchKernelGGL(k_centers_norm, dim3((unsigned)K), dim3(256), 0, st, fc, cmu.as<float>(), dp, K, caux.as<CentersAux>());
    float usub;
    ACAV_TRY(half_underflow_unit(ctx.device, st, &usub));
    hipLaunchKernelGGL(k_centers_scale, dim3(1), dim3(64), 0, st, caux.as<CentersAux>(), dp, usub);
    hipLaunchKernelGGL(k_centers_f16, dim3((unsigned)Kp), dim3(256), 0, st, fc, cmu.as<float>(), dp, K, Kp, cb16.as<fl16>(),
                       caux.as<CentersAux>());
    ACAV_HIP_TRY(hipGetLastError());
    {   // which instantiation the sweeps launch (rows scaled before the conversion or not) is decided on the device: 4 bytes back.
        // Every caller of prepare_filter has just synchronised the stream (end of a training call, set_state) or is about to
        // run a sweep: ~10 us, once per change of state.
        float sx_host = 1.0f;
        ACAV_HIP_TRY(hipMemcpyAsync(&sx_host, &caux.as<CentersAux>()->sx, sizeof(float), hipMemcpyDeviceToHost, st));
        ACAV_HIP_TRY(hipStreamSynchronize(st));
        filter_rows_scaled = sx_host != 1.0f;
    }
    cb16_valid = true;
    // the sweep's one-time objects, so that the first sweep of a handle does not create them inside its own timing
    if (!ev_f0) {
        ACAV_HIP_TRY(hipEventCreate(&ev_f0));
        ACAV_HIP_TRY(hipEventCreate(&ev_f1));
    }
    if (!cand_ctl.p) {
        ACAV_TRY(cand_ctl.ensure(sizeof(AssignCtl)));
        ACAV_HIP_TRY(hipMemsetAsync(cand_ctl.p, 0, sizeof(AssignCtl), st));
    }
    return ACAV_OK;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device, size) instead of once per sweep
static int dyn_lds_once(const void *fn, int device, int bytes)
{
    static std::mutex mu;
    static std::set<std::tuple<const void *, int, int>> done;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({fn, device, bytes})) return ACAV_OK;
    ACAV_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({fn, device, bytes});
    return ACAV_OK;
}

static int read_ctl(acav_kmeans *km, AssignCtl *out)
{
    memset(out, 0, sizeof(*out));
    if (km->cand_ctl.p && km->n_filter_launches) {
        ACAV_HIP_TRY(hipSetDevice(km->ctx.device));
        ACAV_HIP_TRY(hipMemcpyAsync(out, km->cand_ctl.p, sizeof(*out), hipMemcpyDeviceToHost, km->ctx.stream));
        ACAV_HIP_TRY(hipStreamSynchronize(km->ctx.stream));
    }
    return ACAV_OK;
}

ACAV_EXPORT int acav_kmeans_filter_stats(acav_kmeans *km, int64_t *filter_launches, int64_t *rows, int64_t *rechecked)
{
    ACAV_REQUIRE(km, ACAV_EINVAL, "handle is NULL");
    AssignCtl c;
    ACAV_TRY(read_ctl(km, &c));
    if (filter_launches) *filter_launches = km->n_filter_launches;
    if (rows) *rows = (int64_t)km->last_rows;
    // rows the filter could not decide = candidate rows + rows sent straight to the full exact sweep
    // (with the candidate path: the filter's own list; else everything undecided went straight to the full-sweep list)
    if (rechecked) *rechecked = c.last_und ? (int64_t)c.last_und : (int64_t)c.last_f32;
    return ACAV_OK;
}

ACAV_EXPORT int acav_kmeans_recheck_stats(acav_kmeans *km, int64_t *cand_rows, int64_t *cand_pairs, int64_t *full_rows)
{
    ACAV_REQUIRE(km, ACAV_EINVAL, "handle is NULL");
    AssignCtl c;
    ACAV_TRY(read_ctl(km, &c));
    if (cand_rows) *cand_rows = (int64_t)(unsigned)c.last_alloc - (int64_t)c.last_pool_over;
    if (cand_pairs) *cand_pairs = (int64_t)(c.last_alloc >> 32);
    if (full_rows) *full_rows = (int64_t)c.last_f32;
    return ACAV_OK;
}

ACAV_EXPORT int acav_kmeans_filter_time(acav_kmeans *km, float *ms)
{
    ACAV_REQUIRE(km && ms, ACAV_EINVAL, "NULL argument");
    ACAV_REQUIRE(km->ev_f0 && km->n_filter_launches > 0, ACAV_ESTATE, "no filter launch yet");
    ACAV_HIP_TRY(hipSetDevice(km->ctx.device));
    ACAV_HIP_TRY(hipEventSynchronize(km->ev_f1));
    ACAV_HIP_TRY(hipEventElapsedTime(ms, km->ev_f0, km->ev_f1));
    return ACAV_OK;
}

ACAV_EXPORT int acav_kmeans_assign(acav_kmeans *km, const float *x, int64_t n, int64_t *labels, float *mean_dist)
{
    ACAV_REQUIRE(km, ACAV_EINVAL, "handle is NULL");
    ACAV_REQUIRE(n >= 0, ACAV_EINVAL, "n must be >= 0");
    ACAV_REQUIRE(n == 0 || (x && labels), ACAV_EINVAL, "NULL argument");  // an empty tensor has no storage
    ACAV_REQUIRE(!km->warm(), ACAV_ESTATE,
                 "count=%lld < initial_rounds*k=%lld: labels come from the warm-up rng (acav_rng_warmup_best)",
                 (long long)km->count, (long long)km->initial_rounds * km->K);
    if (n == 0) {
        if (mean_dist) *mean_dist = NAN;  // torch: mean of an empty tensor
        return ACAV_OK;
    }
    ACAV_HIP_TRY(hipSetDevice(km->ctx.device));
    hipStream_t st = km->ctx.stream;
    const void *dx = nullptr;
    ACAV_TRY(to_device(x, sizeof(float) * (size_t)n * km->d, km->stage_x, st, &dx));
    const bool lab_dev = is_device_ptr(labels);
    int64_t *dlab = labels;
    if (!lab_dev) {
        ACAV_TRY(km->stage_lab.ensure(sizeof(int64_t) * (size_t)n));
        dlab = km->stage_lab.as<int64_t>();
    }
    const int64_t grid = (n + AS_ROWS - 1) / AS_ROWS;
    ACAV_REQUIRE(grid <= 0x7fffffff, ACAV_EINVAL, "n too large for one launch");
    ACAV_TRY(km->wg_sum.ensure(sizeof(double) * (size_t)grid));
    const bool fast = (km->d % AS_BK) == 0 && ((uintptr_t)dx & 15) == 0;
    // bf16 filter + exact re-check (bit-identical labels, HBM-bound when the clusters are separated): taken when
    // the caller does not need the mean distance (the filter's distances are approximate)
    const char *noflt = getenv("ACAV_ASSIGN_EXACT_ONLY");
    // d % 32 != 0 (88-wide SlowFast layer): the filter sweep runs on zero-padded copies of the rows and the centres (k_pad_rows;
    // ACAV_FILTER_PAD=0: the guarded exact sweep as before)
    const char *vpad = getenv("ACAV_FILTER_PAD");
    const bool ragged = (km->d % FD_BK) != 0;
    const bool filter = !mean_dist && (ragged ? !(vpad && vpad[0] == '0') : fast) && km->K >= 2 && n >= FB_ROWS &&
                        !(noflt && noflt[0] == '1') && n < 0x7fffffff;
    if (filter) {
        ACAV_TRY(km->prepare_filter());
        const int fd = km->filter_d();  // the width the sweep runs at
        const float *fc = km->centers.as<float>();
        if (ragged) {
            ACAV_TRY(km->xpad.ensure(sizeof(float) * (size_t)n * fd));
            hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)std::min<int64_t>((n * fd + 255) / 256, 65536)), dim3(256), 0, st,
                               static_cast<const float *>(dx), km->xpad.as<float>(), n, km->d, fd);
            dx = km->xpad.p;
            fc = km->cpad.as<float>();
        }
        ACAV_TRY(km->recheck_list.ensure(sizeof(int) * (size_t)n * 2));  // [undecided rows | rows for the full exact sweep]
        // control block of the sweep: zero when a sweep starts -- zeroed here once, and by the last kernel of every sweep
        if (!km->cand_ctl.p) {
            ACAV_TRY(km->cand_ctl.ensure(sizeof(AssignCtl)));
            ACAV_HIP_TRY(hipMemsetAsync(km->cand_ctl.p, 0, sizeof(AssignCtl), st));
        }
        AssignCtl *ctl = km->cand_ctl.as<AssignCtl>();
        unsigned *f32_count = &ctl->f32_count;
        // candidate-restricted exact re-check (K <= 256: the filter's epilogue emits the candidates); ACAV_ASSIGN_CAND=0
        // sends every undecided row to the full exact sweep as in round 3
        const char *vcand = getenv("ACAV_ASSIGN_CAND");
        // (K > 256: through the emission pass over the rows k_assign_merge lists.  n < 2^27: slots and pairs -- at most 16 per row --
        // share one 64-bit allocator word, 32 bits each)
        const bool cand = !(vcand && vcand[0] == '0') && n < ((int64_t)1 << 27);
        // ACAV_ASSIGN_EMIT=0: experiment -- the lean filter kernel (no emission code) even with the candidate path on
        // ACAV_ASSIGN_EMIT: 0 = no emission at all (undecided rows -> full exact sweep), 1 = lean filter + emission pass over the
        // undecided rows, 2 (default) = emission in place in the filter's own epilogue (one pass)
        const char *vemit = getenv("ACAV_ASSIGN_EMIT");
        const bool emit_allowed = !(vemit && vemit[0] == '0');
        const bool emit_inplace = !(vemit && vemit[0] == '1');
        const uint64_t pair_cap64 = std::min<uint64_t>(std::max<uint64_t>(4ull * (uint64_t)n, 65536ull), 0x7fffffffull);
        unsigned pair_cap = (unsigned)pair_cap64;
        if (const char *vcap = getenv("ACAV_CAND_PAIR_CAP")) {  // tests: force the pool-overflow path
            const long v = atol(vcap);
            if (v > 0 && (uint64_t)v < pair_cap64) pair_cap = (unsigned)v;
        }
        if (cand) {
            ACAV_TRY(km->cand_rows.ensure(sizeof(CandRow) * (size_t)n));
            ACAV_TRY(km->cand_pairs.ensure(sizeof(CandPair) * (size_t)pair_cap));
            if (km->K > 256) ACAV_TRY(km->cand_T.ensure(sizeof(float) * (size_t)n));
        }
        CandRow *crow = cand ? km->cand_rows.as<CandRow>() : (CandRow *)nullptr;
        CandPair *cpair = cand ? km->cand_pairs.as<CandPair>() : (CandPair *)nullptr;
        int *und_list = km->recheck_list.as<int>(), *f32_list = und_list + n;
        const CandOut cout = {crow, cpair, f32_list, cand && km->K > 256 ? km->cand_T.as<float>() : (float *)nullptr, pair_cap};
        const double acc = 1.01 * (double)fd * ldexp(1.0, -24);  // accumulation error of one fp32 dot, relative
        const float e1c = (float)(2.02 * (ldexp(1.0, -10) * 1.002 + acc) * 1.001);  // x ||c'|| ||x||: half roundings (2^-11 per operand) + filter dot
        const float e1r = (float)(2.02 * (acc + ldexp(1.0, -24)) * 1.001);           // x ||c|| ||x||: canonical dot, c - mu
        // rows are read exactly once: non-temporal DMA policy (ACAV_FILTER_NT=0 restores the default policy)
        const char *vnt = getenv("ACAV_FILTER_NT");
        const bool nt = !(vnt && vnt[0] == '0');
        constexpr bool rw = true;  // (round 1's wave layout, ACAV_FILTER_V1, went with the move to half-precision operands)
        const float e2 = (float)ldexp(1.0, -20);  // epilogue roundings only, the tag is charged separately
        // Tile shape and DMA schedule of k_assign_f16_rw (template parameters there).  K > 256: one workgroup per (row tile,
        // centre group) pair, the pairs of a tile side by side on one XCD, 256-row tiles (8 waves), centre ring of 3, DMA
        // pieces spread between the MFMAs -- rows from HBM once.  Knobs for A/B runs: ACAV_FILTER_GS=0 (loop over the groups
        // inside one workgroup), ACAV_FILTER_NW=4|8, ACAV_FILTER_SCHED=0|2.
        const int ngroups = (km->K + 255) / 256;
        const char *vgs = getenv("ACAV_FILTER_GS"), *vnw = getenv("ACAV_FILTER_NW"), *vsc = getenv("ACAV_FILTER_SCHED");
        const bool gs = rw && ngroups > 1 && !(vgs && vgs[0] == '0');
        // (narrow views, d <= 256: a pair is only 4-8 stages long and its ring fill and epilogue weigh as much as its stage loop
        // -- two 128-row workgroups per CU hide them under each other: d = 128, K = 1024: 0.59 vs 0.72 ms per 1.25M rows)
        const int nw = !rw ? 4 : (vnw && vnw[0] == '8') ? 8 : (vnw && vnw[0] == '4') ? 4 : (gs ? (fd <= 256 ? 4 : 8) : FILTER_NW_DEFAULT);
        const bool nt_eff = gs ? !(vnt && vnt[0] == '0') : nt;  // nt rows are still found in L2 by the tile's other groups (PMC)
        const int dcr = nw == 8 ? 3 : 2;  // centre ring depth (3 only fits the one-workgroup-per-CU tile)
        const int sched = nw == 8 ? 2 : (vsc ? (vsc[0] == '2' ? 2 : 0) : FILTER_SCHED_DEFAULT);
        typedef void (*FilterKern)(const float *, int64_t, int, const fl16 *, const float *, const float *, int, float, float,
                                   const CentersAux *, float, float, float, int64_t *, int *, unsigned *, AssignCtl *, Top2Rec *,
                                   CandOut);
        FilterKern rwk = nullptr;
        // K <= 256: the filter emits in place (or, ACAV_ASSIGN_EMIT=1, lists for the emission pass); K > 256: the (tile, group) pairs
        // cannot know a row's minimum over all groups -- k_assign_merge lists the undecided rows with their thresholds and the
        // emission pass runs the filter's main loop once more over those rows, all groups in one workgroup
        const bool emit_gs = cand && emit_allowed && rw && gs;
        const bool emit = cand && emit_allowed && rw && ((!gs && ngroups == 1 && nw == 4 && sched == 0) || emit_gs);
        if (rw) {
            // (rows scaled before the conversion: the XS instantiations, non-temporal rows only)
            const bool xs = km->filter_rows_scaled;
            if (nw == 8) rwk = gs ? (xs ? k_assign_f16_rw<true, 8, true, 3, 2, 0, true> : nt_eff ? k_assign_f16_rw<true, 8, true, 3, 2> : k_assign_f16_rw<false, 8, true, 3, 2>)
                                  : (xs ? k_assign_f16_rw<true, 8, false, 3, 2, 0, true> : nt_eff ? k_assign_f16_rw<true, 8, false, 3, 2> : k_assign_f16_rw<false, 8, false, 3, 2>);
            else if (gs) rwk = xs ? k_assign_f16_rw<true, 4, true, 2, 0, 0, true> : nt_eff ? k_assign_f16_rw<true, 4, true, 2, 0> : k_assign_f16_rw<false, 4, true, 2, 0>;
            else if (sched == 2) rwk = xs ? k_assign_f16_rw<true, 4, false, 2, 2, 0, true> : nt_eff ? k_assign_f16_rw<true, 4, false, 2, 2> : k_assign_f16_rw<false, 4, false, 2, 2>;
            else if (emit && emit_inplace && !emit_gs)
                rwk = xs ? k_assign_f16_rw<true, 4, false, 2, 0, 2, true> : nt_eff ? k_assign_f16_rw<true, 4, false, 2, 0, 2> : k_assign_f16_rw<false, 4, false, 2, 0, 2>;
            else rwk = xs ? k_assign_f16_rw<true, 4, false, 2, 0, 0, true> : nt_eff ? k_assign_f16_rw<true, 4, false, 2, 0> : k_assign_f16_rw<false, 4, false, 2, 0>;
        }
        const int fsmem = FD_DX * nw * 4096 + dcr * FD_SLOT;
        const int64_t tile_rows = (int64_t)nw * 32, ntiles = (n + tile_rows - 1) / tile_rows;
        const int64_t fgrid = gs ? (ntiles + 7) / 8 * 8 * ngroups : ntiles;
        ACAV_REQUIRE(fgrid <= 0x7fffffff, ACAV_EINVAL, "n too large for one launch");
        if (gs) ACAV_TRY(km->grec.ensure(sizeof(Top2Rec) * (size_t)ngroups * (size_t)n));
        if (!km->ev_f0) {
            ACAV_HIP_TRY(hipEventCreate(&km->ev_f0));
            ACAV_HIP_TRY(hipEventCreate(&km->ev_f1));
        }
        if (rw) {
            ACAV_TRY(dyn_lds_once(reinterpret_cast<const void *>(rwk), km->ctx.device, fsmem));
            ACAV_HIP_TRY(hipEventRecord(km->ev_f0, st));
            hipLaunchKernelGGL(rwk, dim3((unsigned)fgrid), dim3(nw * 64), fsmem, st, static_cast<const float *>(dx), n, fd,
                               km->cb16.as<fl16>(), km->cn.as<float>(), km->counts.as<float>(), km->K, km->threshold(),
                               (float)km->reinit_r, km->caux.as<CentersAux>(), e1c, e1r, e2, dlab, emit ? und_list : f32_list,
                               emit ? &ctl->und_count : f32_count, ctl, gs ? km->grec.as<Top2Rec>() : (Top2Rec *)nullptr, cout);
            if (gs)
                hipLaunchKernelGGL(k_assign_merge, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, km->grec.as<Top2Rec>(),
                                   ngroups, n, km->caux.as<CentersAux>(), e1c, e1r, e2, dlab, emit_gs ? und_list : f32_list,
                                   emit_gs ? &ctl->und_count : f32_count, emit_gs ? cout.und_T : (float *)nullptr);
        }
        ACAV_HIP_TRY(hipGetLastError());
        ACAV_HIP_TRY(hipEventRecord(km->ev_f1, st));
        // exact pass over the listed rows (no host round trip): a fixed grid of 2 workgroups per CU strides over
        // however many row tiles the list turns out to hold
        if (km->num_cus == 0) {
            hipDeviceProp_t prop;
            ACAV_HIP_TRY(hipGetDeviceProperties(&prop, km->ctx.device));
            km->num_cus = prop.multiProcessorCount;
        }
        const int64_t rgrid = grid < 2 * (int64_t)km->num_cus ? grid : 2 * (int64_t)km->num_cus;
        if (emit) {
            // emission pass: the filter's main loop once more over the undecided rows only (a fixed grid strides over the tiles
            // of the list), the epilogue emits each row's candidate centres; then the exact canonical chains of those (row,
            // centre) pairs and the labels of those rows
            if (!emit_inplace || emit_gs) {
                // one workgroup per CU (rings + lists do not fit twice), 4 waves / 128 rows
                // (round 6: the 8-wave emission instantiations are gone -- they kept 156-172 B of scratch and no run ever selected them
                // outside A/B tests; the emission pass always runs 128-row tiles)
                constexpr int enw = 4;
                const bool xs = km->filter_rows_scaled;
                FilterKern ek = xs ? k_assign_f16_rw<true, 4, false, 2, 0, 1, true> : nt ? k_assign_f16_rw<true, 4, false, 2, 0, 1> : k_assign_f16_rw<false, 4, false, 2, 0, 1>;
                const int esmem = FD_DX * enw * 4096 + 2 * FD_SLOT + enw * 32 * (4 + 2 * (int)CAND_MAX);  // rings + lists
                ACAV_TRY(dyn_lds_once(reinterpret_cast<const void *>(ek), km->ctx.device, esmem));
                const int64_t erows = (int64_t)enw * 32;
                const int64_t egrid = std::min<int64_t>((n + erows - 1) / erows, (int64_t)km->num_cus);
                hipLaunchKernelGGL(ek, dim3((unsigned)egrid), dim3(enw * 64), esmem, st, static_cast<const float *>(dx), n, fd,
                                   km->cb16.as<fl16>(), km->cn.as<float>(), km->counts.as<float>(), km->K, km->threshold(),
                                   (float)km->reinit_r, km->caux.as<CentersAux>(), e1c, e1r, e2, dlab, und_list, &ctl->und_count, ctl,
                                   (Top2Rec *)nullptr, cout);
            }
            // (3 workgroups of 4 waves per CU: the kernel is bound by the L2 -> L1 path -- 1, 2, 3, 4, 6 per CU all measure the same)
            hipLaunchKernelGGL(k_assign_cand, dim3((unsigned)(3 * km->num_cus)), dim3(256), 0, st, static_cast<const float *>(dx),
                               fd, fc, km->cn.as<float>(), km->counts.as<float>(), km->threshold(),
                               (float)km->reinit_r, ctl, crow, cpair, pair_cap, dlab);
        }
        // full exact sweep of the rows on the f32 list (more than CAND_MAX candidates, pool overflow, K > 256); also the
        // sweep's last kernel: its last workgroup resets the control block
        hipLaunchKernelGGL(k_assign_f32<false>, dim3((unsigned)rgrid), dim3(256), 0, st, static_cast<const float *>(dx), n,
                           fd, fc, km->cn.as<float>(), km->counts.as<float>(), km->K,
                           km->threshold(), (float)km->reinit_r, dlab, (float *)nullptr, km->wg_sum.as<double>(),
                           f32_list, f32_count);
        km->n_filter_launches += 1;
        km->last_rows = (uint64_t)n;
    } else if (fast)
        hipLaunchKernelGGL(k_assign_f32<false>, dim3((unsigned)grid), dim3(256), 0, st, static_cast<const float *>(dx), n,
                           km->d, km->centers.as<float>(), km->cn.as<float>(), km->counts.as<float>(), km->K,
                           km->threshold(), (float)km->reinit_r, dlab, (float *)nullptr, km->wg_sum.as<double>(),
                           (const int *)nullptr, (const unsigned *)nullptr);
    else
        hipLaunchKernelGGL(k_assign_f32<true>, dim3((unsigned)grid), dim3(256), 0, st, static_cast<const float *>(dx), n,
                           km->d, km->centers.as<float>(), km->cn.as<float>(), km->counts.as<float>(), km->K,
                           km->threshold(), (float)km->reinit_r, dlab, (float *)nullptr, km->wg_sum.as<double>(),
                           (const int *)nullptr, (const unsigned *)nullptr);
    ACAV_HIP_TRY(hipGetLastError());
    km->n_assign_launches += 1;
    if (!lab_dev) ACAV_HIP_TRY(hipMemcpyAsync(labels, dlab, sizeof(int64_t) * (size_t)n, hipMemcpyDeviceToHost, st));
    if (mean_dist) {
        std::vector<double> part((size_t)grid);
        ACAV_HIP_TRY(hipMemcpyAsync(part.data(), km->wg_sum.p, sizeof(double) * (size_t)grid, hipMemcpyDeviceToHost, st));
        ACAV_HIP_TRY(hipStreamSynchronize(st));
        double s = 0.0;
        for (double v : part) s += v;
        *mean_dist = (float)(s / (double)n);
    } else if (!lab_dev || dx != x) {
        ACAV_HIP_TRY(hipStreamSynchronize(st));
    }
    return ACAV_OK;
}
