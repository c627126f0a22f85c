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
//   dot = one sequential fp32 FMA chain over j (what the f32 MFMA computes, k-ordered);
//   sumsq = 32 interleaved FMA chains (class = j mod 32) + fixed butterfly tree;
//   every other op is a single correctly-rounded fp32 op, compiled with -ffp-contract=off.
#include <cmath>

#include "acav_common.h"

using namespace acav;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

// ------------------------------------------------------------------------------- helpers
__device__ __forceinline__ float4 ld4_guard(const float *row, int j, int d, bool row_ok, bool vec_ok)
{
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!row_ok) return v;
    if (vec_ok && j + 3 < d) return *reinterpret_cast<const float4 *>(row + j);
    if (j < d) v.x = row[j];
    if (j + 1 < d) v.y = row[j + 1];
    if (j + 2 < d) v.z = row[j + 2];
    if (j + 3 < d) v.w = row[j + 3];
    return v;
}

// lexicographic (value, index) minimum: smaller value wins, ties -> smaller index (torch.min first index)
__device__ __forceinline__ void lexmin(float &bv, int &bi, float v, int i)
{
    if (v < bv || (v == bv && i < bi)) {
        bv = v;
        bi = i;
    }
}

__device__ __forceinline__ float norm2_from_sumsq(float ss)
{
    const float s = __builtin_sqrtf(ss);  // correctly rounded (-fhip-fp32-correctly-rounded-divide-sqrt)
    return s * s;
}

// one distance of the reference's calc_best (:72-77)
__device__ __forceinline__ float dist_epilogue(float dot, float xn, float cn, bool discount, float r)
{
    float t = -2.0f * dot;  // exact
    t = t + xn;
    t = t + cn;
    if (discount) t = t / r;
    return t;
}

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

// --------------------------------------------------------------------------- k_assign_f32
constexpr int AS_ROWS = 64;   // rows per workgroup: 2 MFMA row tiles
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

__global__ __launch_bounds__(256, 2) void k_assign_f32(const float *__restrict__ x, int64_t n, int d,
                                                       const float *__restrict__ centers,
                                                       const float *__restrict__ cn,
                                                       const float *__restrict__ counts, int K, float thr,
                                                       float r, int64_t *__restrict__ labels,
                                                       float *__restrict__ minval_out,
                                                       double *__restrict__ wg_sum)
{
    __shared__ __attribute__((aligned(16))) float sC[AS_CG * AS_LD];
    __shared__ __attribute__((aligned(16))) float sX[AS_ROWS * AS_LD];
    __shared__ float sXn[AS_ROWS];
    __shared__ float sMinV[4][AS_ROWS];
    __shared__ int sMinI[4][AS_ROWS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * AS_ROWS;
    const bool vec_ok = (d & 3) == 0;

    // staging roles: 8 threads per row (q = float4 index inside the 32-column stage)
    const int srow = tid >> 3;  // 0..31
    const int sq = tid & 7;
    const int nchunks = (d + AS_BK - 1) / AS_BK;
    const int ngroups = (K + AS_CG - 1) / AS_CG;

    float ssq[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float gbv = INFINITY;  // running best of row (tid) across centre groups, threads 0..63
    int gbi = 0x7fffffff;

    for (int cg = 0; cg < ngroups; ++cg) {
        const int kbase = cg * AS_CG;
        f32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

        float4 xr[2], cr[8];
        auto issue_loads = [&](int c) {
            const int j = c * AS_BK + sq * 4;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int64_t gr = row0 + srow + 32 * m;
                xr[m] = ld4_guard(x + (size_t)gr * d, j, d, gr < n, vec_ok);
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int k = kbase + srow + 32 * m;
                cr[m] = ld4_guard(centers + (size_t)k * d, j, d, k < K, vec_ok);
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
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const float xn = sXn[rt * 32 + l31];
            float bv = INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = kbase + (2 * wave + ct) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (k < K) {
                        const float t = dist_epilogue(acc[ct][rt][e], xn, cn[k], counts[k] < thr, r);
                        lexmin(bv, bi, t, k);
                    }
                }
            }
            const float ov = __shfl_xor(bv, 32);
            const int oi = __shfl_xor(bi, 32);
            lexmin(bv, bi, ov, oi);
            if (h == 0) {
                sMinV[wave][rt * 32 + l31] = bv;
                sMinI[wave][rt * 32 + l31] = bi;
            }
        }
        __syncthreads();
        if (tid < AS_ROWS) {
#pragma unroll
            for (int w = 0; w < 4; ++w) lexmin(gbv, gbi, sMinV[w][tid], sMinI[w][tid]);
        }
        // the next group's first __syncthreads orders these reads before sMin* is rewritten
    }

    if (tid < AS_ROWS) {  // wave 0
        const bool ok = row0 + tid < n;
        if (ok) {
            labels[row0 + tid] = (int64_t)gbi;
            if (minval_out) minval_out[row0 + tid] = gbv;
        }
        double s = ok ? (double)gbv : 0.0;
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (tid == 0) wg_sum[blockIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------- k_step_dist
// grid (ceil(K/32), ceil(b/32)); 4 waves, each one 16x16 tile via v_mfma_f32_16x16x4_f32
// (32-cycle issue, 40-cycle dependent latency: the shortest exact-fp32 chain on the chip).
constexpr int SD_BK = 64;
constexpr int SD_LD = 66;  // ds_read_b32 of [i][4t+g]: bank = 2i+g (+4t) -> conflict-free per 32 lanes

__global__ __launch_bounds__(256) void k_step_dist(const float *__restrict__ x, int b, int d,
                                                   const float *__restrict__ centers,
                                                   const float *__restrict__ cn,
                                                   const float *__restrict__ counts, int K, float thr,
                                                   float r, float *__restrict__ part_v,
                                                   int *__restrict__ part_i)
{
    __shared__ __attribute__((aligned(16))) float sC[32 * SD_LD];
    __shared__ __attribute__((aligned(16))) float sX[32 * SD_LD];
    __shared__ float sXn[32];
    __shared__ float sMinV[2][32];
    __shared__ int sMinI[2][32];

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

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float ssq[4] = {0.f, 0.f, 0.f, 0.f};
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
            ssq[0] = __builtin_fmaf(xr[m].x, xr[m].x, ssq[0]);
            ssq[1] = __builtin_fmaf(xr[m].y, xr[m].y, ssq[1]);
            ssq[2] = __builtin_fmaf(xr[m].z, xr[m].z, ssq[2]);
            ssq[3] = __builtin_fmaf(xr[m].w, xr[m].w, ssq[3]);
        }
        __syncthreads();
        if (c + 1 < nchunks) issue_loads(c + 1);
        const float *pa = sC + (cw * 16 + l15) * SD_LD + g;
        const float *pb = sX + (rw * 16 + l15) * SD_LD + g;
#pragma unroll
        for (int t = 0; t < SD_BK / 4; ++t)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[4 * t], pb[4 * t], acc, 0, 0, 0);
    }
    {
        float t = (ssq[0] + ssq[1]) + (ssq[2] + ssq[3]);
        t = t + __shfl_xor(t, 1);
        t = t + __shfl_xor(t, 2);
        t = t + __shfl_xor(t, 4);
        if (sq == 0) sXn[srow] = norm2_from_sumsq(t);
    }
    __syncthreads();
    // D[i][j]: column j = lane&15 (row of x), row i = 4*(lane>>4) + reg (centre)
    const float xn = sXn[rw * 16 + l15];
    float bv = INFINITY;
    int bi = 0x7fffffff;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = kbase + cw * 16 + 4 * g + e;
        if (k < K) {
            const float t = dist_epilogue(acc[e], xn, cn[k], counts[k] < thr, r);
            lexmin(bv, bi, t, k);
        }
    }
    {
        float ov = __shfl_xor(bv, 16);
        int oi = __shfl_xor(bi, 16);
        lexmin(bv, bi, ov, oi);
        ov = __shfl_xor(bv, 32);
        oi = __shfl_xor(bi, 32);
        lexmin(bv, bi, ov, oi);
    }
    if (g == 0) {
        sMinV[cw][rw * 16 + l15] = bv;
        sMinI[cw][rw * 16 + l15] = bi;
    }
    __syncthreads();
    if (tid < 32 && rbase + tid < b) {
        float v = sMinV[0][tid];
        int i = sMinI[0][tid];
        lexmin(v, i, sMinV[1][tid], sMinI[1][tid]);
        part_v[(size_t)blockIdx.x * b + rbase + tid] = v;
        part_i[(size_t)blockIdx.x * b + rbase + tid] = i;
    }
}

// -------------------------------------------------------------------------- k_step_update
// grid = b blocks.  Block i owns centre best[i] iff no earlier row of the batch has the same
// label; it then applies   c <- c*(1 - n_k*lr) + sum_{rows of the batch with label k, in batch
// order} lr*x   (sgd_clustering.py:120-127; the sum order is torch_scatter's CPU order) and
// refreshes ||c||^2.  Block 0 also folds the batch histogram into counts and handles the
// lr fallback bookkeeping (:116-119).
struct StepScalars {
    long long fallback;  // self.fallback
    float mean;          // return value of add(): mean of the row minima
    float lr_used;       // the (possibly fallen-back) fp32 lr of the last step
};

constexpr int SU_MAXB = 1024;

__global__ __launch_bounds__(256) void k_step_update(const float *__restrict__ x, int b, int d,
                                                     float *__restrict__ centers, float *__restrict__ cn,
                                                     float *__restrict__ counts, int K,
                                                     const float *__restrict__ part_v,
                                                     const int *__restrict__ part_i, int nparts,
                                                     const int64_t *__restrict__ forced, double lr,
                                                     StepScalars *__restrict__ sc, float forced_mean)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int *sBest = reinterpret_cast<int *>(smem_raw);            // [b]
    float *sMin = reinterpret_cast<float *>(sBest + SU_MAXB);  // [b]
    float *sRow = sMin + SU_MAXB;                              // [d]
    __shared__ int sCmax;
    const int tid = threadIdx.x;
    const int me = blockIdx.x;
    if (tid == 0) sCmax = 0;
    for (int i = tid; i < b; i += blockDim.x) {
        if (forced) {
            sBest[i] = (int)forced[i];
            sMin[i] = 0.f;
        } else {
            float v = part_v[i];
            int k = part_i[i];
            for (int p = 1; p < nparts; ++p) lexmin(v, k, part_v[(size_t)p * b + i], part_i[(size_t)p * b + i]);
            sBest[i] = k;
            sMin[i] = v;
        }
    }
    __syncthreads();
    const int mine = sBest[me];
    // my centre's batch count, and the batch maximum (each block recomputes: b is small)
    int cnt_mine = 0;
    bool first = true;
    for (int i = 0; i < b; ++i) {
        const bool same = sBest[i] == mine;
        cnt_mine += same;
        if (same && i < me) first = false;
    }
    // max count over centres = max over rows of (count of that row's label)
    int local_max = 0;
    for (int i = tid; i < b; i += blockDim.x) {
        int c = 0;
        const int ki = sBest[i];
        for (int i2 = 0; i2 < b; ++i2) c += (sBest[i2] == ki);
        local_max = max(local_max, c);
    }
    atomicMax(&sCmax, local_max);
    __syncthreads();
    const float cmax = (float)sCmax;
    bool fell = false;
    if ((double)cmax * lr >= 1.0) {  // Python float64 comparison (:116)
        lr = 0.5 / (double)cmax;
        fell = true;
    }
    const float lr32 = (float)lr;

    if (me == 0) {
        // counts += histogram (exact small integers in fp32), one thread per first-occurrence row
        for (int i = tid; i < b; i += blockDim.x) {
            const int ki = sBest[i];
            bool f = true;
            int c = 0;
            for (int i2 = 0; i2 < b; ++i2) {
                const bool same = sBest[i2] == ki;
                c += same;
                if (same && i2 < i) f = false;
            }
            if (f) counts[ki] = counts[ki] + (float)c;
        }
        if (tid == 0) {
            if (fell) sc->fallback += 1;
            sc->lr_used = lr32;
            if (forced) {
                sc->mean = forced_mean;
            } else {
                double s = 0.0;
                for (int i = 0; i < b; ++i) s += (double)sMin[i];
                sc->mean = (float)(s / (double)b);
            }
        }
    }
    if (!first) return;  // uniform per block

    const float f = 1.0f - (float)cnt_mine * lr32;
    float *crow = centers + (size_t)mine * d;
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
    __syncthreads();
    if (tid < 32) {
        float p = 0.f;
        for (int j = tid; j < d; j += 32) p = __builtin_fmaf(sRow[j], sRow[j], p);
        p = p + __shfl_xor(p, 1);
        p = p + __shfl_xor(p, 2);
        p = p + __shfl_xor(p, 4);
        p = p + __shfl_xor(p, 8);
        p = p + __shfl_xor(p, 16);
        if (tid == 0) cn[mine] = norm2_from_sumsq(p);
    }
}

}  // namespace

// =============================================================================== handle
struct acav_kmeans {
    StreamCtx ctx;
    int K = 0, d = 0;
    int initial_rounds = 10;
    double reinit_p = 0.7, reinit_r = 5.0;
    int64_t count = 0;  // python int self.count (deterministic on the host)
    DevBuf centers, cn, counts, scalars;
    DevBuf stage_x, stage_lab, stage_forced, part_v, part_i, wg_sum, minval;
    int64_t n_assign_launches = 0, n_step_launches = 0;

    float threshold() const { return (float)pow((double)count / (double)K, reinit_p); }
    bool warm() const { return count < (int64_t)initial_rounds * K; }
    int refresh_cn()
    {
        hipLaunchKernelGGL(k_row_norm2, dim3((K + 7) / 8), dim3(256), 0, ctx.stream, centers.as<float>(), K, d,
                           cn.as<float>());
        ACAV_HIP_TRY(hipGetLastError());
        return ACAV_OK;
    }
};

ACAV_EXPORT int acav_kmeans_create(acav_kmeans **out, int device, int k, int d, const float *centers0,
                                   void *stream)
{
    ACAV_REQUIRE(out && centers0, ACAV_EINVAL, "NULL argument");
    ACAV_REQUIRE(k > 0 && d > 0, ACAV_EINVAL, "k and d must be positive (k=%d d=%d)", k, d);
    ACAV_REQUIRE(d <= 16384, ACAV_EINVAL, "d=%d exceeds the supported 16384", d);
    acav_kmeans *km = new (std::nothrow) acav_kmeans;
    ACAV_REQUIRE(km, ACAV_ENOMEM, "out of host memory");
    int rc = km->ctx.init(device, stream);
    if (rc != ACAV_OK) {
        delete km;
        return rc;
    }
    km->K = k;
    km->d = d;
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
    km->ctx.fini();
    delete km;
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
    return ACAV_OK;
}

ACAV_EXPORT int acav_kmeans_assign(acav_kmeans *km, const float *x, int64_t n, int64_t *labels, float *mean_dist)
{
    ACAV_REQUIRE(km && x && labels, ACAV_EINVAL, "NULL argument");
    ACAV_REQUIRE(n >= 0, ACAV_EINVAL, "n must be >= 0");
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
    hipLaunchKernelGGL(k_assign_f32, dim3((unsigned)grid), dim3(256), 0, st, static_cast<const float *>(dx), n,
                       km->d, km->centers.as<float>(), km->cn.as<float>(), km->counts.as<float>(), km->K,
                       km->threshold(), (float)km->reinit_r, dlab, (float *)nullptr, km->wg_sum.as<double>());
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

// one add() on device-resident x [b,d]; forced (device) optional
static int step_device(acav_kmeans *km, const float *dx, int64_t b, double lr, const int64_t *dforced,
                       float forced_mean)
{
    ACAV_REQUIRE(b > 0 && b <= SU_MAXB, ACAV_EINVAL, "batch size %lld outside 1..%d", (long long)b, SU_MAXB);
    hipStream_t st = km->ctx.stream;
    const int nct = (km->K + 31) / 32;
    if (!dforced) {
        ACAV_REQUIRE(!km->warm(), ACAV_ESTATE, "warm-up step needs forced labels (acav_rng_warmup_best)");
        ACAV_TRY(km->part_v.ensure(sizeof(float) * (size_t)nct * b));
        ACAV_TRY(km->part_i.ensure(sizeof(int) * (size_t)nct * b));
        hipLaunchKernelGGL(k_step_dist, dim3(nct, (unsigned)((b + 31) / 32)), dim3(256), 0, st, dx, (int)b, km->d,
                           km->centers.as<float>(), km->cn.as<float>(), km->counts.as<float>(), km->K,
                           km->threshold(), (float)km->reinit_r, km->part_v.as<float>(), km->part_i.as<int>());
        ACAV_HIP_TRY(hipGetLastError());
    }
    const size_t smem = sizeof(int) * SU_MAXB + sizeof(float) * SU_MAXB + sizeof(float) * (size_t)km->d;
    hipLaunchKernelGGL(k_step_update, dim3((unsigned)b), dim3(256), smem, st, dx, (int)b, km->d,
                       km->centers.as<float>(), km->cn.as<float>(), km->counts.as<float>(), km->K,
                       km->part_v.as<float>(), km->part_i.as<int>(), nct, dforced, lr,
                       km->scalars.as<StepScalars>(), forced_mean);
    ACAV_HIP_TRY(hipGetLastError());
    km->count += b;
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
    ACAV_TRY(step_device(km, static_cast<const float *>(dx), b, lr, static_cast<const int64_t *>(df),
                         mean_dist ? *mean_dist : 0.f));
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

ACAV_EXPORT int acav_kmeans_train(acav_kmeans *km, const float *x, int64_t n, int64_t b, double lr,
                                  const int64_t *warm_best, int64_t n_warm)
{
    ACAV_REQUIRE(km && x, ACAV_EINVAL, "NULL argument");
    ACAV_REQUIRE(n >= 0 && b > 0 && n_warm >= 0, ACAV_EINVAL, "bad sizes");
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
    const void *dx = nullptr, *dw = nullptr;
    ACAV_TRY(to_device(x, sizeof(float) * (size_t)n * km->d, km->stage_x, st, &dx));
    if (need) ACAV_TRY(to_device(warm_best, sizeof(int64_t) * (size_t)need * b, km->stage_forced, st, &dw));
    const float *fx = static_cast<const float *>(dx);
    for (int64_t t = 0; t < steps; ++t) {
        const int64_t *f = t < need ? static_cast<const int64_t *>(dw) + t * b : nullptr;
        ACAV_TRY(step_device(km, fx + (size_t)t * b * km->d, b, lr, f, 0.f));
    }
    if (dx != x || (need && dw != warm_best)) ACAV_HIP_TRY(hipStreamSynchronize(st));
    return ACAV_OK;
}
