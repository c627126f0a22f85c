// acav_kmeans_shared.h -- what the two translation units of the k-means path share: the exact-arithmetic helpers
// (canonical fp32 contract with oracle/acav_oracle.c) and the handle.  acav_kmeans.hip holds the SGD training side
// (KMeans.add, sgd_clustering.py:94-129), acav_kmeans_assign.hip the assign sweep (KMeans.calc_best, :63-79).
#pragma once
#include <cmath>
#include <cstdlib>

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

constexpr int AS_ROWS = 64;   // rows per workgroup of the exact sweep: 2 MFMA row tiles

}  // namespace

// =============================================================================== handle
struct acav_kmeans {
    StreamCtx ctx;
    int K = 0, d = 0;
    int initial_rounds = 10;
    double reinit_p = 0.7, reinit_r = 5.0;
    int64_t count = 0;  // python int self.count (deterministic on the host)
    DevBuf centers, cn, counts, scalars;
    DevBuf stage_x, stage_lab, stage_forced, keys, xn, wg_sum, minval, thr, ctl;
    DevBuf cb16, caux, cmu, recheck_list, backup, grec, split_rings;
    DevBuf cand_ctl, cand_rows, cand_pairs, cand_T;
    DevBuf cpad, xpad;  // zero-padded centres / rows of the assign filter when d % 32 != 0 (acav_kmeans_assign.hip)
    unsigned ctl_pair_cap = 0;  // candidate-restricted re-check of the assign sweep (acav_kmeans_assign.hip)
    hipEvent_t ev_f0 = nullptr, ev_f1 = nullptr;  // around the last filter launch (acav_kmeans_filter_time)
    bool cb16_valid = false;  // the filter's half-precision copy of the centres matches `centers`
    bool filter_rows_scaled = false;  // ... and its sweeps multiply the rows by aux->sx before the conversion (XS instantiations)
    bool rg_attr_set = false; // dynamic-LDS attribute of the large-batch distance kernels set
    int64_t n_filter_launches = 0;
    uint64_t last_recheck = 0, last_rows = 0;
    int64_t n_persistent_launches = 0, n_persistent_fallbacks = 0;
    int num_cus = 0;  // multiProcessorCount of the handle's device (queried on first use)
    int key_phase = 0;  // which half of `keys` the next step's distance kernel folds into
    int64_t n_assign_launches = 0, n_step_launches = 0;

    void bind_buffers()  // every buffer of the handle is only ever used on the handle's stream (DevBuf::ensure synchronises that one)
    {
        for (DevBuf *b : {&centers, &cn, &counts, &scalars, &stage_x, &stage_lab, &stage_forced, &keys, &xn, &wg_sum, &minval, &thr, &ctl,
                          &cb16, &caux, &cmu, &recheck_list, &backup, &grec, &split_rings, &cand_ctl, &cand_rows, &cand_pairs, &cand_T,
                          &cpad, &xpad})
            b->bind(ctx.stream);
    }
    float threshold() const { return (float)pow((double)count / (double)K, reinit_p); }
    bool warm() const { return count < (int64_t)initial_rounds * K; }
    int filter_d() const { return (d + 31) / 32 * 32; }  // the width the assign filter runs at (d padded to its 32-column stage)
    int refresh_cn();          // acav_kmeans.hip (k_row_norm2 over the centres)
    int prepare_filter();      // acav_kmeans_assign.hip: bf16 / centred copy of the centres for the filter
};
