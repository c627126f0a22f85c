/*
 * acav_hip.h -- C ABI of libacav_hip.so: the MI355X (gfx950) implementation of ACAV100M's
 * post-extraction curation hot path (SGD k-means + greedy batch-MI subset selection).
 *
 * This is the drop-in boundary of SURVEY.md section 8(b).  Every entry point below replaces a call the
 * reference makes into torch / torch_scatter / torch.distributed; the reference interface it
 * stands in for is cited as file:line under /root/reference.  The Python classes in
 * acav100m_amd/ (same names and signatures as the reference's KMeans / EfficientBatchMI)
 * bind these symbols through ctypes; INTEGRATION.md shows the binding a reference maintainer
 * would add.
 *
 * Conventions
 *   - plain C, no torch types.  Bulk pointers (x, labels, assignments ...) may be DEVICE
 *     pointers (e.g. tensor.data_ptr()) or HOST pointers; the library detects which
 *     (hipPointerGetAttributes) and stages host buffers itself.  Buffers are caller-owned.
 *   - every function returns 0 on success, <0 on error (ACAV_E*); acav_last_error() returns
 *     a thread-local message.  No C++ exception crosses the ABI.
 *   - a handle binds one HIP device and one stream (given at create, or library-created when
 *     NULL).  Calls are stream-ordered; *_sync() blocks.  Handles are not thread-safe,
 *     distinct handles are independent.
 *   - integer outputs (labels, selected ids) are int64 like the reference's LongTensors.
 */
#ifndef ACAV_HIP_H
#define ACAV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACAV_OK 0
#define ACAV_EINVAL (-1)   /* bad argument (mirrors the reference's assert / ValueError)      */
#define ACAV_EHIP (-2)     /* HIP runtime error                                              */
#define ACAV_ENOMEM (-3)   /* allocation failed                                              */
#define ACAV_ESTATE (-4)   /* call not valid in the handle's current state                   */
#define ACAV_ERANGE (-5)   /* the reference would raise here (e.g. topk with k > batch)      */

typedef struct acav_rng acav_rng;       /* torch's CPU generator (mt19937)                    */
typedef struct acav_kmeans acav_kmeans; /* one reference KMeans object                        */
typedef struct acav_mi acav_mi;         /* one reference EfficientBatchMI object              */

/* ------------------------------------------------------------------------------ library */
const char *acav_last_error(void);
int acav_version(void);
/* number of visible HIP devices; 0 when there is no GPU (the library still loads) */
int acav_device_count(int *count);
int acav_device_info(int device, char *name, int name_len, int *compute_units, int64_t *hbm_bytes);

/* ---------------------------------------------------------------------------------- rng
 * The reference draws from torch's default CPU generator on this path:
 *   torch.rand(k, d) * 1e-5            clustering/code/sgd_clustering.py:24   (centre init)
 *   torch.rand(k, b)                   clustering/code/sgd_clustering.py:68   (warm-up distances)
 *   torch.randperm(L)                  subset_selection/code/measures/batch.py:31
 * acav_rng reproduces that stream (MT19937 seeded as torch.manual_seed does). */
int acav_rng_create(acav_rng **out, uint32_t seed);
int acav_rng_destroy(acav_rng *rng);
int acav_rng_seed(acav_rng *rng, uint32_t seed);                     /* torch.manual_seed   */
int acav_rng_u32(acav_rng *rng, uint32_t *out);
int acav_rng_rand_f32(acav_rng *rng, float *out_host, int64_t n);    /* torch.rand(n)       */
int acav_rng_randperm(acav_rng *rng, int64_t n, int64_t *out_host);  /* torch.randperm(n)   */
int acav_rng_get_state(const acav_rng *rng, uint32_t *mt624, int *idx);
int acav_rng_set_state(acav_rng *rng, const uint32_t *mt624, int idx);
/* labels of the warm-up phase: argmin over k of torch.rand(k, b) per column, and the mean of
 * the minima (sgd_clustering.py:67-68,78-79).  Consumes k*b draws. */
int acav_rng_warmup_best(acav_rng *rng, int k, int64_t b, int64_t *best_host, float *mean_out);

/* ------------------------------------------------------------------------------- kmeans
 * KMeans(args, d, k) -- clustering/code/sgd_clustering.py:18-32.
 * centers0 (host or device, [k,d] fp32 row-major) is the initial torch.rand(k,d)*1e-5 (draw it
 * with acav_rng_rand_f32); counts start at 0, count = 0, fallback = 0, initial_rounds = 10,
 * reinit = (0.7, 5.0).  stream: a hipStream_t, or NULL for a library-owned stream. */
int acav_kmeans_create(acav_kmeans **out, int device, int k, int d, const float *centers0, void *stream);
int acav_kmeans_destroy(acav_kmeans *km);
/* KMeans.get_attrs / load_from_saves / .centers / .counts / .count / .fallback
 * (sgd_clustering.py:34-57).  NULL pointers are skipped. */
int acav_kmeans_get_state(acav_kmeans *km, float *centers, float *counts, int64_t *count, int64_t *fallback);
int acav_kmeans_set_state(acav_kmeans *km, const float *centers, const float *counts, int64_t count,
                          int64_t fallback);
int acav_kmeans_set_hyper(acav_kmeans *km, int initial_rounds, double reinit_p, double reinit_r);
/* KMeans.calc_best(batch)[0] for n rows of x [n,d] once the warm-up is over (count >=
 * initial_rounds*k): labels[n] int64, *mean_dist = mean of the row minima
 * (sgd_clustering.py:63-79; called per batch by process_batch._extract_batch :37-49 and as the
 * first half of add :111).  Labels do not depend on how n is split into batches.
 * mean_dist == NULL selects the HBM-bound path: bf16-MFMA filter + exact fp32 re-check of the rows whose
 * top-2 gap is below the proven error bound -- the labels are bit-identical to the exact path.
 * During the warm-up the caller uses acav_rng_warmup_best instead (ACAV_ESTATE here). */
int acav_kmeans_assign(acav_kmeans *km, const float *x, int64_t n, int64_t *labels, float *mean_dist);
/* KMeans.add(batch) -- one SGD step on the b rows of x with learning rate lr
 * (sgd_clustering.py:94-129, non-distributed "fast parallel update" branch).
 * forced_best != NULL: labels to use instead of calc_best (the warm-up phase, or labels the
 * caller already holds).  mean_dist may be NULL (no host sync then). */
int acav_kmeans_step(acav_kmeans *km, const float *x, int64_t b, double lr, const int64_t *forced_best,
                     float *mean_dist);
/* The train-loop body of run_clustering.py:229-241 for one clustering: floor(n/b) consecutive
 * add() calls over x [n,d] (drop_last), no host synchronisation in between.
 * warm_best [n_warm*b]: labels for the first n_warm steps (the steps taken while
 * count < initial_rounds*k; draw them with acav_rng_warmup_best in step order). */
int acav_kmeans_train(acav_kmeans *km, const float *x, int64_t n, int64_t b, double lr,
                      const int64_t *warm_best, int64_t n_warm);
/* The update half of add() on an already-labelled GLOBAL batch (sgd_clustering.py:113-128):
 * used by the multi-GPU driver after the per-rank labels/rows have been all-gathered. */
int acav_kmeans_apply_update(acav_kmeans *km, const float *x, int64_t b, const int64_t *best, double lr);
int acav_kmeans_sync(acav_kmeans *km);
/* HIP-event timing on the handle's stream (bench.py roofline figures). */
int acav_kmeans_timer_begin(acav_kmeans *km);
int acav_kmeans_timer_end(acav_kmeans *km, float *ms);
/* bf16-filter bookkeeping of acav_kmeans_assign (mean_dist == NULL path): number of filter launches, and for
 * the last one the rows labelled and how many of them needed the exact fp32 re-check. */
int acav_kmeans_filter_stats(acav_kmeans *km, int64_t *filter_launches, int64_t *rows, int64_t *rechecked);
/* duration (HIP events on the handle's stream) of the last k_assign_bf16 launch alone -- the kernel bench.py's
 * roofline is quoted on; blocks until that launch has finished. */
int acav_kmeans_filter_time(acav_kmeans *km, float *ms);
/* counters: number of kernel launches of the dominant kernels since create (bench bookkeeping) */
int acav_kmeans_stats(acav_kmeans *km, int64_t *assign_launches, int64_t *step_launches);

/* ----------------------------------------------------------------------------------- mi
 * EfficientBatchMI(assignments[V,D], ncentroids=C, batch_size=B, selection_size=k, ...)
 * followed by .init(pairs, candidates) -- subset_selection/code/measures/batch.py:10-27,
 * measures/mi.py:20-39.  assignments int64 [V,D] row-major, pairs int32 [P,2]
 * (pairing.py:5-41).  Tables are N[P,C,C], a[P,C], b[P,C], n (integer counts on the device;
 * the reference's float eps initialisation only keeps its logs finite). */
int acav_mi_create(acav_mi **out, int device, const int64_t *assignments, int64_t V, int D, int C,
                   const int32_t *pairs, int P, void *stream);
int acav_mi_destroy(acav_mi *mi);
/* add_samples / update_cache: cache += one-hots of ids (mi.py:127-148, batch.py:152-154,190-193) */
int acav_mi_add_samples(acav_mi *mi, const int64_t *ids, int64_t n);
/* calc_MI(get_last(sample_batch(ids))).mean(-1): the score of each of the B candidates if it
 * alone were added, averaged over the P pairs (batch.py:34-54,123-130,144; mi.py:85-98).
 * float64, relative error vs the reference's fp32 < 1e-6. */
int acav_mi_score_batch(acav_mi *mi, const int64_t *ids, int B, double *scores_host);
/* EfficientBatchMI.run_greedy (batch.py:195-260): candidates[L] as handed to .init
 * (run_greedy.py:44-48), start[ns] the start indices (added to the tables, not reported).
 * Per iteration: full in-place Fisher-Yates of the candidate list with the rng stream
 * (== candidate_ids[torch.randperm(L)]), score the first B, take the top k (ties: lower batch
 * position first), commit them, re-queue the unselected in ascending id order.
 * S_out/GAIN_out must hold subset + k entries; *n_selected = min(iters*k, subset) entries of
 * S_out are the selection (batch.py:258 cut), GAIN_out holds iters*k entries.
 * trace_ids [iters,B] / trace_scores [iters,B] / trace_pos [iters,k] optional (host).
 * forced_pos [iters,k] optional: commit these batch positions instead of the top-k (replays a
 * recorded run; used by the parity tests against the reference trace). */
int acav_mi_run_greedy(acav_mi *mi, const int64_t *candidates, int64_t L, const int64_t *start, int ns,
                       int64_t subset, int B, int k, int keep_unselected, acav_rng *rng, int64_t *S_out,
                       double *GAIN_out, int64_t *n_selected, int64_t *n_iters, int64_t *trace_ids,
                       double *trace_scores, int32_t *trace_pos, const int32_t *forced_pos, int64_t max_iters);
/* nchunks independent selections (chunk.py:21-53: one EfficientBatchMI + run_greedy per chunk) driven in lockstep
 * by ONE set of kernel launches per iteration -- the greedy loop of a single chunk is a chain of small dependent
 * kernels that leaves the GPU mostly idle.  Arrays are indexed by chunk; B, k and keep_unselected are shared.
 * Chunk c yields exactly what acav_mi_run_greedy(mis[c], candidates[c], L[c], start[c], ns[c], subset[c], B, k,
 * keep_unselected, rngs[c], ...) yields, and leaves rngs[c] where that call would.  All handles on one device, no
 * handle or generator twice.  (The reference runs the chunks of a process one after the other on ONE generator;
 * chunks in flight together need a generator each -- see subset_selection/run.py.) */
int acav_mi_run_greedy_multi(acav_mi **mis, int nchunks, const int64_t *const *candidates, const int64_t *L,
                             const int64_t *const *start, const int *ns, const int64_t *subset, int B, int k,
                             int keep_unselected, acav_rng **rngs, int64_t *const *S_out, double *const *GAIN_out,
                             int64_t *n_selected, int64_t *n_iters);
/* EfficientMI.run_greedy / EfficientMemMI (measures/mi.py:150-192; 'mi' and 'mem_mi' of measures/__init__.py:5-14):
 * the exact greedy.  Each of the subset - 1 - ns iterations scores ALL remaining candidates (mi.py:108-110),
 * commits the first maximum (scores.max(dim=0), mi.py:79) and removes it, keeping the order of the rest.  The ns
 * start indices are NOT added to the tables (mi.py never does; batch.py does) -- they only shorten the loop.
 * S_out / GAIN_out hold *n_selected = max(0, min(subset - 1 - ns, L)) entries (the picks after the start indices).
 * Optional (host): forced_pos [iters] = ORIGINAL positions (index into candidates) to commit instead of the argmax;
 * trace_scores [iters, L] = canonical scores by original position (NaN once removed); trace_argmax [iters]. */
int acav_mi_run_exact(acav_mi *mi, const int64_t *candidates, int64_t L, int ns, int64_t subset, int64_t *S_out,
                      double *GAIN_out, int64_t *n_selected, const int64_t *forced_pos, double *trace_scores,
                      int64_t *trace_argmax);
int acav_mi_get_counts(acav_mi *mi, int32_t *N, int32_t *a, int32_t *b, int64_t *n);
int acav_mi_sync(acav_mi *mi);
int acav_mi_timer_begin(acav_mi *mi);
int acav_mi_timer_end(acav_mi *mi, float *ms);

#ifdef __cplusplus
}
#endif
#endif /* ACAV_HIP_H */
