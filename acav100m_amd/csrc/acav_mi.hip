// acav_mi.hip -- greedy batch mutual-information subset selection on gfx950.
// Reference: subset_selection/code/measures/mi.py:14-148 (EfficientMI tables / calc_MI) and
// measures/batch.py:10-260 (EfficientBatchMI greedy loop).
//
// What one greedy iteration is in the reference, and what runs here instead:
//   shuffle_candidate_ids (batch.py:29-32)  candidate_ids[torch.randperm(L)]: L-1 MT19937 draws and a
//        sequential Fisher-Yates.  Here: k_mt_generate continues the same MT19937 stream on the
//        GPU, k_fy_build / k_fy_apply evaluate the SAME swap sequence in parallel (dependence
//        chains instead of a serial loop) -- integer work, bit-identical by construction.
//   sample_batch + get_last + calc_MI (batch.py:34-54, mi.py:85-98)  dense [B,P,C,C] fp32 tensors.
//        Here: O(1) delta-MI per (candidate, pair) from the integer contingency tables
//        (k_mi_select), float64, formula in oracle/acav_oracle.c "canon".
//   calc_ids top-k, update_cache, update_candidates (batch.py:143-171)  k_mi_select.
// The whole loop is enqueued without host synchronisation: L shrinks deterministically.
#include <chrono>
#include <cmath>

#include "acav_common.h"
#include <vector>

using namespace acav;

// Experiment builds only (-DACAV_EXPERIMENT_BUILD -DACAV_MI_ABL_EMPTY, tools/exp/build_empty_mi.sh): every kernel of the greedy
// loop returns at once, so that ACAV_MI_TIMING=1 shows what the HOST side of the loop costs per iteration with the same launch
// pattern (grids, streams, events, generator bookkeeping) and an idle GPU.  Results are garbage by construction.
#ifdef ACAV_MI_ABL_EMPTY
#define ACAV_MI_EMPTY_RETURN return;
#else
#define ACAV_MI_EMPTY_RETURN
#endif

namespace {

struct MiScalars {
    long long nc;  // n of the tables: number of samples added so far
};

#ifdef ACAV_FY_PROF  // tools/exp/fy_bench.hip: 100 MHz wall-clock ticks per phase, summed over workgroups
__device__ unsigned long long fy_prof[16];
#define FY_CLK(slot) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&fy_prof[slot], t_ - fy_t0); fy_t0 = t_; } } while (0)
#define FY_CLK0() unsigned long long fy_t0 = wall_clock64()
#else
#define FY_CLK(slot) do { } while (0)
#define FY_CLK0() do { } while (0)
#endif

// -------------------------------------------------------------------------- table kernels
// cache += one-hots of ids (mi.py:127-148): thread p owns pair p, ids applied in order so the
// running sums see the same sequence of float64 updates as the oracle.
__global__ __launch_bounds__(256) void k_mi_commit(const int *__restrict__ asg, int D, int C, int P,
                                                   const int *__restrict__ pairs, const int *__restrict__ ids,
                                                   int n, int *__restrict__ Nc, int *__restrict__ ac,
                                                   int *__restrict__ bc, double *__restrict__ SN,
                                                   double *__restrict__ Sa, double *__restrict__ Sb,
                                                   const double *__restrict__ phi, MiScalars *__restrict__ sc)
{
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        const int d0 = pairs[2 * p], d1 = pairs[2 * p + 1];
        double sN = SN[p], sa = Sa[p], sb = Sb[p];
        for (int w = 0; w < n; ++w) {
            const int *row = asg + (size_t)ids[w] * D;
            const int i = row[d0], j = row[d1];
            const size_t cell = ((size_t)p * C + i) * C + j;
            const int cN = Nc[cell], ca = ac[(size_t)p * C + j], cb = bc[(size_t)p * C + i];
            Nc[cell] = cN + 1;
            ac[(size_t)p * C + j] = ca + 1;
            bc[(size_t)p * C + i] = cb + 1;
            sN = sN - phi[cN] + phi[cN + 1];
            sa = sa - phi[ca] + phi[ca + 1];
            sb = sb - phi[cb] + phi[cb + 1];
        }
        SN[p] = sN;
        Sa[p] = sa;
        Sb[p] = sb;
    }
    if (threadIdx.x == 0) sc->nc += n;
}

// score of candidate `id` for pair p if it alone were added (canonical float64 closed form)
__device__ __forceinline__ double mi_pair_score(const int *__restrict__ asg, int D, int C, int p,
                                                const int *__restrict__ pairs, int id,
                                                const int *__restrict__ Nc, const int *__restrict__ ac,
                                                const int *__restrict__ bc, const double *__restrict__ SN,
                                                const double *__restrict__ Sa, const double *__restrict__ Sb,
                                                const double *__restrict__ phi, long long nc)
{
    const int *row = asg + (size_t)id * D;
    const int i = row[pairs[2 * p]], j = row[pairs[2 * p + 1]];
    const int cN = Nc[((size_t)p * C + i) * C + j];
    const int ca = ac[(size_t)p * C + j], cb = bc[(size_t)p * C + i];
    const double sN = SN[p] - phi[cN] + phi[cN + 1];
    const double sa = Sa[p] - phi[ca] + phi[ca + 1];
    const double sb = Sb[p] - phi[cb] + phi[cb + 1];
    return (((sN - sa) - sb) + phi[nc + 1]) / (double)(nc + 1);
}

// exp(x) of the `ami` score: the SAME sequence of IEEE double operations as oracle/acav_oracle.c canon_exp (the device's
// and libm's exp differ in the last bit); x <= ~0 (logs of probabilities)
__device__ __forceinline__ double canon_exp(double x)
{
    if (x < -745.0) return 0.0;
    const double k = rint(x * 1.4426950408889634074);
    double r = x - k * 6.93147180369123816490e-01;
    r = r - k * 1.90821492927058770002e-10;
    double p = 1.0 / 6227020800.0;
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    return ldexp(p, (int)k);
}

// EfficientAMI._calc_score (mi.py:212-259) of cache + candidate `id` for pair p: (MI - EMI) / max(mean entropy - EMI, eps)
// with the reference's one-term-per-cell EMI, over integer counts; every log / log-factorial is a look-up in the host-built
// tables lnk / lf (the oracle's doubles), the cells are walked row-major: the canonical form of oracle ami_score_canon.
__device__ double ami_pair_score(const int *__restrict__ asg, int D, int C, int p, const int *__restrict__ pairs, int id,
                                 const int *__restrict__ Nc, const int *__restrict__ ac, const int *__restrict__ bc,
                                 const double *__restrict__ SN, const double *__restrict__ Sa, const double *__restrict__ Sb,
                                 const double *__restrict__ phi, const double *__restrict__ lnk, const double *__restrict__ lf,
                                 long long nc)
{
    const int *row = asg + (size_t)id * D;
    const int i = row[pairs[2 * p]], j = row[pairs[2 * p + 1]];
    const int *Np = Nc + (size_t)p * C * C, *ap = ac + (size_t)p * C, *bp = bc + (size_t)p * C;
    const long long n1i = nc + 1;
    const double n1 = (double)n1i, ln_n = lnk[n1i];
    const int cN = Np[(size_t)i * C + j], ca = ap[j], cb = bp[i];
    const double sN = SN[p] - phi[cN] + phi[cN + 1];
    const double sa = Sa[p] - phi[ca] + phi[ca + 1];
    const double sb = Sb[p] - phi[cb] + phi[cb + 1];
    const double mi = (((sN - sa) - sb) + phi[n1i]) / n1;
    double emi = 0.0;
    for (int r = 0; r < C; ++r) {
        const long long b1 = bp[r] + (r == i);
        if (b1 == 0) continue;
        for (int c = 0; c < C; ++c) {
            const long long N1 = Np[(size_t)r * C + c] + (r == i && c == j);
            if (N1 == 0) continue;
            const long long a1 = ap[c] + (c == j);
            const double t1 = ((double)N1 / n1) * (((lnk[N1] + ln_n) - lnk[a1]) - lnk[b1]);
            double l2 = lf[a1] + lf[b1];
            l2 = l2 + lf[n1i - a1];
            l2 = l2 + lf[n1i - b1];
            l2 = l2 - lf[n1i];
            l2 = l2 - lf[N1];
            l2 = l2 - lf[a1 - N1];
            l2 = l2 - lf[b1 - N1];
            l2 = l2 - lf[n1i - a1 - b1 + N1];
            emi = emi + t1 * canon_exp(l2);
        }
    }
    const double ha = ln_n - sa / n1, hb = ln_n - sb / n1;
    double den = (ha + hb) / 2.0 - emi;
    if (den < 2.220446049250313e-16) den = 2.220446049250313e-16;
    return (mi - emi) / den;
}

// EfficientNMI._calc_score (mi.py:262-271) for pair p: 2 MI / max(mean entropy, eps) -- the MI and entropy terms of the
// adjusted score without the EMI; the canonical form of oracle nmi_score_canon (same running sums, same operations)
__device__ __forceinline__ double nmi_pair_score(const int *__restrict__ asg, int D, int C, int p, const int *__restrict__ pairs, int id,
                                                 const int *__restrict__ Nc, const int *__restrict__ ac, const int *__restrict__ bc,
                                                 const double *__restrict__ SN, const double *__restrict__ Sa, const double *__restrict__ Sb,
                                                 const double *__restrict__ phi, const double *__restrict__ lnk, long long nc)
{
    const int *row = asg + (size_t)id * D;
    const int i = row[pairs[2 * p]], j = row[pairs[2 * p + 1]];
    const long long n1i = nc + 1;
    const double n1 = (double)n1i, ln_n = lnk[n1i];
    const int cN = Nc[((size_t)p * C + i) * C + j], ca = ac[(size_t)p * C + j], cb = bc[(size_t)p * C + i];
    const double sN = SN[p] - phi[cN] + phi[cN + 1];
    const double sa = Sa[p] - phi[ca] + phi[ca + 1];
    const double sb = Sb[p] - phi[cb] + phi[cb + 1];
    if ((long long)cN + 1 == n1i) {
        // DEGENERATE (every sample in ONE cell: MI = entropies = 0 over integer counts): the reference returns the ratio of its eps
        // artefacts, a closed form of C and n -- derivation in oracle/acav_oracle.c nmi_score_canon; the same operations here
        const double ln_eps = -36.043653389117154, ln_c = lnk[C];
        const double num = (double)(C - 1) * ((ln_n - ln_eps) - 2.0 * ln_c) - 2.0 * ln_c;
        const double dd = (double)C * ((ln_n - ln_c) - ln_eps);
        return (2.0 * num) / dd;
    }
    const double mi = (((sN - sa) - sb) + phi[n1i]) / n1;
    const double ha = ln_n - sa / n1, hb = ln_n - sb / n1;
    double den = (ha + hb) / 2.0;
    if (den < 2.220446049250313e-16) den = 2.220446049250313e-16;  // ensure_nonzero (mi.py:194-199)
    return (2.0 * mi) / den;
}

constexpr int SEL_MAXB = 64;
constexpr int SEL_MAXBP = 8192;

// One workgroup per greedy iteration: score the B candidates (mean over P pairs), pick the top k
// (descending, ties -> lower batch position), commit them, emit S/GAIN, and append the
// unselected ids in ascending order to the new candidate array (batch.py:132-171).
// ids == batch_in when called for plain scoring (k = 0: nothing committed).
// The kernel sits on the critical path of every iteration and is pure latency -- a chain of dependent memory round
// trips (~1 us each, ~2 us beside a gather in full flight) -- so the chain is kept short:
//   * what does not depend on the batch (n, the pair list, the running sums) is staged in LDS by mi_select_stage before
//     anything else
//   * the batch arrives with its label rows (the gather that produced it fetched them: lab_in), or in registers
//   * scoring is then 2 dependent levels (labels -> counts -> phi); every phase issues all of its independent loads
//     before the first use
//   * the top-k is a rank count over the B scores in LDS (no reduction rounds); barriers order LDS only
//   * the commit re-uses the labels, counts and phi values the scoring read (LDS); only a pick whose cell an earlier pick
//     of the same iteration touched pays a level for the phi of its adjusted counts.  Beyond the staged sizes (SEL_FAST*)
//     it re-reads (3 levels per pick)
//   * few registers: the gather workgroups of the same launch inherit the kernel's allocation
constexpr int SEL_LDSP = 256;     // pairs (and their running sums) staged in LDS
constexpr int SEL_FASTBP = 2048;  // B * P up to which the scoring's labels / counts are kept for the commit ...
constexpr int SEL_FASTKP = 512;   // ... and k * P up to which the commit's phi values are staged (both: SEL_FAST)
constexpr int SEL_PHIBP = 256;    // B * P up to which the scoring's phi values are kept too (SEL_KEEPPHI)
constexpr int SEL_FAST = 1, SEL_KEEPPHI = 2;

struct SelShared {
    double score[SEL_MAXB];
    double SN[SEL_LDSP], Sa[SEL_LDSP], Sb[SEL_LDSP];
    long long nc;
    unsigned long long used;
    int id[SEL_MAXB], pos[SEL_MAXB], pick[SEL_MAXB];
    int pairs[2 * SEL_LDSP];
};

// Dynamic LDS of a selection launch:  scores [B P] f64 | labels + counts 5 x [B P] i32 (SEL_FAST) | scoring's phi [B P][6] f64
// (SEL_KEEPPHI) | commit's phi [k P][6] f64 (SEL_FAST) | label rows [B D] i32.  The mode is decided on the host for the
// largest P and D of the launch; every chunk lays its own sizes out in the same order.
struct SelLayout {
    unsigned off_c, off_phi, off_phik, off_lab, total;
};
__host__ __device__ inline int sel_mode(int B, int P, int k)
{
    const long long bp = (long long)B * P, kp = (long long)k * P;
    const int fast = bp <= SEL_FASTBP && kp <= SEL_FASTKP ? SEL_FAST : 0;
    return fast | (fast && bp <= SEL_PHIBP ? SEL_KEEPPHI : 0);
}
__host__ __device__ inline SelLayout sel_layout(int B, int P, int D, int k, int mode)
{
    const unsigned bp = (unsigned)B * (unsigned)P, kp = (unsigned)k * (unsigned)P;
    SelLayout l;
    l.off_c = 8u * bp;
    l.off_phi = l.off_c + ((mode & SEL_FAST) ? 20u * bp + 4u * (bp & 1u) : 0u);
    l.off_phik = l.off_phi + ((mode & SEL_KEEPPHI) ? 48u * bp : 0u);
    l.off_lab = l.off_phik + ((mode & SEL_FAST) ? 48u * kp : 0u);
    l.total = l.off_lab + 4u * (unsigned)B * (unsigned)D;
    return l;
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global store to be
// acknowledged (~1 us each time on the selection's critical path)
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// everything the selection needs that does not depend on the batch; a barrier must separate it from mi_select_body
__device__ __forceinline__ void mi_select_stage(SelShared &ss, int P, const int *__restrict__ pairs,
                                                const double *__restrict__ SN, const double *__restrict__ Sa,
                                                const double *__restrict__ Sb, const MiScalars *__restrict__ sc)
{
    const int tid = threadIdx.x;
    if (tid == 0) ss.nc = sc->nc;
    for (int p = tid; p < P && p < SEL_LDSP; p += blockDim.x) {
        ss.pairs[2 * p] = pairs[2 * p];
        ss.pairs[2 * p + 1] = pairs[2 * p + 1];
        ss.SN[p] = SN[p];
        ss.Sa[p] = Sa[p];
        ss.Sb[p] = Sb[p];
    }
}

// batch == nullptr: the id of batch position tid (< B) is in reg_id.  lab_in: the B label rows [B][D] when the caller has
// them (else they are read from asg: one more level).  mode: sel_mode() for the launch (decided on the host).
__device__ __forceinline__ void mi_select_body(
    SelShared &ss, const int *__restrict__ asg, int D, int C, int P, const int *__restrict__ pairs,
    const int *__restrict__ batch, int reg_id, const int *__restrict__ lab_in, int B, int k, int mode, int *__restrict__ Nc,
    int *__restrict__ ac, int *__restrict__ bc, double *__restrict__ SN, double *__restrict__ Sa, double *__restrict__ Sb,
    const double *__restrict__ phi, MiScalars *__restrict__ sc, double *__restrict__ scores_out,
    long long *__restrict__ S_out, double *__restrict__ G_out, const int *__restrict__ forced_pos,
    int *__restrict__ trace_pos, long long *__restrict__ trace_ids, double *__restrict__ trace_scores,
    int keep_unselected, int *__restrict__ requeue_out, int requeue_stride = 1)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const bool fast = (mode & SEL_FAST) != 0, keepphi = (mode & SEL_KEEPPHI) != 0;
    const SelLayout lay = sel_layout(B, P, D, k, mode);
    const size_t BP = (size_t)B * P;
    double *sS = reinterpret_cast<double *>(smem_raw);  // [B*P]
    int *sCi = reinterpret_cast<int *>(smem_raw + lay.off_c), *sCj = sCi + BP, *sCN = sCj + BP, *sCa = sCN + BP, *sCb = sCa + BP;
    double *sPhiAll = reinterpret_cast<double *>(smem_raw + lay.off_phi);  // [B*P][6]  (keepphi)
    double *sPhiK = reinterpret_cast<double *>(smem_raw + lay.off_phik);   // [k*P][6]  (fast)
    int *sLab = reinterpret_cast<int *>(smem_raw + lay.off_lab);           // [B][D]
    const int tid = threadIdx.x;
    FY_CLK0();
    if (tid < B) ss.id[tid] = batch ? batch[tid] : reg_id;
    if (lab_in)
        for (int t = tid; t < B * D; t += blockDim.x) sLab[t] = lab_in[t];
    lds_barrier();  // also: the staged constants are visible
    if (!lab_in) {
        for (int t = tid; t < B * D; t += blockDim.x) {
            const int w = t / D;
            sLab[t] = asg[(size_t)ss.id[w] * D + (t - w * D)];
        }
        lds_barrier();
    }
    FY_CLK(8);
    const long long nc = ss.nc;
    const double phin = phi[nc + 1];
    for (int t = tid; t < B * P; t += blockDim.x) {
        // score of candidate w for pair p if it alone were added (canonical float64 closed form)
        const int w = t / P, p = t - w * P;
        const bool lp = p < SEL_LDSP;
        const int d0 = lp ? ss.pairs[2 * p] : pairs[2 * p], d1 = lp ? ss.pairs[2 * p + 1] : pairs[2 * p + 1];
        const double sN0 = lp ? ss.SN[p] : SN[p], sa0 = lp ? ss.Sa[p] : Sa[p], sb0 = lp ? ss.Sb[p] : Sb[p];
        const int i = sLab[w * D + d0], j = sLab[w * D + d1];
        const int cN = Nc[((size_t)p * C + i) * C + j];
        const int ca = ac[(size_t)p * C + j], cb = bc[(size_t)p * C + i];
        const double f0 = phi[cN], f1 = phi[cN + 1], f2 = phi[ca], f3 = phi[ca + 1], f4 = phi[cb], f5 = phi[cb + 1];
        const double sN = sN0 - f0 + f1;
        const double sa = sa0 - f2 + f3;
        const double sb = sb0 - f4 + f5;
        sS[t] = (((sN - sa) - sb) + phin) / (double)(nc + 1);
        if (fast) sCi[t] = i, sCj[t] = j, sCN[t] = cN, sCa[t] = ca, sCb[t] = cb;
        if (keepphi) {
            double *o6 = sPhiAll + (size_t)t * 6;
            o6[0] = f0, o6[1] = f1, o6[2] = f2, o6[3] = f3, o6[4] = f4, o6[5] = f5;
        }
    }
    lds_barrier();
    FY_CLK(9);
    // Wave 0 (B <= 64), lane = batch position: mean score, then the top k by RANK -- every lane counts the candidates that
    // beat it, their keys broadcast lane by lane (v_readlane; k rounds of a wave-wide argmax were 2.6 us of shuffles, a
    // loop of LDS reads 1 us).  Order: score descending, ties -> lower batch position; a NaN score never wins against a
    // number and ranks by position among its like (as the argmax rounds did).  The un-selected ids are ranked the same way.
    if (tid < 64) {
        double s = 0.0, key = -INFINITY;
        const int myid = tid < B ? ss.id[tid] : 0;
        if (tid < B) {
            double tot = 0.0;
            for (int p = 0; p < P; ++p) tot = tot + sS[tid * P + p];
            s = tot / (double)P;
            key = s == s ? s : -INFINITY;
            ss.score[tid] = s;
            if (scores_out) scores_out[tid] = s;
            if (trace_scores) trace_scores[tid] = s;
            if (trace_ids) trace_ids[tid] = myid;
        }
        if (k > 0) {
            const int khi = __double2hiint(key), klo = __double2loint(key);
            int rank = 0;
            for (int w = 0; w < B; ++w) {
                const double o = __hiloint2double(__builtin_amdgcn_readlane(khi, w), __builtin_amdgcn_readlane(klo, w));
                rank += (o > key || (o == key && w < tid)) ? 1 : 0;
            }
            const bool picked = tid < B && rank < k;
            if (picked) {
                if (!forced_pos) ss.pos[rank] = tid;
                if (trace_pos) trace_pos[rank] = tid;  // the free-running choice, also under teacher forcing
            }
            unsigned long long used = __ballot(picked);
            if (forced_pos) {
                used = 0ull;
                for (int r = 0; r < k; ++r) used |= 1ull << forced_pos[r];
                if (tid < k) ss.pos[tid] = forced_pos[tid];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ss.pos / ss.score: written and read by this wave, in order
            if (tid < k) {
                const int pos = ss.pos[tid];
                ss.pick[tid] = ss.id[pos];
                S_out[tid] = (long long)ss.id[pos];
                G_out[tid] = ss.score[pos];
            }
            if (keep_unselected) {  // get_unselected: torch.unique -> ascending ids (batch.py:167-171)
                int rq = 0;
                for (int w = 0; w < B; ++w) {
                    const int o = __builtin_amdgcn_readlane(myid, w);
                    rq += (!(used >> w & 1ull) && (o < myid || (o == myid && w < tid))) ? 1 : 0;
                }
                if (tid < B && !(used >> tid & 1ull)) requeue_out[rq * requeue_stride] = myid;
            }
        }
    }
    FY_CLK(10);
    if (k == 0) return;
    lds_barrier();  // ss.pos, ss.pick; the global stores above need not have landed
    FY_CLK(11);
    // update_cache (batch.py:152-154): commit the k winners; per pair the picks are applied in pick order (the float64
    // running sums see the oracle's sequence of updates)
    if (fast) {
        // (pick, pair)-parallel: the labels and counts are the ones the scoring read (LDS), adjusted by the earlier picks
        // of this iteration; the phi values too unless the adjustment moved the count (then one round of loads); staged in
        // LDS, then one thread per pair applies them in order.  (Registers: an unrolled per-pair loop over the picks took
        // the kernel to 253 VGPRs -- 2 waves per SIMD for every gather workgroup of the same launch.)
        for (int t = tid; t < k * P; t += blockDim.x) {
            const int r = t / P, p = t - r * P;
            const int me = ss.pos[r] * P + p;
            const int ci = sCi[me], cj = sCj[me];
            const int bN = sCN[me], ba = sCa[me], bb = sCb[me];
            int cN = bN, ca = ba, cb = bb;
            bool lastN = true, lasta = true, lastb = true;  // no later pick of this iteration touches the same cell
#pragma unroll 4
            for (int e = 0; e < k; ++e) {
                if (e == r) continue;
                const int o = ss.pos[e] * P + p;
                const bool mi_ = sCi[o] == ci, mj = sCj[o] == cj;
                if (e < r) {
                    cN += (mi_ && mj) ? 1 : 0;
                    ca += mj ? 1 : 0;
                    cb += mi_ ? 1 : 0;
                } else {
                    lastN = lastN && !(mi_ && mj);
                    lasta = lasta && !mj;
                    lastb = lastb && !mi_;
                }
            }
            const double *q6 = sPhiAll + (size_t)me * 6;
            const bool hN = keepphi && cN == bN, ha = keepphi && ca == ba, hb = keepphi && cb == bb;
            const double f0 = hN ? q6[0] : phi[cN], f1 = hN ? q6[1] : phi[cN + 1];
            const double f2 = ha ? q6[2] : phi[ca], f3 = ha ? q6[3] : phi[ca + 1];
            const double f4 = hb ? q6[4] : phi[cb], f5 = hb ? q6[5] : phi[cb + 1];
            if (lastN) Nc[((size_t)p * C + ci) * C + cj] = cN + 1;
            if (lasta) ac[(size_t)p * C + cj] = ca + 1;
            if (lastb) bc[(size_t)p * C + ci] = cb + 1;
            double *o6 = sPhiK + (size_t)t * 6;
            o6[0] = f0, o6[1] = f1, o6[2] = f2, o6[3] = f3, o6[4] = f4, o6[5] = f5;
        }
        lds_barrier();
        for (int p = tid; p < P; p += blockDim.x) {
            const bool lp = p < SEL_LDSP;
            double sN = lp ? ss.SN[p] : SN[p], sa = lp ? ss.Sa[p] : Sa[p], sb = lp ? ss.Sb[p] : Sb[p];
#pragma unroll 4
            for (int r = 0; r < k; ++r) {
                const double *f = sPhiK + ((size_t)r * P + p) * 6;
                sN = sN - f[0] + f[1];
                sa = sa - f[2] + f[3];
                sb = sb - f[4] + f[5];
            }
            SN[p] = sN;
            Sa[p] = sa;
            Sb[p] = sb;
        }
    } else {
        for (int p = tid; p < P; p += blockDim.x) {  // one thread per pair, pick by pick
            const bool lp = p < SEL_LDSP;
            double sN = lp ? ss.SN[p] : SN[p], sa = lp ? ss.Sa[p] : Sa[p], sb = lp ? ss.Sb[p] : Sb[p];
            const int d0 = lp ? ss.pairs[2 * p] : pairs[2 * p], d1 = lp ? ss.pairs[2 * p + 1] : pairs[2 * p + 1];
            for (int r = 0; r < k; ++r) {
                const int *row = sLab + (size_t)ss.pos[r] * D;
                const int i = row[d0], j = row[d1];
                const size_t cell = ((size_t)p * C + i) * C + j;
                const int cN = Nc[cell], ca = ac[(size_t)p * C + j], cb = bc[(size_t)p * C + i];
                Nc[cell] = cN + 1;
                ac[(size_t)p * C + j] = ca + 1;
                bc[(size_t)p * C + i] = cb + 1;
                sN = sN - phi[cN] + phi[cN + 1];
                sa = sa - phi[ca] + phi[ca + 1];
                sb = sb - phi[cb] + phi[cb + 1];
            }
            SN[p] = sN;
            Sa[p] = sa;
            Sb[p] = sb;
        }
    }
    if (tid == 0) sc->nc = nc + k;
    FY_CLK(12);
}

__global__ __launch_bounds__(256) void k_mi_select(
    const int *__restrict__ asg, int D, int C, int P, const int *__restrict__ pairs,
    const int *__restrict__ batch, int B, int k, int mode, int *__restrict__ Nc, int *__restrict__ ac,
    int *__restrict__ bc, double *__restrict__ SN, double *__restrict__ Sa, double *__restrict__ Sb,
    const double *__restrict__ phi, MiScalars *__restrict__ sc, double *__restrict__ scores_out,
    long long *__restrict__ S_out, double *__restrict__ G_out, const int *__restrict__ forced_pos,
    int *__restrict__ trace_pos, long long *__restrict__ trace_ids, double *__restrict__ trace_scores,
    int keep_unselected, int *__restrict__ requeue_out)
{
    __shared__ SelShared ss;
    mi_select_stage(ss, P, pairs, SN, Sa, Sb, sc);
    mi_select_body(ss, asg, D, C, P, pairs, batch, 0, nullptr, B, k, mode, Nc, ac, bc, SN, Sa, Sb, phi, sc, scores_out, S_out, G_out,
                   forced_pos, trace_pos, trace_ids, trace_scores, keep_unselected, requeue_out);
}

// ------------------------------------------------------------------- exact greedy (mi / mem_mi)
// EfficientMI.run_greedy (mi.py:150-192): every iteration scores ALL remaining candidates and commits the
// first maximum.  One launch per iteration: each workgroup scores 256 candidates and reduces them to one
// (score, position) pair, ordered by (score descending, position ascending) -- removal keeps the order of the
// rest, so "first among the remaining" is "smallest original position".  The last workgroup to arrive
// (device-scope ticket) reduces the block results, commits the winner to the tables, marks it removed and emits
// S / GAIN.  forced: commit this original position instead (teacher forcing).
struct ExactBest {
    double s;
    int pos;
    int pad;
};
__device__ __forceinline__ bool exact_better(double s, int p, double so, int po)
{
    return so > s || (so == s && po < p);  // candidate (so, po) beats (s, p)
}

__global__ __launch_bounds__(256) void k_mi_exact_iter(
    const int *__restrict__ asg, int D, int C, int P, const int *__restrict__ pairs, const int *__restrict__ A, int L,
    unsigned char *__restrict__ removed, int *__restrict__ Nc, int *__restrict__ ac, int *__restrict__ bc,
    double *__restrict__ SN, double *__restrict__ Sa, double *__restrict__ Sb, const double *__restrict__ phi,
    MiScalars *__restrict__ sc, ExactBest *__restrict__ blockbest, unsigned *__restrict__ ticket,
    long long *__restrict__ S_out, double *__restrict__ G_out, const int *__restrict__ forced,
    double *__restrict__ trace_scores, int *__restrict__ trace_argmax, int measure, const double *__restrict__ lnk,
    const double *__restrict__ lf)
{
    // measure 0: calc_MI ('mi' / 'mem_mi'); 1: calc_AMI ('ami'); 2: calc_NMI (mi.py:262-271); 3: ConstantMeasure (mi.py:274-281:
    // every candidate scores 1, the first remaining one is taken)
    auto pair_score = [&](int p, int id, long long n) -> double {
        if (measure == 1) return ami_pair_score(asg, D, C, p, pairs, id, Nc, ac, bc, SN, Sa, Sb, phi, lnk, lf, n);
        if (measure == 2) return nmi_pair_score(asg, D, C, p, pairs, id, Nc, ac, bc, SN, Sa, Sb, phi, lnk, n);
        if (measure == 3) return 1.0;
        return mi_pair_score(asg, D, C, p, pairs, id, Nc, ac, bc, SN, Sa, Sb, phi, n);
    };
    __shared__ double sS[4];
    __shared__ int sP[4];
    __shared__ int sLast;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = blockIdx.x * 256 + tid;
    const long long nc = sc->nc;
    double s = -INFINITY;
    int pos = 0x7fffffff;
    if (w < L && !removed[w]) {
        const int id = A[w];
        double tot = 0.0;
        for (int p = 0; p < P; ++p) tot = tot + pair_score(p, id, nc);
        s = tot / (double)P;
        pos = w;
    }
    if (trace_scores && w < L) trace_scores[w] = pos == w ? s : (double)NAN;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const double so = __shfl_xor(s, dlt);
        const int po = __shfl_xor(pos, dlt);
        if (exact_better(s, pos, so, po)) s = so, pos = po;
    }
    if (lane == 0) sS[wave] = s, sP[wave] = pos;
    __syncthreads();
    if (tid == 0) {
        for (int q = 1; q < 4; ++q)
            if (exact_better(s, pos, sS[q], sP[q])) s = sS[q], pos = sP[q];
        blockbest[blockIdx.x].s = s;
        blockbest[blockIdx.x].pos = pos;
        __threadfence();  // the result is visible device-wide before the ticket is taken
        const unsigned t = atomicAdd(ticket, 1u);
        sLast = (t == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!sLast) return;  // uniform
    __threadfence();
    s = -INFINITY, pos = 0x7fffffff;
    for (int q = tid; q < (int)gridDim.x; q += 256) {
        const double so = __hip_atomic_load(&blockbest[q].s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int po = __hip_atomic_load(&blockbest[q].pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (exact_better(s, pos, so, po)) s = so, pos = po;
    }
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const double so = __shfl_xor(s, dlt);
        const int po = __shfl_xor(pos, dlt);
        if (exact_better(s, pos, so, po)) s = so, pos = po;
    }
    if (lane == 0) sS[wave] = s, sP[wave] = pos;
    __syncthreads();
    if (tid == 0) {
        for (int q = 1; q < 4; ++q)
            if (exact_better(s, pos, sS[q], sP[q])) s = sS[q], pos = sP[q];
        if (trace_argmax) *trace_argmax = pos;
        if (forced) {
            pos = *forced;
            double tot = 0.0;
            for (int p = 0; p < P; ++p) tot = tot + pair_score(p, A[pos], nc);
            s = tot / (double)P;
        }
        sP[0] = pos;
        *S_out = (long long)A[pos];
        *G_out = s;
        removed[pos] = 1;
        *ticket = 0u;
        sc->nc = nc + 1;
    }
    __syncthreads();
    const int id = A[sP[0]];
    for (int p = tid; p < P; p += 256) {  // update_cache: one pick
        const int *row = asg + (size_t)id * D;
        const int i = row[pairs[2 * p]], j = row[pairs[2 * p + 1]];
        const size_t cell = ((size_t)p * C + i) * C + j;
        const int cN = Nc[cell], ca = ac[(size_t)p * C + j], cb = bc[(size_t)p * C + i];
        Nc[cell] = cN + 1;
        ac[(size_t)p * C + j] = ca + 1;
        bc[(size_t)p * C + i] = cb + 1;
        SN[p] = SN[p] - phi[cN] + phi[cN + 1];
        Sa[p] = Sa[p] - phi[ca] + phi[ca + 1];
        Sb[p] = Sb[p] - phi[cb] + phi[cb + 1];
    }
}

// ------------------------------------------------------------------------------ MT19937
// Continues torch's CPU generator stream on the device: mt[624] + idx in global memory, one
// workgroup.  A block of 624 words is regenerated in three dependent phases (k<227, <454, <624).
__device__ __forceinline__ unsigned mt_temper(unsigned y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
__device__ __forceinline__ unsigned mt_twist(unsigned y) { return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }

// The refill loop of mt19937 is the uniform recurrence over the concatenated blocks
//     X[m] = X[m-227] ^ T(m-624),   T(j) = twist(U(j)),  U(j) = (X[j] & UPPER) | (X[j+1] & LOWER),   m >= 624
// (the "new mt[0]" the last word of a block pairs with is simply X[m-623]).  twist and U are GF(2)-linear, so the
// recurrence can be substituted into itself and the three twists folded into one:
//     X[m] = X[m-681] ^ twist(U(m-1078) ^ U(m-851) ^ U(m-624)),   m >= 1078
// whose newest operand is X[m-623]: 623 consecutive words are independent of each other.  One workgroup walks
// the stream in steps of 623 words (the first 454 words past the state in two plain 227-word steps), one barrier
// per step.  The window is LINEAR in LDS (all seven operands at constant offsets from one address register) and is
// slid back every MT_EPOCH steps.  The loop is VALU-issue bound (10 waves on 4 SIMDs), hence the folding, and hence
// the words are stored RAW: the consumer (k_fy_build) applies the tempering.  `out` must hold n + MT_PAD words.
// The state is left exactly as the sequential generator would leave it (block holding the last drawn word,
// idx in 1..624).
constexpr int MT_THREADS = 640;
constexpr int MT_WIDE = 623;
constexpr int MT_BACK = 1078;
constexpr int MT_EPOCH = 24;                              // wide steps between two slides of the window
constexpr int MT_WIN = MT_BACK + MT_WIDE * MT_EPOCH;      // 16030 words = 62.6 KB of LDS
constexpr int MT_GROUP = 8;   // greedy iterations whose draws one launch generates
constexpr int MT_PAD = 1280;  // the generator completes the 624-word block of the last draw (+ up to 622 words of the last step)
template <bool WRITE_BACK = true>
__device__ __forceinline__ void mt_generate_body(unsigned *__restrict__ mt_state, unsigned *__restrict__ out, long long n)
{
    __shared__ unsigned X[MT_WIN];
    const unsigned tid = threadIdx.x;
    if (n <= 0) return;
    const long long p = (long long)mt_state[624];  // first draw = X[p], 0 <= p <= 624
    for (unsigned k = tid; k < 624; k += MT_THREADS) X[k] = mt_state[k];
    const long long q = p + n;                      // one past the last draw
    const long long base = 624 * ((q - 1) / 624);   // block the sequential generator would hold
    const long long need = base + 624;              // generate at least up to here
    __syncthreads();
    for (long long m = p + tid; m < 624 && m < q; m += MT_THREADS) out[m - p] = X[m];  // draws left in the block
    unsigned *outp = out + (624 - p);  // raw word of stream index m goes to outp[m - 624]; nothing is stored past q
    long long m0 = 624;     // stream index of the next word to generate
    long long shift = 0;    // stream index of X[0]
    for (; m0 < need && m0 < MT_BACK; m0 += 227) {  // two plain steps: 624..850, 851..1077
        if (tid < 227) {
            const unsigned w = (unsigned)m0 + tid;
            const unsigned v = X[w - 227] ^ mt_twist((X[w - 624] & 0x80000000u) | (X[w - 623] & 0x7fffffffu));
            X[w] = v;
            if ((long long)w < q) outp[w - 624] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // loop-carried: ONE LDS address (all seven operands and the result at constant offsets from it) and one
    // byte offset into `out` (SGPR base + 32-bit VGPR offset store): the loop is VALU-issue bound
    const unsigned *xw = X + ((unsigned)(m0 - shift) + tid - MT_BACK);  // &X[bi - MT_BACK]
    unsigned ob = ((unsigned)(m0 - p) + tid) * 4u;                      // byte offset of out[m - p]  (n < 2^29 here)
    const unsigned ob_end = (unsigned)n * 4u;                          // lanes write back to back: no word past draw n
    const char *outb = reinterpret_cast<const char *>(out);
    int left = MT_EPOCH;
    while (m0 < need) {
        if (left == 0) {  // slide: the last MT_BACK words move to the front of the window (disjoint ranges)
            unsigned keep[2];
            const unsigned src = (unsigned)(m0 - shift) - MT_BACK;
#pragma unroll
            for (int u = 0; u < 2; ++u) keep[u] = tid + u * MT_THREADS < MT_BACK ? X[src + tid + u * MT_THREADS] : 0u;
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (tid + u * MT_THREADS < MT_BACK) X[tid + u * MT_THREADS] = keep[u];
            __syncthreads();
            shift = m0 - MT_BACK;
            xw = X + tid;
            left = MT_EPOCH;
        }
        if (tid < MT_WIDE) {
            // operands at constant offsets: 0,1 | 227,228 | 454,455 | 397; result at MT_BACK
            const unsigned a = xw[0] ^ xw[227] ^ xw[454], b2 = xw[1] ^ xw[228] ^ xw[455];
            const unsigned v = xw[MT_BACK - 681] ^ mt_twist((a & 0x80000000u) | (b2 & 0x7fffffffu));
            const_cast<unsigned *>(xw)[MT_BACK] = v;
            if (ob < ob_end) *reinterpret_cast<unsigned *>(const_cast<char *>(outb) + ob) = v;
        }
        xw += MT_WIDE;
        ob += MT_WIDE * 4u;
        m0 += MT_WIDE;
        --left;
        // LDS ordering only: a __syncthreads() would also wait for the global stores above to be acknowledged
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    __syncthreads();
    if (!WRITE_BACK) return;
    // words [base, base+624) are inside the window: at most 622 words were generated past `need`
    const unsigned wb = (unsigned)(base - shift);
    for (unsigned k = tid; k < 624; k += MT_THREADS) mt_state[k] = X[wb + k];
    if (tid == 0) mt_state[624] = (unsigned)(q - base);
}

__global__ __launch_bounds__(MT_THREADS) void k_mt_generate(unsigned *__restrict__ mt_state, unsigned *__restrict__ out,
                                                           long long n)
{
    mt_generate_body(mt_state, out, n);
}

// ONE stream from several workgroups ("lanes").  The stream past the host state is cut into blocks of `blk` words
// (a multiple of 624); lane w of W generates blocks w, w + W, w + 2W, ... and moves from one to the next with the GF(2)
// jump polynomial t^(W blk) mod phi (acav_mtjump.hip).  states[lane] = the 624 words preceding the lane's current block
// + idx (624 = all consumed; lane 0 starts from the host state and first hands out what is left of it); the generator
// leaves it alone, k_mt_jump advances it.  W is a power of two: every polynomial needed -- t^(2^r blk) to spread the
// lanes, t^(W blk) to advance them -- is a repeated square of t^blk.
__global__ __launch_bounds__(MT_THREADS) void k_mt_generate_lanes(unsigned *__restrict__ states, unsigned *__restrict__ out0,
                                                                 long long blk, long long head)
{
    ACAV_MI_EMPTY_RETURN
    const long long lane = blockIdx.x;
    const long long extra = lane == 0 ? head : 0;  // leftover draws of the host block, placed right before block 0
    mt_generate_body<false>(states + lane * 625, out0 + lane * blk - extra, blk + extra);  // the state stays at the block start
}

// states[dst0 + e] <- the window J words ahead of states[src0 + e] (e = blockIdx.x), J given by its polynomial
// g(t) = t^J mod phi:  Y[k] = XOR over { i : g_i = 1 } of X[i + k], X = the stream continued from the source window.
// The workgroup regenerates the 19937 + 623 words it needs in LDS (the same folded 623-wide recurrence as the
// generator), then thread k accumulates Y[k] branch-free.  ~0.1 ms per jump; a lane block takes ~0.85 ms to generate.
constexpr int MJ_WORDS = MT_BACK + MT_WIDE * 32;  // 21014 >= 624 + 19937
__global__ __launch_bounds__(MT_THREADS) void k_mt_jump(unsigned *__restrict__ states, int src0, int dst0,
                                                       const unsigned *__restrict__ poly)
{
    ACAV_MI_EMPTY_RETURN
    __shared__ unsigned X[MJ_WORDS];
    const unsigned tid = threadIdx.x;
    const unsigned *src = states + (size_t)(src0 + blockIdx.x) * 625;
    unsigned *dst = states + (size_t)(dst0 + blockIdx.x) * 625;
    for (unsigned k = tid; k < 624; k += MT_THREADS) X[k] = src[k];
    __syncthreads();
    for (unsigned m0 = 624; m0 < MT_BACK; m0 += 227) {
        if (tid < 227) {
            const unsigned w = m0 + tid;
            X[w] = X[w - 227] ^ mt_twist((X[w - 624] & 0x80000000u) | (X[w - 623] & 0x7fffffffu));
        }
        __syncthreads();
    }
    for (unsigned m0 = MT_BACK; m0 < MJ_WORDS; m0 += MT_WIDE) {
        if (tid < MT_WIDE) {
            const unsigned *xw = X + (m0 + tid - MT_BACK);
            const unsigned a = xw[0] ^ xw[227] ^ xw[454], b2 = xw[1] ^ xw[228] ^ xw[455];
            X[m0 + tid] = xw[MT_BACK - 681] ^ mt_twist((a & 0x80000000u) | (b2 & 0x7fffffffu));
        }
        __syncthreads();
    }
    if (tid < 624) {
        unsigned acc = 0u;
        const unsigned *xk = X + tid;
        for (int wi = 0; wi < 623; ++wi) {  // 19937 = 623 * 32 + 1 coefficients
            const unsigned g = poly[wi];  // uniform
#pragma unroll
            for (int b = 0; b < 32; ++b) acc ^= xk[wi * 32 + b] & (0u - ((g >> b) & 1u));
        }
        acc ^= xk[623 * 32] & (0u - (poly[623] & 1u));
        dst[tid] = acc;
    }
    if (tid == 0) dst[624] = 624u;
}

// ------------------------------------------------------------------- parallel Fisher-Yates
// Sequential semantics (torch.randperm + index_select == in-place):  for i in 0..L-2:
//   swap(A[i], A[h_i]),  h_i = i + draw_i % (L - i).
// Parallel evaluation.  list(p) = { j < p : h_j = p } (steps that drop a value into position p),
// g(p) = max list(p).  The ORIGINAL index of the value sitting at position p just before step p
// is f(p) = p if list(p) is empty, else f(g(p))  (a chain of expected length ~1).  The final
// value at position i is the content of h_i just before step i:
//   h_i == i            -> A[f(i)]
//   pred = max{ j in list(h_i) : j < i } exists  -> A[f(pred)]
//   otherwise                                     -> A[h_i]
// and position L-1 ends with A[f(L-1)].
__device__ __forceinline__ void fy_build_body(const unsigned *__restrict__ draws, int L, int *__restrict__ h,
                                              int *__restrict__ head, int *__restrict__ next, int *__restrict__ g, int i)
{
    if (i >= L) return;
    if (i == L - 1) {
        h[i] = i;
        next[i] = -1;
        return;
    }
    const int hh = i + (int)(mt_temper(draws[i]) % (unsigned)(L - i));  // k_mt_generate stores raw words
    h[i] = hh;
    if (hh != i) {
        next[i] = atomicExch(&head[hh], i);
        atomicMax(&g[hh], i);
    } else {
        next[i] = -1;
    }
}

__global__ __launch_bounds__(256) void k_fy_build(const unsigned *__restrict__ draws, int L, int *__restrict__ h,
                                                  int *__restrict__ head, int *__restrict__ next,
                                                  int *__restrict__ g)
{
    fy_build_body(draws, L, h, head, next, g, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

__device__ __forceinline__ void fy_apply_body(const int *__restrict__ A, int L, int B, const int *__restrict__ h,
                                              const int *__restrict__ head, const int *__restrict__ next,
                                              const int *__restrict__ g, int *__restrict__ batch_out,
                                              int *__restrict__ A_new, int *__restrict__ head_next,
                                              int *__restrict__ g_next, int i)
{
    if (i >= L) return;
    head_next[i] = -1;  // the list heads of the NEXT iteration (the other buffer pair; L only shrinks)
    g_next[i] = -1;
    const int p = h[i];
    int j;
    if (p == i) {
        j = i;
    } else {
        int pred = -1;
        for (int q = head[p]; q >= 0; q = next[q])
            if (q < i && q > pred) pred = q;
        j = pred;
    }
    int src;
    if (j < 0) {
        src = p;
    } else {
        while (g[j] >= 0) j = g[j];
        src = j;
    }
    const int v = A[src];
    if (i < B)
        batch_out[i] = v;
    else
        A_new[i - B] = v;
}

__global__ __launch_bounds__(256) void k_fy_apply(const int *__restrict__ A, int L, int B,
                                                  const int *__restrict__ h, const int *__restrict__ head,
                                                  const int *__restrict__ next, const int *__restrict__ g,
                                                  int *__restrict__ batch_out, int *__restrict__ A_new,
                                                  int *__restrict__ head_next, int *__restrict__ g_next)
{
    fy_apply_body(A, L, B, h, head, next, g, batch_out, A_new, head_next, g_next, (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

// ------------------------------------------------------------------- tiled Fisher-Yates
// The kernels above pay two device-scope atomics per candidate (list head exchange + max) -- at L = 10^6 that is the
// whole iteration (k_fy_build 88 us of 165).  This evaluation of the SAME swap sequence keeps every atomic in LDS:
//   k_fy_part     step j -> its target h_j; steps with h_j = j (and position L-1, which has no step) keep their own
//                 content reference; every other step is appended to the bucket of the TILE its target lies in
//                 (LDS histogram per workgroup, one global reservation per non-empty bin, unordered append)
//   k_fy_tile     one workgroup per tile: the pull lists of the tile's positions are built in LDS (ds exchange / max),
//                 every entry finds its predecessor in its list, and the tile emits  src[j] = "content of position h_j
//                 as it was before any step" (A-ref) or "what step pred left behind" (E-ref pred),  g[q] = last step
//                 that pulled from q
//   k_fy_resolve  perm[j] = the position whose ORIGINAL content output j receives: E-refs walk back through the last
//                 pullers (g chains, expected length ~1.5)
//   k_fy_gather_select   out[i] = A[perm[i]], and -- in workgroup 0, which is the one that produces the B batch entries --
//                 the scoring / top-k / commit of the greedy iteration (mi_select_body): the selection runs while the
//                 other workgroups still gather
// Only the last kernel touches content: the first three depend on the draws alone, run a GROUP of iterations per launch
// on a second stream, a group ahead.  Random 4-byte accesses are bound by the L2 request rate (one request per element,
// ~100-160 G/s on the whole chip), not by bytes, and either stream alone saturates it: at L = 10^6 the iteration takes
// what its kernels take one after the other (tools/exp/fy_bench.hip), so every request saved anywhere counts.
// Tiles are defined on e = L - 1 - h, the distance from the END of the list: the expected number of pulls on a
// position is ~ln(L / (e + 1)), it depends on e (not on L, which shrinks every iteration), so one tiling computed for
// the first iteration bounds every later one.  A tile is closed when it is `wcap` positions wide or its expected load
// reaches 70 % of the LDS entry capacity; a (never expected) overload is handled by sub-ranging the tile, a bucket or
// sub-range overflow raises the error flag and the run is refused -- never a wrong permutation.
#ifndef ACAV_FYA_CH
#define ACAV_FYA_CH 8192
#endif
#ifndef ACAV_FYT_THREADS
#define ACAV_FYT_THREADS 1024
#endif
constexpr int FYA_CH = ACAV_FYA_CH;  // steps per k_fy_part workgroup
constexpr int FYA_THREADS = 1024;
constexpr int64_t FY_TILED_MAX = 16 << 20;
constexpr unsigned FY_EREF = 0x80000000u;
// store flavours of the two scattered write streams of the position kernels (experiments: -DACAV_FY_BUCKET_NT, -DACAV_FY_SRC_NT,
// -DACAV_FY_SRC_SC1; measured in tools/exp/NOTES_r03.md)
#if defined(ACAV_FY_BUCKET_NT)
#define FY_ST_BUCKET(p, v) __builtin_nontemporal_store((v).x, &(p)->x), __builtin_nontemporal_store((v).y, &(p)->y)
#else
#define FY_ST_BUCKET(p, v) (*(p) = (v))
#endif
#if defined(ACAV_FY_SRC_NT)
#define FY_ST_SRC(p, v) __builtin_nontemporal_store((unsigned)(v), (p))
#elif defined(ACAV_FY_SRC_SC1)
#define FY_ST_SRC(p, v) __hip_atomic_store((p), (unsigned)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#else
#define FY_ST_SRC(p, v) (*(p) = (v))
#endif
#ifndef ACAV_FY_GROUP
#define ACAV_FY_GROUP 16
#endif
constexpr int FY_GROUP = ACAV_FY_GROUP;  // iterations per launch of the position kernels and per cross-stream hand-off (an event
                                         // wait + record on the content stream costs it ~10 us per group even when long satisfied)
#ifndef ACAV_FY_DEPTH
#define ACAV_FY_DEPTH 3
#endif
constexpr int FY_DEPTH = ACAV_FY_DEPTH;     // groups of perm buffers in flight: the position kernels run up to FY_DEPTH - 1 groups ahead
constexpr int FY_NBUF = FY_DEPTH * FY_GROUP;  // of the gathers (a cross-stream hand-off costs 15-45 us: two deep, both hand-offs
                                              // of a group sat on the critical cycle and left a ~30 us bubble per group)
constexpr int GS_EPT = 4;          // outputs per thread of the gather
constexpr int FY_SHARDS = 8;       // sub-buckets per tile (capg entries each), filled by workgroups b with b % 8 == shard
constexpr int FYT_THREADS = ACAV_FYT_THREADS;  // k_fy_tile: the list walks are chains of dependent LDS reads -- many waves hide them

// exclusive prefix sums of cnt[0 .. n) into pre[0 .. n) (both in LDS, distinct), by all threads of the workgroup (a multiple of
// 64, at most 1024); wsum: 16 ints of LDS; returns the total.  Ends with a barrier.
__device__ __forceinline__ int block_excl_scan(const int *cnt, int *pre, int n, int *wsum)
{
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, nw = nthr >> 6;
    const int per = (n + nthr - 1) / nthr, lo = tid * per;
    int s = 0;
    for (int q = 0; q < per; ++q) s += lo + q < n ? cnt[lo + q] : 0;
    int v = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        v += lane >= o ? t : 0;
    }
    if (lane == 63) wsum[tid >> 6] = v;
    __syncthreads();
    if (tid < 64) {
        int w = tid < nw ? wsum[tid] : 0;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const int t = __shfl_up(w, o, 64);
            w += lane >= o ? t : 0;
        }
        if (tid < nw) wsum[tid] = w;
    }
    __syncthreads();
    int run = ((tid >> 6) ? wsum[(tid >> 6) - 1] : 0) + v - s;
    for (int q = 0; q < per; ++q)
        if (lo + q < n) {
            pre[lo + q] = run;
            run += cnt[lo + q];
        }
    const int total = wsum[nw - 1];
    __syncthreads();
    return total;
}

// STAGED: the workgroup's appends are first sorted by tile in LDS and leave as runs of consecutive slots per tile (adjacent
// lanes -> adjacent addresses: a few L2 requests per wave store instead of one per lane)
template <bool STAGED>
__device__ __forceinline__ void fy_part_body(const unsigned *__restrict__ draws, int L, const unsigned short *__restrict__ table,
                                             int ntab, int gsh, int NT, int capg, int2 *__restrict__ bucket,
                                             int *__restrict__ gcount, unsigned *__restrict__ src, unsigned *__restrict__ err,
                                             int bx, int shard)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fy_smem[];
    int *lhist = reinterpret_cast<int *>(fy_smem), *lbase = lhist + NT;
    int2 *stage = reinterpret_cast<int2 *>(lbase + NT);                    // [FYA_CH] (STAGED)
    int *lpre = reinterpret_cast<int *>(stage + (STAGED ? FYA_CH : 0));    // [NT] (STAGED)
    int *wsum = lpre + (STAGED ? NT : 0);                                  // [16] (STAGED)
    unsigned short *ltab = reinterpret_cast<unsigned short *>(wsum + (STAGED ? 16 : 0));  // the tile table, a few KB: LDS lookups
    const int tid = threadIdx.x;
    constexpr int PER = FYA_CH / FYA_THREADS;
    const int base = bx * FYA_CH;
    unsigned dr[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {  // all draw loads in flight before anything else
        const int j = base + u * FYA_THREADS + tid;
        dr[u] = draws[j < L - 1 ? j : 0];
    }
    for (int t = tid; t < NT; t += FYA_THREADS) lhist[t] = 0;
    for (int t = tid; t < ntab; t += FYA_THREADS) ltab[t] = table[t];
    __syncthreads();
    int hh[PER], tr[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int j = base + u * FYA_THREADS + tid;
        tr[u] = -1;
        hh[u] = 0;
        if (j < L) {
            int h = j;
            if (j < L - 1) h = j + (int)(mt_temper(dr[u]) % (unsigned)(L - j));
            // every step leaves its reference here, in step order (coalesced): "the content of position j just before step j"
            // when it pulls from itself, else "the ORIGINAL content of h" -- right for the FIRST puller of a position (about
            // half of all steps); k_fy_tile overwrites it for the others only, so half of its scattered stores never happen
            src[j] = h == j ? (FY_EREF | (unsigned)j) : (unsigned)h;
            if (h != j) {
                const int tile = ltab[(L - 1 - h) >> gsh];
                tr[u] = (tile << 16) | atomicAdd(&lhist[tile], 1);
                hh[u] = h;
            }
        }
    }
    __syncthreads();
    // one reservation per non-empty bin -- in the workgroup's SHARD of the tile's bucket (the workgroup with linear id b
    // lands on XCD b % 8: a shard is written through one L2 only)
    for (int t = tid; t < NT; t += FYA_THREADS) {
        const int c = lhist[t];
        lbase[t] = c ? atomicAdd(&gcount[t * FY_SHARDS + shard], c) : 0;
    }
    if constexpr (STAGED) {
        const int total = block_excl_scan(lhist, lpre, NT, wsum);  // (its first barrier also publishes lbase)
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (tr[u] >= 0) stage[lpre[tr[u] >> 16] + (tr[u] & 0xffff)] = make_int2(base + u * FYA_THREADS + tid, hh[u]);
        __syncthreads();
        for (int x = tid; x < total; x += FYA_THREADS) {
            const int2 e = stage[x];
            const int tile = ltab[(L - 1 - e.y) >> gsh], pos = lbase[tile] + (x - lpre[tile]);
            if (pos < capg)
                FY_ST_BUCKET(&bucket[((size_t)tile * FY_SHARDS + shard) * capg + pos], e);
            else
                atomicOr(err, 1u);
        }
        return;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PER; ++u)
        if (tr[u] >= 0) {
            const int tile = tr[u] >> 16, pos = lbase[tile] + (tr[u] & 0xffff);
#ifdef ACAV_FY_ABL_BKSEQ  // timing-only ablation (tools/exp/fy_bench.hip): the appends as one sequential stream
            if (pos < capg)
                FY_ST_BUCKET(&bucket[(size_t)base + u * FYA_THREADS + tid], make_int2(base + u * FYA_THREADS + tid, hh[u]));
#else
            if (pos < capg)
                FY_ST_BUCKET(&bucket[((size_t)tile * FY_SHARDS + shard) * capg + pos], make_int2(base + u * FYA_THREADS + tid, hh[u]));
#endif
            else
                atomicOr(err, 1u);
        }
}

__device__ __forceinline__ void fy_tile_body(int L, const int *__restrict__ ebound, int capg, int ecap, int wcap,
                                             const int2 *__restrict__ bucket, int *__restrict__ gcount, unsigned *__restrict__ src,
                                             int *__restrict__ g, unsigned *__restrict__ err, int tile)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char fy_smem[];
    int *lhead = reinterpret_cast<int *>(fy_smem);   // [wcap] newest entry of the position's pull list
    int *lg = lhead + wcap;                            // [wcap] largest step that pulled from the position
    int *ej = lg + wcap;                               // [ecap] step of the entry
    unsigned short *ep = reinterpret_cast<unsigned short *>(ej + ecap);  // [ecap] local position
    unsigned short *enx = ep + ecap;                   // [ecap] next entry of the list, 0xFFFF = end
    __shared__ int lcnt;
    __shared__ int soff[FY_SHARDS + 1];
    const int tid = threadIdx.x;
    FY_CLK0();
    const int e_lo = ebound[tile], e_hi = ebound[tile + 1];
    if (e_lo > L - 1) return;  // the list no longer reaches this tile (no step targets it: its counts are 0)
    const int q_lo = L - e_hi, w = e_hi - e_lo;  // positions q_lo .. q_lo + w - 1 (q_lo may be negative)
    if (tid == 0) {
        int tot = 0;
        for (int sh = 0; sh < FY_SHARDS; ++sh) {
            int c = gcount[tile * FY_SHARDS + sh];
            gcount[tile * FY_SHARDS + sh] = 0;  // for the next group's k_fy_part
            c = c > capg ? capg : c;           // k_fy_part has raised the error flag
            soff[sh] = tot;
            tot += c;
        }
        soff[FY_SHARDS] = tot;
    }
    __syncthreads();
    FY_CLK(0);
    const int count = soff[FY_SHARDS];
    const int2 *bk = bucket + (size_t)tile * FY_SHARDS * capg;
    const int nsub = count <= ecap ? 1 : (count + (ecap >> 2) - 1) / (ecap >> 2);
    for (int r = 0; r < nsub; ++r) {
        const int p0 = (int)((long long)w * r / nsub), p1 = (int)((long long)w * (r + 1) / nsub);
        for (int p = p0 + tid; p < p1; p += FYT_THREADS) {
            lhead[p] = -1;
            lg[p] = -1;
        }
        if (tid == 0) lcnt = 0;
        __syncthreads();
        FY_CLK(1);
        // entry x of the concatenated shards; every load of the thread is issued before the first one is used (a loop of
        // load -> LDS update pays one memory round trip per entry: that was most of this kernel)
        constexpr int TPF = 8;
        for (int x0 = tid; x0 < count; x0 += TPF * FYT_THREADS) {
            int2 jh[TPF];
#pragma unroll
            for (int u = 0; u < TPF; ++u) {
                const int x = x0 + u * FYT_THREADS;
                int sh = 0;
#pragma unroll
                for (int q = 1; q < FY_SHARDS; ++q) sh += x >= soff[q] ? 1 : 0;
                jh[u] = x < count ? bk[(size_t)sh * capg + (x - soff[sh])] : make_int2(0, 0);
            }
#pragma unroll
            for (int u = 0; u < TPF; ++u) {
                const int x = x0 + u * FYT_THREADS;
                const int p = jh[u].y - q_lo;
                if (x >= count || (nsub > 1 && (p < p0 || p >= p1))) continue;
                const int e = nsub == 1 ? x : atomicAdd(&lcnt, 1);
                if (e < ecap) {
                    ej[e] = jh[u].x;
                    ep[e] = (unsigned short)p;
                    enx[e] = (unsigned short)atomicExch(&lhead[p], e);  // -1 -> 0xFFFF
                    atomicMax(&lg[p], jh[u].x);
                } else {
                    atomicOr(err, 2u);
                }
            }
        }
        __syncthreads();
        FY_CLK(2);
        const int n = nsub == 1 ? count : (lcnt < ecap ? lcnt : ecap);
        for (int e = tid; e < n; e += FYT_THREADS) {
            const int p = ep[e], j = ej[e];
            int pred = -1;
            for (int x = lhead[p]; x >= 0; x = enx[x] == 0xFFFFu ? -1 : (int)enx[x]) {
                const int jj = ej[x];
                if (jj < j && jj > pred) pred = jj;
            }
#ifdef ACAV_FY_ABL_SRCSEQ  // timing-only ablation: the src stores in entry order instead of step order
            FY_ST_SRC(&src[((size_t)tile * 4096 + e) % (size_t)L], pred >= 0 ? (FY_EREF | (unsigned)pred) : (unsigned)(q_lo + p));
#else
            if (pred >= 0) FY_ST_SRC(&src[j], FY_EREF | (unsigned)pred);  // a first puller keeps k_fy_part's A-ref (= q_lo + p)
#endif
        }
        FY_CLK(3);
        for (int p = p0 + tid; p < p1; p += FYT_THREADS)
            if (q_lo + p >= 0) g[q_lo + p] = lg[p];
        __syncthreads();
        FY_CLK(4);
    }
}

// One chunk of a tiled run (chunk.py:21-53; a single-chunk run is the case of one descriptor).  The three position
// kernels serve a GROUP of iterations of every chunk per launch (blockIdx.y = chunk, blockIdx.z = iteration within the
// group): nothing they compute depends on what the greedy selects, only on the draws, so the iterations of a group are
// independent.  Everything per chunk comes from the descriptor, everything that changes per iteration is a function of
// the iteration number (L_t = L0 - t dl, buffer indices, the position of the iteration's draws in the chunk's generator
// ring: r0(t) = t (L0 - 1) - dl t (t - 1) / 2).
struct TileChunk {
    const unsigned *ring;  // draw r' (counted from the first GENERATED word) at ring[r' mod ring_words]; r < head: ring[r - head]
    long long head, ring_words;
    const unsigned short *table;
    const int *ebound;
    int2 *bucket;            // [FY_GROUP][NT][FY_SHARDS][capg]
    int *gcount;             // [FY_GROUP][NT][FY_SHARDS]
    unsigned *src[FY_GROUP]; // content references and last pullers, per iteration of the group in flight
    int *g[FY_GROUP];
    unsigned *perm[FY_NBUF]; // source position of every output position, per iteration (FY_DEPTH groups deep)
    int *tailinv;            // [FY_NBUF][SEL_MAXB] the output position that reads the r-th of the last `nreq` list positions
    int *A[2];
    unsigned *err;
    const int *asg, *pairs;
    int *batch, *Nc, *ac, *bc;  // batch: [2][SEL_MAXB (1 + D)]: ids and label rows of the batch of iteration t in half t & 1
    double *SN, *Sa, *Sb;
    const double *phi;
    MiScalars *sc;
    long long *S;
    double *G;
    const int *forced;       // teacher forcing / traces of a single-chunk run (NULL otherwise)
    int *tr_pos;
    long long *tr_ids;
    double *tr_sc;
    int L0, iters, ntab, gsh, NT, capg, ecap, wcap, D, C, P, pad;
};

__device__ __forceinline__ const unsigned *chunk_draw_ptr(const TileChunk &c, int it, int dl)
{
    const long long r0 = (long long)it * (c.L0 - 1) - (long long)dl * it * (it - 1) / 2;
    return r0 < c.head ? c.ring - (c.head - r0) : c.ring + (r0 - c.head) % c.ring_words;
}

template <bool STAGED>
__global__ __launch_bounds__(FYA_THREADS) void k_fy_part_multi(const TileChunk *__restrict__ cd, int it0, int dl)
{
    ACAV_MI_EMPTY_RETURN
    const TileChunk &c = cd[blockIdx.y];
    const int z = (int)blockIdx.z, it = it0 + z;
    const int L = c.L0 - it * dl;
    if (it >= c.iters || (int)blockIdx.x * FYA_CH >= L) return;
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    fy_part_body<STAGED>(chunk_draw_ptr(c, it, dl), L, c.table, c.ntab, c.gsh, c.NT, c.capg, c.bucket + (size_t)z * c.NT * FY_SHARDS * c.capg,
                 c.gcount + (size_t)z * c.NT * FY_SHARDS, c.src[z], c.err, (int)blockIdx.x, (int)(lin & (FY_SHARDS - 1)));
}

__global__ __launch_bounds__(FYT_THREADS) void k_fy_tile_multi(const TileChunk *__restrict__ cd, int it0, int dl)
{
    ACAV_MI_EMPTY_RETURN
    const TileChunk &c = cd[blockIdx.y];
    const int z = (int)blockIdx.z, it = it0 + z;
    if (it >= c.iters || (int)blockIdx.x >= c.NT) return;
    fy_tile_body(c.L0 - it * dl, c.ebound, c.capg, c.ecap, c.wcap, c.bucket + (size_t)z * c.NT * FY_SHARDS * c.capg,
                 c.gcount + (size_t)z * c.NT * FY_SHARDS, c.src[z], c.g[z], c.err, (int)blockIdx.x);
}

// perm[i] = the position (before the iteration) whose content output position i receives: the E-ref chains are walked
// here, beside the content path -- the gather that waits for the previous selection is then one indexed copy
__global__ __launch_bounds__(256) void k_fy_resolve_multi(const TileChunk *__restrict__ cd, int it0, int dl, int nreq)
{
    ACAV_MI_EMPTY_RETURN
    // one element per thread: a wave waits for the longest of its chains, and four elements per thread (256 chains per
    // wave) made the kernel 1.5x slower
    const TileChunk &c = cd[blockIdx.y];
    const int z = (int)blockIdx.z, it = it0 + z;
    const int L = c.L0 - it * dl;
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (it >= c.iters || i >= L) return;
    const unsigned s = c.src[z][i];
    int a = (int)(s & 0x7fffffffu);
#ifdef ACAV_FY_ABL_NOWALK  // timing-only ablation: no chain reads
    if (0) {
#else
    if (s & FY_EREF) {  // what step a left behind: walk back to the position whose original content that was
#endif
        const int *__restrict__ g = c.g[z];
        int ga;
        while ((ga = g[a]) >= 0) a = ga;
    }
    c.perm[it % FY_NBUF][i] = (unsigned)a;
    // the last nreq positions of the list hold what the PREVIOUS iteration's selection re-queues (see
    // k_fy_gather_select_multi): who reads them
    if (a >= L - nreq) c.tailinv[(size_t)(it % FY_NBUF) * SEL_MAXB + (a - (L - nreq))] = i;
}

// Launch `it` of the content stream, two things that do not depend on each other (software pipelining: the selection is a
// latency chain of ~5 us, the gather a bandwidth burst of ~6 us; in one dependent sequence they cost their sum, and the
// selection's round trips were slowed further by the gather's traffic):
//   workgroups 1..   GATHER of iteration `it`: out[i] = A_it[perm_it[i]]; outputs i < B are the batch of iteration `it` (kept
//                    for the next launch), the rest the head of list it+1.  The last nreq = B - k positions of A_it are what
//                    the selection of iteration it-1 re-queues -- which runs in THIS launch: an output that reads one of
//                    them is left to workgroup 0 (exactly one output per position: tailinv, from k_fy_resolve).
//   workgroup 0      SELECTION of iteration it-1 (mi_select_body: scoring / top-k / commit on the batch gathered by the
//                    previous launch), then the nreq re-queued ids go straight to the outputs of iteration `it` that read them.
// Launches run it = 0 .. iters (the first has no selection, the last no gather).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_fy_gather_select_multi(
    const TileChunk *__restrict__ cd, int it, int dl, int B, int k, int mode, int keep_unselected)
{
    ACAV_MI_EMPTY_RETURN
    const TileChunk &c = cd[blockIdx.y];
    const int L = c.L0 - it * dl;  // list length of iteration `it`
    const int nreq = keep_unselected ? B - k : 0;
    const int bstride = SEL_MAXB * (1 + c.D);  // one half of the batch buffer: ids [SEL_MAXB], label rows [SEL_MAXB][D]
    if (blockIdx.x != 0) {
        if (it >= c.iters) return;
        // GS_EPT outputs per thread, every index load before the first content load (1 -> 4 per thread: 34.9 -> 33.7 us per
        // iteration at L = 10^6; 8 the same, 16 slower)
        int av[GS_EPT], vv[GS_EPT];
#pragma unroll
        for (int u = 0; u < GS_EPT; ++u) {
            const int i = (int)(((blockIdx.x - 1) * GS_EPT + u) * 256 + threadIdx.x);
            av[u] = i < L ? (int)c.perm[it % FY_NBUF][i] : -1;
        }
#pragma unroll
        for (int u = 0; u < GS_EPT; ++u) {
            if (av[u] >= 0 && it > 0 && av[u] >= L - nreq) av[u] = -1;  // not there yet: workgroup 0 delivers it
            vv[u] = av[u] >= 0 ? c.A[it & 1][av[u]] : 0;
        }
#pragma unroll
        for (int u = 0; u < GS_EPT; ++u) {
            const int i = (int)(((blockIdx.x - 1) * GS_EPT + u) * 256 + threadIdx.x);
            if (av[u] < 0) continue;
            const int v = vv[u];
            if (i < B) {  // a batch entry travels with its label row: the next launch's selection starts one level further on
                int *bb = c.batch + (it & 1) * bstride;
                bb[i] = v;
                for (int d = 0; d < c.D; ++d) bb[SEL_MAXB + i * c.D + d] = c.asg[(size_t)v * c.D + d];
            } else {
                c.A[(it + 1) & 1][i - B] = v;
            }
        }
        return;
    }
    if (it < 1 || it > c.iters) return;
    const int ps = it - 1;  // the iteration whose selection this is
    __shared__ SelShared ss;
    __shared__ int sReq[SEL_MAXB];
    const int *bprev = c.batch + (ps & 1) * bstride;
    mi_select_stage(ss, c.P, c.pairs, c.SN, c.Sa, c.Sb, c.sc);
    const int id = (int)threadIdx.x < B ? bprev[threadIdx.x] : 0;
    const bool feeds = it < c.iters && (int)threadIdx.x < nreq;  // this thread delivers a re-queued id to the gather of `it`
    const int dest = feeds ? c.tailinv[(size_t)(it % FY_NBUF) * SEL_MAXB + threadIdx.x] : 0;
    mi_select_body(ss, c.asg, c.D, c.C, c.P, c.pairs, nullptr, id, bprev + SEL_MAXB, B, k, mode, c.Nc, c.ac, c.bc, c.SN, c.Sa, c.Sb,
                   c.phi, c.sc, nullptr, c.S + (size_t)ps * k, c.G + (size_t)ps * k, c.forced ? c.forced + (size_t)ps * k : nullptr,
                   c.tr_pos ? c.tr_pos + (size_t)ps * k : nullptr, c.tr_ids ? c.tr_ids + (size_t)ps * B : nullptr,
                   c.tr_sc ? c.tr_sc + (size_t)ps * B : nullptr, keep_unselected, sReq);
    if (it >= c.iters || nreq == 0) return;  // no gather left to feed
    lds_barrier();
    if (feeds) {
        const int v = sReq[threadIdx.x];
        if (dest < B) {
            int *bb = c.batch + (it & 1) * bstride;
            bb[dest] = v;
            for (int d = 0; d < c.D; ++d) bb[SEL_MAXB + dest * c.D + d] = c.asg[(size_t)v * c.D + d];
        } else {
            c.A[(it + 1) & 1][dest - B] = v;
        }
    }
}

// ------------------------------------------------------------------ several chunks in lockstep
// The greedy loop of ONE chunk is a chain of small dependent kernels: most of the GPU idles, and several chunks
// driven from several host threads do not overlap (the HIP runtime serialises the launches).  Chunks are
// independent (chunk.py:21-53), so the same three launches per iteration (+ one generator launch per group) can
// serve a whole batch of them: blockIdx.y (or .x for the one-workgroup kernels) picks the chunk, every per-chunk
// pointer and size comes from a descriptor array in device memory, and everything that changes per iteration is a
// function of the iteration number (L_t = L0 - t * dl, buffer parities, offsets).  Same device functions as the
// single-chunk kernels, hence the same results.
struct ChunkDesc {
    const int *asg, *pairs;
    int *Nc, *ac, *bc;
    double *SN, *Sa, *Sb;
    const double *phi;
    MiScalars *sc;
    int *A[2];
    unsigned *draws[2];
    int *h, *next, *head[2], *g[2];
    unsigned *mt;
    int *batch;
    long long *S;
    double *G;
    int D, C, P, L0, iters, pad;
};

__device__ __forceinline__ long long chunk_draws(const ChunkDesc &c, int t0, int t1, int dl)
{  // draws of iterations [t0, t1) of the chunk: sum of (L_t - 1)
    long long tot = 0;
    for (int t = t0; t < t1 && t < c.iters; ++t) {
        const int lt = c.L0 - t * dl;
        tot += lt > 1 ? lt - 1 : 0;
    }
    return tot;
}

__global__ __launch_bounds__(MT_THREADS) void k_mt_generate_multi(const ChunkDesc *__restrict__ cd, int group, int dl)
{
    const ChunkDesc c = cd[blockIdx.x];
    const long long n = chunk_draws(c, group * MT_GROUP, (group + 1) * MT_GROUP, dl);
    if (n > 0) mt_generate_body(c.mt, c.draws[group & 1], n);
}

__global__ __launch_bounds__(256) void k_fy_build_multi(const ChunkDesc *__restrict__ cd, int it, int dl)
{
    const ChunkDesc c = cd[blockIdx.y];
    if (it >= c.iters) return;
    const int group = it / MT_GROUP;
    const unsigned *draws = c.draws[group & 1] + chunk_draws(c, group * MT_GROUP, it, dl);
    fy_build_body(draws, c.L0 - it * dl, c.h, c.head[it & 1], c.next, c.g[it & 1], (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

__global__ __launch_bounds__(256) void k_fy_apply_multi(const ChunkDesc *__restrict__ cd, int it, int dl, int B)
{
    const ChunkDesc c = cd[blockIdx.y];
    if (it >= c.iters) return;
    fy_apply_body(c.A[it & 1], c.L0 - it * dl, B, c.h, c.head[it & 1], c.next, c.g[it & 1], c.batch, c.A[(it + 1) & 1],
                  c.head[(it + 1) & 1], c.g[(it + 1) & 1], (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

__global__ __launch_bounds__(256) void k_mi_select_multi(const ChunkDesc *__restrict__ cd, int it, int dl, int B, int k,
                                                         int mode, int keep_unselected)
{
    const ChunkDesc c = cd[blockIdx.x];
    if (it >= c.iters) return;
    __shared__ SelShared ss;
    mi_select_stage(ss, c.P, c.pairs, c.SN, c.Sa, c.Sb, c.sc);
    mi_select_body(ss, c.asg, c.D, c.C, c.P, c.pairs, c.batch, 0, nullptr, B, k, mode, c.Nc, c.ac, c.bc, c.SN, c.Sa, c.Sb, c.phi, c.sc, nullptr,
                   c.S + (size_t)it * k, c.G + (size_t)it * k, nullptr, nullptr, nullptr, nullptr, keep_unselected,
                   c.A[(it + 1) & 1] + (c.L0 - it * dl - B));
}

__global__ void k_i64_to_i32(const long long *__restrict__ in, int *__restrict__ out, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int)in[i];
}

}  // namespace

// =============================================================================== handle
struct acav_mi {
    StreamCtx ctx;
    int64_t V = 0;
    int D = 0, C = 0, P = 0;
    DevBuf asg, pairs, Nc, ac, bc, SN, Sa, Sb, phi, scalars;
    DevBuf stage, ids32, scores;
    // greedy buffers
    DevBuf A0, A1, draws, draws2, h, head, head2, next, g, g2, mt, batch, S, G, tr_pos, tr_ids, tr_sc, forced;
    DevBuf removed, blockbest, ticket, tr_am;  // exact greedy
    DevBuf chunk_desc;                         // descriptor array of a multi-chunk run (lead handle)
    DevBuf lane_states, ring, polys;           // MT19937 lanes of the single-chunk greedy (MtStream)
    DevBuf lnk, lf;   // ln k and ln k! tables of the `ami` score (acav_mi_set_measure)
    int measure = 0;  // exact greedy: 0 = calc_MI, 1 = calc_AMI, 2 = calc_NMI, 3 = constant
    int queue_probe_replaced = 0;  // streams replaced by mi_separate_queues (diagnostics: ACAV_MI_TIMING prints it)
    bool lockstep_member = false;  // ran as one of several chunks of acav_mi_run_greedy_multi: its streams / events are RETIRED at destroy
    bool queue_probe_pending = false;  // the three streams have not been checked for a shared hardware queue yet
    bool prio_streams = false;         // ACAV_MI_STREAM_PRIO: priority classes of the generator / position stream (created lazily)
    int class_mt = 0, class_fy = 0;
    DevBuf fy_table, fy_bounds, fy_bucket, fy_count, fy_src[FY_GROUP], fy_g[FY_GROUP], fy_perm[FY_NBUF], fy_tail, fy_err;  // tiled Fisher-Yates
    hipStream_t st_fy = nullptr;               // the position kernels of group g+1 run beside the gathers of group g
    hipEvent_t ev_tile[FY_NBUF] = {}, ev_gather[FY_NBUF] = {};
    // the mt19937 stream does not depend on the selection state: it is generated one iteration ahead on
    // its own stream (double-buffered draws), overlapping the Fisher-Yates / select kernels
    hipStream_t st_mt = nullptr;
    hipEvent_t ev_mt[2] = {nullptr, nullptr}, ev_used[2] = {nullptr, nullptr};
};

struct acav_rng;  // state access through the C ABI below

// Host-side plan of one chunk's draw stream: T draws in all, produced on the generator stream `smt` by W lanes in
// superblocks of S = W * blk words into a two-slot ring, consumed by the Fisher-Yates kernels on `st` iteration by
// iteration.  Draw r (0 = the first draw of the run) lives at
//     r <  head                ring[pad - head + r]                 (what was left of the host state's block)
//     r >= head, r' = r - head ring[pad + r' mod (NSLOT S)]         (+ a mirror of the first `lmax` words of slot 0
//                                                                    behind the last slot: an iteration that straddles
//                                                                    the wrap still reads one contiguous range)
// Nothing depends on what the greedy selects, so the whole schedule (which superblock an iteration needs, when a slot
// may be overwritten) is computed on the host; the two streams meet through events only.
static int mi_ensure_gen_events(acav_mi *mi);
struct MtStream {
    static constexpr int NSLOT = 2;
    static constexpr int64_t PAD = 624;
    static constexpr int64_t BLK_DEFAULT = 624 * 4096;  // 2.5 M words per lane block: ~0.85 ms of generation per hop
    int64_t T = 0, head = 0, blk = 0, S = 0, nblocks = 0, nsuper = 0, lmax = 0;
    int p0 = 0, W = 1, logW = 0;
    bool wraps = false;
    unsigned *ring = nullptr, *states = nullptr, *polys = nullptr;
    hipStream_t st = nullptr, smt = nullptr;  // st: the stream of the kernel that reads the draws
    hipEvent_t ev_mt[NSLOT] = {nullptr, nullptr}, ev_used[NSLOT] = {nullptr, nullptr};
    int64_t produced = 0;  // superblocks enqueued on smt
    int64_t waited = -1;   // highest superblock `st` has been told to wait for
    int64_t freed = 0;     // superblocks [0, freed) are no longer read by any iteration still to be enqueued

    // L: the longest list (one iteration reads at most L - 1 contiguous draws); span: the most draws one acquire() covers
    int plan(acav_mi *mi, hipStream_t consumer, const uint32_t *mtbuf, int idx, int64_t total_draws, int64_t L, int64_t span,
             hipStream_t generator = nullptr)
    {
        T = total_draws;
        p0 = idx;
        head = 624 - p0;
        lmax = L;
        const int64_t gen = T > head ? T - head : 0;  // words that must be generated past the host block
        if (gen <= BLK_DEFAULT) {  // a single lane block, cut to size
            W = 1;
            blk = 624 * ((gen + 623) / 624);
            nblocks = gen > 0 ? 1 : 0;
        } else {
            blk = BLK_DEFAULT;
            nblocks = (gen + blk - 1) / blk;
            W = 1;
            while (W < 32 && W < nblocks) W *= 2;
            if ((nblocks + W - 1) / W > NSLOT && (int64_t)W * blk < 2 * span) {  // a slot must hold what one acquire() covers twice over
                blk = 624 * ((2 * span + 624 * (int64_t)W - 1) / (624 * (int64_t)W));
                nblocks = (gen + blk - 1) / blk;
            }
        }
        logW = 0;
        while ((1 << logW) < W) ++logW;
        S = (int64_t)W * blk;
        nsuper = nblocks ? (nblocks + W - 1) / W : 0;
        wraps = nsuper > NSLOT;
        st = consumer;
        smt = generator ? generator : mi->st_mt;
        ACAV_TRY(mi_ensure_gen_events(mi));  // (created on first need: acav_mi_create)
        for (int q = 0; q < NSLOT; ++q) ev_mt[q] = mi->ev_mt[q], ev_used[q] = mi->ev_used[q];
        const int64_t slots = nsuper < NSLOT ? nsuper : NSLOT;
        ACAV_TRY(mi->ring.ensure(sizeof(unsigned) * (size_t)(PAD + slots * S + (wraps ? lmax : 0) + 8)));
        ACAV_TRY(mi->lane_states.ensure(sizeof(unsigned) * 625 * (size_t)W));
        ring = mi->ring.as<unsigned>();
        states = mi->lane_states.as<unsigned>();
        // lane 0 = the host state; the leftover draws of its block are also what iteration 0 starts with when nothing
        // is generated at all (T <= head)
        uint32_t st0[625];
        memcpy(st0, mtbuf, 624 * sizeof(uint32_t));
        st0[624] = (uint32_t)p0;
        ACAV_HIP_TRY(hipMemcpyAsync(states, st0, sizeof(st0), hipMemcpyHostToDevice, st));
        if (head > 0) ACAV_HIP_TRY(hipMemcpyAsync(ring + PAD - head, mtbuf + p0, sizeof(unsigned) * (size_t)head, hipMemcpyHostToDevice, st));
        ACAV_HIP_TRY(hipStreamSynchronize(st));  // st0 / mtbuf are the caller's locals
        // jump polynomials t^(2^r blk): r < logW spreads the lanes, r = logW advances a lane to its next block
        const int npoly = logW + 1;
        std::vector<uint32_t> hp((size_t)npoly * 624, 0u);
        for (int r = 0; r < npoly; ++r) {
            if (r == logW && nsuper <= 1) break;
            const uint32_t *g = mt_jump_poly(((int64_t)1 << r) * blk);  // cached; each is the square of the one before
            ACAV_REQUIRE(g, ACAV_ESTATE, "could not derive the MT19937 jump polynomial");
            memcpy(&hp[(size_t)r * 624], g, 624 * sizeof(uint32_t));
        }
        ACAV_TRY(mi->polys.ensure(sizeof(uint32_t) * hp.size()));
        polys = mi->polys.as<unsigned>();
        ACAV_HIP_TRY(hipMemcpy(polys, hp.data(), sizeof(uint32_t) * hp.size(), hipMemcpyHostToDevice));
        produced = 0, waited = -1, freed = 0;
        // everything queued so far on the main stream happens before the first generator launch
        ACAV_HIP_TRY(hipEventRecord(ev_used[0], st));
        ACAV_HIP_TRY(hipStreamWaitEvent(smt, ev_used[0], 0));
        // spread: lanes [2^r, 2^(r+1)) <- lanes [0, 2^r) jumped by 2^r blocks (before lane 0 starts consuming its state)
        for (int r = 0; r < logW; ++r)
            hipLaunchKernelGGL(k_mt_jump, dim3(1u << r), dim3(MT_THREADS), 0, smt, states, 0, 1 << r, polys + (size_t)r * 624);
        ACAV_HIP_TRY(hipGetLastError());
        for (int64_t q = 0; q < slots; ++q) ACAV_TRY(produce());
        return ACAV_OK;
    }
    int produce()  // enqueue superblock `produced` (its slot is free)
    {
        const int64_t s = produced++;
        const int slot = (int)(s % NSLOT);
        if (s >= NSLOT) ACAV_HIP_TRY(hipStreamWaitEvent(smt, ev_used[slot], 0));  // the readers of superblock s - NSLOT are done
        const int64_t left = nblocks - s * W;
        const unsigned lanes = (unsigned)(left < W ? left : W);
        unsigned *out0 = ring + PAD + (int64_t)slot * S;
        hipLaunchKernelGGL(k_mt_generate_lanes, dim3(lanes), dim3(MT_THREADS), 0, smt, states, out0, (long long)blk,
                           (long long)(s == 0 ? head : 0));
        if (s + 1 < nsuper)  // every lane moves on to its next block, W blocks further
            hipLaunchKernelGGL(k_mt_jump, dim3((unsigned)W), dim3(MT_THREADS), 0, smt, states, 0, 0, polys + (size_t)logW * 624);
        ACAV_HIP_TRY(hipGetLastError());
        if (wraps && slot == 0 && s > 0)  // mirror of the slot's first lmax words behind the last slot
            ACAV_HIP_TRY(hipMemcpyAsync(ring + PAD + NSLOT * S, out0, sizeof(unsigned) * (size_t)lmax, hipMemcpyDeviceToDevice, smt));
        ACAV_HIP_TRY(hipEventRecord(ev_mt[slot], smt));
        return ACAV_OK;
    }
    // called before the kernels of an iteration that reads draws [r0, r0 + n) are enqueued on st: make st wait for the
    // superblocks it needs; returns the device pointer of draw r0
    int acquire(int64_t r0, int64_t n, const unsigned **ptr)
    {
        const int64_t r1 = r0 + n;  // one past the last draw
        const int64_t hi = r1 - 1 >= head ? (r1 - 1 - head) / S : -1;
        ACAV_REQUIRE(hi < produced, ACAV_ESTATE, "MT stream schedule: superblock %lld needed, %lld produced", (long long)hi,
                     (long long)produced);
        for (int64_t s = waited + 1; s <= hi; ++s) ACAV_HIP_TRY(hipStreamWaitEvent(st, ev_mt[s % NSLOT], 0));
        if (hi > waited) waited = hi;
        *ptr = r0 < head ? ring + PAD - head + r0 : ring + PAD + (r0 - head) % (NSLOT * S);
        return ACAV_OK;
    }
    // called after the last kernel reading the draws of the current iteration has been enqueued; next_r0 = first draw
    // of the following iteration (T when there is none): superblocks entirely below it are free for the producer
    int release(int64_t next_r0)
    {
        const int64_t lo = next_r0 >= head ? (next_r0 - head) / S : 0;
        while (freed < lo && freed < nsuper) {
            const int64_t s = freed++;
            if (s + NSLOT < nsuper) {
                ACAV_HIP_TRY(hipEventRecord(ev_used[s % NSLOT], st));
                ACAV_TRY(produce());
            }
        }
        return ACAV_OK;
    }
    // the generator state the host continues from: the 624-block holding the last draw, idx in 1..624
    int final_state(uint32_t *mtbuf, int *idx) const
    {
        const int64_t q = (int64_t)p0 + T;  // one past the last draw, counted from word 0 of the host block
        if (q <= 624) {
            *idx = (int)q;  // the host block itself (mtbuf still holds it)
            return ACAV_OK;
        }
        const int64_t base = 624 * ((q - 1) / 624);
        const int64_t r = base - p0;  // draw index of the block's first word (>= head)
        ACAV_HIP_TRY(hipMemcpy(mtbuf, ring + PAD + (r - head) % (NSLOT * S), 624 * sizeof(uint32_t), hipMemcpyDeviceToHost));
        *idx = (int)(q - base);
        return ACAV_OK;
    }
};

// Tiling of the tiled Fisher-Yates for a list of (at most) L0 candidates; see the kernels' header comment.
struct FyPlan {
    int gsh = 4, wcap = 256, ecap = 256, ecap_lds = 256, capg = 512, NT = 0;
    std::vector<unsigned short> table;  // (e >> gsh) -> tile
    std::vector<int> ebound;            // tile t covers e in [ebound[t], ebound[t+1])
    void build(int64_t L0)
    {
        int cap = 256, cap_max = 8192;
        if (const char *v = getenv("ACAV_FY_CAP")) {  // experiments: smaller tiles (more workgroups per CU)
            const int x = atoi(v);
            if (x >= 256 && x <= 8192 && (x & (x - 1)) == 0) cap_max = x;
        }
        // at least ~32 tiles per iteration (the kernels take FY_GROUP iterations per launch: hundreds of workgroups); smaller
        // tiles only multiply 1024-thread workgroups with a few hundred entries each (8 x 100k chunks in lockstep: 4.85 us per
        // chunk-iteration at 128 tiles, 4.0 at 32)
        int tiles_min = 32;
        if (const char *v = getenv("ACAV_FY_TILES_MIN")) tiles_min = atoi(v) > 0 ? atoi(v) : tiles_min;
        while (cap < cap_max && (int64_t)cap * tiles_min < L0) cap *= 2;
        wcap = ecap = cap;
        // capacity of ONE shard of a tile's bucket: with many k_fy_part workgroups the shards fill evenly (an eighth of
        // the tile's load each, 4x headroom); with few, one shard may receive everything
        capg = (L0 + FYA_CH - 1) / FYA_CH >= 64 ? cap / 2 : 2 * cap;
        gsh = 0;
        while ((1 << gsh) < cap / 16) ++gsh;
        const int gran = 1 << gsh;
        const double limit = 0.7 * ecap, lnL = log((double)L0);
        table.assign((size_t)((L0 + gran - 1) >> gsh) + 1, 0);
        ebound.clear();
        int64_t e = 0;
        while (e < L0) {
            int width = 0;
            double mass = 0.0;
            do {  // expected pulls on the granule [x, x + gran): at most gran * ln(L0 / (x + 1))
                const double x = (double)(e + width);
                const double m = gran * (lnL - log(x + 1.0)) + 1.0;
                if (width > 0 && mass + (m > 0 ? m : 0) > limit) break;
                mass += m > 0 ? m : 0;
                width += gran;
            } while (width < wcap && e + width < L0);
            const int t = (int)ebound.size();
            ebound.push_back((int)e);
            for (int64_t x = e; x < e + width; x += gran) table[(size_t)(x >> gsh)] = (unsigned short)t;
            e += width;
        }
        ebound.push_back((int)e);
        NT = (int)ebound.size() - 1;
        ecap_lds = ecap;
        if (const char *v = getenv("ACAV_FY_ECAP")) {  // tests: force the sub-ranged overload path
            const int x = atoi(v);
            if (x >= 16 && x < ecap) ecap_lds = x;
        }
    }
    size_t tile_smem() const { return (size_t)wcap * 8 + (size_t)ecap_lds * 8; }
};

static size_t fy_part_smem(int NT, size_t ntab, bool staged)
{
    return sizeof(int) * (staged ? 3 : 2) * (size_t)NT + (staged ? sizeof(int2) * FYA_CH + sizeof(int) * 16 : 0) + sizeof(unsigned short) * ntab;
}

// tiling of a list of (at most) L candidates and the handle's buffers for it: table, tile bounds, sharded buckets and their
// counters (one set per iteration of a group; zeroed), error flag (cleared), src / g (per iteration of a group) and perm
// (two groups deep) buffers; uploads are stream-ordered on st (fp must stay alive
// until st has been synchronised)
static int fy_setup(acav_mi *mi, int64_t L, FyPlan &fp, hipStream_t st)
{
    fp.build(L);
    ACAV_TRY(mi->fy_table.ensure(sizeof(unsigned short) * fp.table.size()));
    ACAV_TRY(mi->fy_bounds.ensure(sizeof(int) * fp.ebound.size()));
    ACAV_TRY(mi->fy_bucket.ensure(sizeof(int2) * (size_t)FY_GROUP * fp.NT * FY_SHARDS * fp.capg));
    ACAV_TRY(mi->fy_count.ensure(sizeof(int) * (size_t)FY_GROUP * fp.NT * FY_SHARDS));
    ACAV_TRY(mi->fy_err.ensure(sizeof(unsigned)));
    for (int q = 0; q < FY_GROUP; ++q) {
        ACAV_TRY(mi->fy_src[q].ensure(sizeof(unsigned) * (size_t)L));
        ACAV_TRY(mi->fy_g[q].ensure(sizeof(int) * (size_t)L));
    }
    for (int q = 0; q < FY_NBUF; ++q) ACAV_TRY(mi->fy_perm[q].ensure(sizeof(unsigned) * (size_t)L));
    ACAV_TRY(mi->fy_tail.ensure(sizeof(int) * (size_t)FY_NBUF * SEL_MAXB));
    ACAV_HIP_TRY(hipMemcpyAsync(mi->fy_table.p, fp.table.data(), sizeof(unsigned short) * fp.table.size(), hipMemcpyHostToDevice, st));
    ACAV_HIP_TRY(hipMemcpyAsync(mi->fy_bounds.p, fp.ebound.data(), sizeof(int) * fp.ebound.size(), hipMemcpyHostToDevice, st));
    ACAV_HIP_TRY(hipMemsetAsync(mi->fy_count.p, 0, sizeof(int) * (size_t)FY_GROUP * fp.NT * FY_SHARDS, st));
    ACAV_HIP_TRY(hipMemsetAsync(mi->fy_err.p, 0, sizeof(unsigned), st));
    return ACAV_OK;
}

// candidate / sample ids (int64 at the boundary, as the reference's lists) -> int32 on the device; `on`: the stream the copy and the
// conversion are ordered on (default: the handle's own)
static int ids_to_device32(acav_mi *mi, const int64_t *ids, int64_t n, DevBuf &stage, DevBuf &out32, hipStream_t on = nullptr)
{
    hipStream_t st = on ? on : mi->ctx.stream;
    if (!is_device_ptr(ids)) {
        for (int64_t i = 0; i < n; ++i)
            ACAV_REQUIRE(ids[i] >= 0 && ids[i] < mi->V, ACAV_EINVAL, "id %lld out of range [0,%lld)",
                         (long long)ids[i], (long long)mi->V);
    }
    const void *d = nullptr;
    ACAV_TRY(to_device(ids, sizeof(int64_t) * (size_t)n, stage, st, &d));
    ACAV_TRY(out32.ensure(sizeof(int) * (size_t)(n > 0 ? n : 1)));
    if (n > 0) {
        hipLaunchKernelGGL(k_i64_to_i32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           static_cast<const long long *>(d), out32.as<int>(), (long long)n);
        ACAV_HIP_TRY(hipGetLastError());
    }
    return ACAV_OK;
}

ACAV_EXPORT int acav_mi_destroy(acav_mi *mi);

// ---- the loop's three streams on three hardware queues, whatever the process did before (round 6) --------------------------------
// Two of them on one queue cost a third of the loop's speed (43.5 vs 31 us per iteration, NOTES_r05 section 13), and which queue the
// runtime gives a new stream depends on every stream the process has created and destroyed.  The API does not say which queue a
// stream is on, so the handle MEASURES it once: two one-wave kernels that each wait 100 us of s_memrealtime, one per stream -- side by
// side (~0.1 ms) on two queues, one after the other (~0.2 ms) on one.  A stream that shares its queue is replaced by a newly created
// one (the old one stays alive until the end of the search, so that the runtime does not hand the same slot out again); at most six
// replacements, then the handle takes what it has.  ~0.5 ms per handle that owns its streams; ACAV_MI_QUEUE_PROBE=0 skips it.
__global__ void k_spin_ticks(unsigned long long ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}
static bool mi_streams_share_queue(hipStream_t a, hipStream_t b)
{
    double best = 1e30;
    for (int rep = 0; rep < 2; ++rep) {  // the better of two tries: a busy device can only make a pair look serialised, never overlapped
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_spin_ticks, dim3(1), dim3(64), 0, a, 10000ull);  // 100 us at 100 MHz
        hipLaunchKernelGGL(k_spin_ticks, dim3(1), dim3(64), 0, b, 10000ull);
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        best = us < best ? us : best;
        if (best < 160.0) break;
    }
    return best >= 160.0;
}
static void mi_separate_queues(acav_mi *mi, bool own_content, int class_mt, int class_fy);
static int mi_ensure_streams(acav_mi *mi);  // st_mt / st_fy on first need


ACAV_EXPORT int acav_mi_create(acav_mi **out, int device, const int64_t *assignments, int64_t V, int D, int C,
                               const int32_t *pairs, int P, void *stream)
{
    ACAV_REQUIRE(out && assignments && pairs, ACAV_EINVAL, "NULL argument");
    ACAV_REQUIRE(V > 0 && V < 0x7fffffff && D > 0 && C > 0 && P > 0, ACAV_EINVAL, "bad sizes V=%lld D=%d C=%d P=%d",
                 (long long)V, D, C, P);
    ACAV_REQUIRE(!is_device_ptr(assignments) && !is_device_ptr(pairs), ACAV_EINVAL,
                 "assignments and pairs are host arrays (as in the reference constructor)");
    for (int p = 0; p < 2 * P; ++p)
        ACAV_REQUIRE(pairs[p] >= 0 && pairs[p] < D, ACAV_EINVAL, "pair index %d out of range", pairs[p]);
    std::vector<int> a32((size_t)V * D);
    for (size_t i = 0; i < a32.size(); ++i) {
        ACAV_REQUIRE(assignments[i] >= 0 && assignments[i] < C, ACAV_EINVAL,
                     "assignment %lld outside [0,%d) (one_hot would fail: mi.py:69-74)", (long long)assignments[i], C);
        a32[i] = (int)assignments[i];
    }
    acav_mi *mi = new (std::nothrow) acav_mi;
    ACAV_REQUIRE(mi, ACAV_ENOMEM, "out of host memory");
    // The greedy loop runs on three streams (content: gather + select per iteration; positions: the Fisher-Yates kernels, groups ahead;
    // generator).  They need three different hardware queues: two of them on one queue and the loop takes 43.5 us per iteration
    // instead of 31, all three on one 48.8 (GPU_MAX_HW_QUEUES = 2 / 1, V = 10^6) -- and which queue a stream gets depends on every
    // other stream the process has created and destroyed (bench.py once lost a third of its MI speed to two idle k-means handles
    // that outlived their pass: NOTES_r05 section 13).  The runtime keeps one pool of hardware queues PER PRIORITY CLASS, so three
    // streams in three classes are on three queues whatever the process's history: ACAV_MI_STREAM_PRIO=hnl (classes of the content /
    // position / generator stream) does that -- 30.9 us even with GPU_MAX_HW_QUEUES=1.  It is NOT the default: chunks in lockstep run
    // 40 % slower with any non-default class on any of the three (5.1-5.6 vs 3.7 us per chunk-iteration, all six assignments), and
    // a fresh process -- the CLI's case -- places three consecutively created streams on three queues anyway.
    const char *vprio = getenv("ACAV_MI_STREAM_PRIO");
    const bool prio = vprio && strlen(vprio) == 3;
    const char *pmap = prio ? vprio : "nnn";  // three letters h / n / l
    auto cls = [](char ch) { return ch == 'h' ? 1 : ch == 'l' ? -1 : 0; };
    int rc = mi->ctx.init(device, stream, prio ? cls(pmap[0]) : 0);
    if (rc != ACAV_OK) {
        delete mi;
        return rc;
    }
    mi->V = V;
    mi->D = D;
    mi->C = C;
    mi->P = P;
    hipStream_t st = mi->ctx.stream;
    const size_t cc = (size_t)P * C * C, pc = (size_t)P * C;
    std::vector<double> phi((size_t)V + 2);
    phi[0] = 0.0;
    for (int64_t k = 1; k < V + 2; ++k) phi[(size_t)k] = (double)k * log((double)k);
    auto body = [&]() -> int {
        ACAV_TRY(mi->pairs.ensure(sizeof(int) * 2 * (size_t)P));
        ACAV_TRY(mi->Nc.ensure(sizeof(int) * cc));
        ACAV_TRY(mi->ac.ensure(sizeof(int) * pc));
        ACAV_TRY(mi->bc.ensure(sizeof(int) * pc));
        ACAV_TRY(mi->SN.ensure(sizeof(double) * P));
        ACAV_TRY(mi->Sa.ensure(sizeof(double) * P));
        ACAV_TRY(mi->Sb.ensure(sizeof(double) * P));
        ACAV_TRY(mi->scalars.ensure(sizeof(MiScalars)));
        ACAV_TRY(upload(mi->asg, a32.data(), sizeof(int) * a32.size(), st));
        ACAV_HIP_TRY(hipMemcpyAsync(mi->pairs.p, pairs, sizeof(int) * 2 * (size_t)P, hipMemcpyHostToDevice, st));
        ACAV_TRY(upload(mi->phi, phi.data(), sizeof(double) * phi.size(), st));
        ACAV_HIP_TRY(hipMemsetAsync(mi->Nc.p, 0, sizeof(int) * cc, st));
        ACAV_HIP_TRY(hipMemsetAsync(mi->ac.p, 0, sizeof(int) * pc, st));
        ACAV_HIP_TRY(hipMemsetAsync(mi->bc.p, 0, sizeof(int) * pc, st));
        ACAV_HIP_TRY(hipMemsetAsync(mi->SN.p, 0, sizeof(double) * P, st));
        ACAV_HIP_TRY(hipMemsetAsync(mi->Sa.p, 0, sizeof(double) * P, st));
        ACAV_HIP_TRY(hipMemsetAsync(mi->Sb.p, 0, sizeof(double) * P, st));
        ACAV_HIP_TRY(hipMemsetAsync(mi->scalars.p, 0, sizeof(MiScalars), st));
        ACAV_HIP_TRY(hipStreamSynchronize(st));
        return ACAV_OK;
    };
    rc = body();
    if (rc == ACAV_OK) {
        // The generator and position STREAMS are created when a greedy loop first needs them (mi_ensure_streams): a lockstep group
        // creates a handle per chunk and runs every launch on the LEAD's streams -- ten chunks used to create 30 streams per group
        // for the 12 they use, and the lockstep loop slows down with the number of hardware queues the process keeps busy
        // (3.6 us per chunk-iteration with up to 16, 5.6-6.1 with 32: profiles/r06_mi_lockstep_by_hw_queues.txt).
        // ... and so are the loop's EVENTS (mi_ensure_gen_events / mi_ensure_streams): 100 per handle, of which a chunk that is not its
        // group's lead uses four
        mi->prio_streams = prio;
        mi->class_mt = cls(pmap[2]);
        mi->class_fy = stream ? 1 : cls(pmap[1]);  // (a caller's own content stream is most likely of the default class: the position stream goes up instead)
    }
    if (rc != ACAV_OK) {
        acav_mi_destroy(mi);
        return rc;
    }
    // (the queue probe runs lazily, at the handle's first single-chunk greedy loop over a long candidate list: run_greedy_tiled)
    mi->queue_probe_pending = !prio;
    *out = mi;
    return ACAV_OK;
}

static int mi_ensure_gen_events(acav_mi *mi)  // the generator ring's events of THIS handle's draws (MtStream::plan, legacy loop)
{
    for (int q = 0; q < 2; ++q) {
        if (!mi->ev_mt[q]) ACAV_HIP_TRY(hipEventCreateWithFlags(&mi->ev_mt[q], hipEventDisableTiming));
        if (!mi->ev_used[q]) ACAV_HIP_TRY(hipEventCreateWithFlags(&mi->ev_used[q], hipEventDisableTiming));
    }
    return ACAV_OK;
}

static int mi_ensure_streams(acav_mi *mi)
{
    ACAV_TRY(mi_ensure_gen_events(mi));
    for (int q = 0; q < FY_DEPTH; ++q) {  // the group hand-offs between the position and the content stream (indices < FY_DEPTH are used)
        if (!mi->ev_tile[q]) ACAV_HIP_TRY(hipEventCreateWithFlags(&mi->ev_tile[q], hipEventDisableTiming));
        if (!mi->ev_gather[q]) ACAV_HIP_TRY(hipEventCreateWithFlags(&mi->ev_gather[q], hipEventDisableTiming));
    }
    if (mi->st_mt && mi->st_fy) return ACAV_OK;
    ACAV_HIP_TRY(hipSetDevice(mi->ctx.device));
    int plo = 0, phi = 0;
    ACAV_HIP_TRY(hipDeviceGetStreamPriorityRange(&plo, &phi));
    auto make = [&](hipStream_t *s, int cl) -> hipError_t {
        if (*s) return hipSuccess;
        return mi->prio_streams ? hipStreamCreateWithPriority(s, hipStreamNonBlocking, cl > 0 ? phi : cl < 0 ? plo : (plo + phi) / 2)
                                : hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    };
    ACAV_HIP_TRY(make(&mi->st_mt, mi->class_mt));
    ACAV_HIP_TRY(make(&mi->st_fy, mi->class_fy));
    return ACAV_OK;
}

static void mi_separate_queues(acav_mi *mi, bool own_content, int class_mt, int class_fy)
{
    if (mi_ensure_streams(mi) != ACAV_OK) return;
    (void)own_content, (void)class_mt, (void)class_fy;
    std::vector<hipStream_t> parked;
    for (int attempt = 0; attempt < 6; ++attempt) {
        hipStream_t *victim = nullptr;
        if (mi_streams_share_queue(mi->ctx.stream, mi->st_mt)) victim = &mi->st_mt;
        else if (mi_streams_share_queue(mi->ctx.stream, mi->st_fy)) victim = &mi->st_fy;
        else if (mi_streams_share_queue(mi->st_mt, mi->st_fy)) victim = &mi->st_fy;
        if (!victim) break;
        hipStream_t fresh = nullptr;
        if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) break;
        parked.push_back(*victim);
        *victim = fresh;
        mi->queue_probe_replaced += 1;
    }
    for (hipStream_t s : parked) (void)hipStreamDestroy(s);
    (void)hipGetLastError();
}

ACAV_EXPORT int acav_mi_destroy(acav_mi *mi)
{
    if (!mi) return ACAV_OK;
    (void)hipSetDevice(mi->ctx.device);
    (void)hipStreamSynchronize(mi->ctx.stream);
    // A handle that ran in a lockstep group RETIRES its streams and events (destroyed later, in bulk: acav_common.h): the ten handles
    // of a group cost 45-90 ms to destroy between two groups (cfg5 slice: selection 2.98 -> 2.51 s).  Every other handle destroys them
    // on the spot, as always: with retired streams alive the NEXT single-chunk loop's streams are placed differently (bench.py 28.9
    // instead of 28.6 us per iteration; 37.8 when the retired ones are flushed right before that loop -- the placement of three
    // streams on hardware queues is the process's history, tools/exp/NOTES_r06.md sections 5 and 10).
    const bool retire = mi->lockstep_member;
    auto drop_stream = [&](hipStream_t s) {
        if (!s) return;
        (void)hipStreamSynchronize(s);
        if (retire) retire_stream(s);
        else (void)hipStreamDestroy(s);
    };
    auto drop_event = [&](hipEvent_t e) {
        if (!e) return;
        if (retire) retire_event(e);
        else (void)hipEventDestroy(e);
    };
    drop_stream(mi->st_mt);
    drop_stream(mi->st_fy);
    for (int q = 0; q < 2; ++q) drop_event(mi->ev_mt[q]), drop_event(mi->ev_used[q]);
    for (int q = 0; q < FY_NBUF; ++q) drop_event(mi->ev_tile[q]), drop_event(mi->ev_gather[q]);
    mi->ctx.fini(retire);
    delete mi;
    return ACAV_OK;
}
// single-chunk options of acav_mi_run_greedy (teacher forcing, traces, an iteration cap)
struct TiledExtras {
    int64_t *trace_ids = nullptr;
    double *trace_scores = nullptr;
    int32_t *trace_pos = nullptr;
    const int32_t *forced_pos = nullptr;
    int64_t max_iters = -1;
};

// The greedy run on the lane generator + tiled Fisher-Yates kernels: one chunk (acav_mi_run_greedy) or several independent
// chunks in lockstep (acav_mi_run_greedy_multi; one handle, candidate list, start set, subset size and generator each;
// every list within FY_TILED_MAX).  All handles live on the same device; the first handle's streams carry the work.
// Two streams: the position kernels (k_fy_part / k_fy_tile / k_fy_resolve: no content, one launch each per GROUP of
// iterations) run on st_fy ahead of the gathers + selections on the main stream; perm is buffered FY_DEPTH groups
// deep and the streams meet through one event pair per group (a cross-stream hand-off costs ~20 us of latency,
// an event record / wait a few us of queue time: per iteration they sat on the critical path).
static int run_greedy_tiled(acav_mi **mis, int nchunks, const int64_t *const *candidates, const int64_t *L,
                            const int64_t *const *start, const int *ns, const int64_t *subset, int B, int k,
                            int keep_unselected, acav_rng **rngs, int64_t *const *S_out, double *const *GAIN_out,
                            int64_t *n_selected, int64_t *n_iters, const TiledExtras &ex)
{
    acav_mi *lead = mis[0];
    // Three streams on three hardware queues (mi_separate_queues): checked once per handle, and only where it matters and cannot
    // hurt -- ONE chunk over a long candidate list (the position and content streams of a 10^6-candidate loop overlap for ~30 us
    // per iteration).  Probing each of the 125 handles of the cfg5 slice at creation cost the slice 0.6 s (2.49 -> 3.15 s of selection),
    // which is why the probe moved here.  Probing the LEAD of every lockstep group was tried as well (its three streams carry the
    // group's launches): 3-8 streams replaced over 52 groups and no effect on the lockstep loop's own bimodality -- the cfg5 slice's
    // selection takes 2.46-2.52 s in most processes and 3.3-4.0 s in about one of three, with the probe (3.98 / 2.46) and without
    // (3.53 / 2.48 / 2.50): tools/exp/NOTES_r06.md section 5.
    if (nchunks == 1 && lead->queue_probe_pending && L[0] >= 250000) {
        const char *vq = getenv("ACAV_MI_QUEUE_PROBE");
        if (!(vq && vq[0] == '0')) mi_separate_queues(lead, true, 0, 0);
        lead->queue_probe_pending = false;
    }
    ACAV_TRY(mi_ensure_streams(lead));
    hipStream_t st = lead->ctx.stream, sf = lead->st_fy;
    const bool timing = getenv("ACAV_MI_TIMING") != nullptr;
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); };
    const auto t_setup0 = clk::now();
    double t_ids = 0, t_fy = 0, t_mt = 0, t_add = 0;
    const int64_t dl = B - (keep_unselected ? B - k : 0);  // candidates consumed per iteration
    std::vector<TileChunk> desc((size_t)nchunks);
    std::vector<int64_t> iters((size_t)nchunks, 0), r0((size_t)nchunks, 0);
    std::vector<FyPlan> plans((size_t)nchunks);
    std::vector<MtStream> streams((size_t)nchunks);
    int64_t iters_max = 0, lmax = 0;
    int pmax = 1, dmax = 1, ntmax = 1;
    size_t part_smem = 0, part_smem_staged = 0, tile_smem = 0;
    for (int c = 0; c < nchunks; ++c) {
        acav_mi *mi = mis[c];
        ACAV_HIP_TRY(hipStreamSynchronize(mi->ctx.stream));  // whatever the handle was doing on its own stream is over
        if (nchunks > 1) mi->lockstep_member = true;
        // the staging block at its final size BEFORE the start samples use it: growing a DevBuf that is in use synchronises the
        // device (it cannot know which stream still reads the old block) -- 4.2 ms in a process with a few dozen streams, ten times
        // per lockstep group = 38 of a group's 41 ms of set-up (what the timers called "ids": the copy itself takes 0.02 ms)
        if (!is_device_ptr(candidates[c])) ACAV_TRY(mi->stage.ensure(sizeof(int64_t) * (size_t)L[c]));
        { const auto t0 = clk::now();
        if (ns[c]) ACAV_TRY(acav_mi_add_samples(mi, start[c], ns[c]));  // batch.py:215
        t_add += ms_since(t0); }
        // plan: the number of iterations and every L_t are known on the host (no device feedback)
        int64_t nS = 0, l = L[c], itc = 0, draws = 0;
        while (nS < subset[c] && (ex.max_iters < 0 || itc < ex.max_iters)) {
            ACAV_REQUIRE(l >= B, ACAV_ERANGE,
                         "chunk %d: %lld candidates left < batch_size %d: the reference's topk(k=floor(B/k*B')) raises here "
                         "(batch.py:143-150)", c, (long long)l, B);
            draws += l > 1 ? l - 1 : 0;
            nS += k;
            l -= dl;
            ++itc;
        }
        iters[(size_t)c] = itc;
        iters_max = itc > iters_max ? itc : iters_max;
        lmax = L[c] > lmax ? L[c] : lmax;
        pmax = mi->P > pmax ? mi->P : pmax;
        dmax = mi->D > dmax ? mi->D : dmax;
        const size_t Lc = (size_t)L[c];
        ACAV_TRY(mi->A0.ensure(sizeof(int) * (Lc + B)));  // before the conversion: ensure() does not copy
        { const auto t0 = clk::now();
        // on the LEAD's content stream, the one the loop's gathers run on: ordered before them by the stream itself (on the chunk's
        // own stream the conversion kernel was ordered by nothing but its brevity)
        ACAV_TRY(ids_to_device32(mi, candidates[c], L[c], mi->stage, mi->A0, st));
        t_ids += ms_since(t0); }
        ACAV_TRY(mi->A1.ensure(sizeof(int) * (Lc + B)));
        ACAV_TRY(mi->batch.ensure(sizeof(int) * 2 * SEL_MAXB * (size_t)(1 + mi->D)));
        ACAV_TRY(mi->S.ensure(sizeof(long long) * (size_t)(itc * k + 1)));
        ACAV_TRY(mi->G.ensure(sizeof(double) * (size_t)(itc * k + 1)));
        if (ex.trace_pos) ACAV_TRY(mi->tr_pos.ensure(sizeof(int) * (size_t)(itc * k + 1)));
        if (ex.trace_ids) ACAV_TRY(mi->tr_ids.ensure(sizeof(long long) * (size_t)(itc * B + 1)));
        if (ex.trace_scores) ACAV_TRY(mi->tr_sc.ensure(sizeof(double) * (size_t)(itc * B + 1)));
        if (ex.forced_pos) {
            for (int64_t i = 0; i < itc * k; ++i)
                ACAV_REQUIRE(ex.forced_pos[i] >= 0 && ex.forced_pos[i] < B, ACAV_EINVAL, "forced position out of range");
            ACAV_TRY(mi->forced.ensure(sizeof(int) * (size_t)(itc * k + 1)));
            ACAV_HIP_TRY(hipMemcpyAsync(mi->forced.p, ex.forced_pos, sizeof(int) * (size_t)(itc * k), hipMemcpyHostToDevice, st));
        }
        FyPlan &fp = plans[(size_t)c];
        { const auto t0 = clk::now();
        ACAV_TRY(fy_setup(mi, L[c], fp, st));
        t_fy += ms_since(t0); }
        ntmax = fp.NT > ntmax ? fp.NT : ntmax;
        const size_t ps = fy_part_smem(fp.NT, fp.table.size(), false), pss = fy_part_smem(fp.NT, fp.table.size(), true);
        part_smem = ps > part_smem ? ps : part_smem;
        part_smem_staged = pss > part_smem_staged ? pss : part_smem_staged;
        tile_smem = fp.tile_smem() > tile_smem ? fp.tile_smem() : tile_smem;
        // hand the host MT19937 stream to the device: W lanes generate it superblock by superblock on their own stream
        // (MtStream); nothing they do depends on what gets selected (L shrinks by a fixed amount per iteration)
        unsigned mtbuf[625];
        int idx = 0;
        ACAV_TRY(acav_rng_get_state(rngs[c], mtbuf, &idx));
        MtStream &ms = streams[(size_t)c];
        // Chunks in lockstep: ALL their generators run on the lead handle's generator stream (round 5).  A generator launch is rare
        // (a superblock of 82 M words lasts ~800 iterations of a 100k-clip chunk) and off the critical path, but ten generator
        // streams are ten more queues for the runtime to service: with GPU_MAX_HW_QUEUES = 16 (which ten clusterings training
        // side by side need, acav100m_amd/__init__.py) the lockstep loop ran at 7.4 us per chunk-iteration instead of 3.8;
        // on one shared stream 3.66 at either setting.  ACAV_MI_SHARE_GEN=0: a stream per chunk again, =n: n streams (A/B).
        const char *vsh = getenv("ACAV_MI_SHARE_GEN");
        const int nshare = vsh ? atoi(vsh) : 1;
        if (nshare > 0) ACAV_TRY(mi_ensure_streams(mis[c % nshare]));
        else ACAV_TRY(mi_ensure_streams(mi));
        ACAV_TRY(mi_ensure_gen_events(mi));
        { const auto t0 = clk::now();
        ACAV_TRY(ms.plan(mi, sf, mtbuf, idx, draws, L[c], (int64_t)FY_GROUP * L[c], nshare > 0 ? mis[c % nshare]->st_mt : nullptr));
        t_mt += ms_since(t0); }
        TileChunk &d = desc[(size_t)c];
        d.ring = ms.ring + MtStream::PAD;
        d.head = ms.head;
        d.ring_words = ms.wraps ? MtStream::NSLOT * ms.S : ((long long)1 << 62);
        d.table = mi->fy_table.as<unsigned short>(), d.ebound = mi->fy_bounds.as<int>();
        d.bucket = mi->fy_bucket.as<int2>(), d.gcount = mi->fy_count.as<int>(), d.err = mi->fy_err.as<unsigned>();
        for (int q = 0; q < FY_GROUP; ++q) d.src[q] = mi->fy_src[q].as<unsigned>(), d.g[q] = mi->fy_g[q].as<int>();
        for (int q = 0; q < FY_NBUF; ++q) d.perm[q] = mi->fy_perm[q].as<unsigned>();
        d.tailinv = mi->fy_tail.as<int>();
        d.A[0] = mi->A0.as<int>(), d.A[1] = mi->A1.as<int>();
        d.asg = mi->asg.as<int>(), d.pairs = mi->pairs.as<int>(), d.batch = mi->batch.as<int>();
        d.Nc = mi->Nc.as<int>(), d.ac = mi->ac.as<int>(), d.bc = mi->bc.as<int>();
        d.SN = mi->SN.as<double>(), d.Sa = mi->Sa.as<double>(), d.Sb = mi->Sb.as<double>();
        d.phi = mi->phi.as<double>(), d.sc = mi->scalars.as<MiScalars>();
        d.S = mi->S.as<long long>(), d.G = mi->G.as<double>();
        d.forced = ex.forced_pos ? mi->forced.as<int>() : nullptr;
        d.tr_pos = ex.trace_pos ? mi->tr_pos.as<int>() : nullptr;
        d.tr_ids = ex.trace_ids ? mi->tr_ids.as<long long>() : nullptr;
        d.tr_sc = ex.trace_scores ? mi->tr_sc.as<double>() : nullptr;
        d.L0 = (int)L[c], d.iters = (int)itc, d.ntab = (int)fp.table.size(), d.gsh = fp.gsh, d.NT = fp.NT, d.capg = fp.capg;
        d.ecap = fp.ecap_lds, d.wcap = fp.wcap, d.D = mi->D, d.C = mi->C, d.P = mi->P, d.pad = 0;
    }
    ACAV_TRY(lead->chunk_desc.ensure(sizeof(TileChunk) * (size_t)nchunks));
    ACAV_HIP_TRY(hipMemcpyAsync(lead->chunk_desc.p, desc.data(), sizeof(TileChunk) * (size_t)nchunks, hipMemcpyHostToDevice, st));
    const double t_chunks = ms_since(t_setup0);
    ACAV_HIP_TRY(hipStreamSynchronize(st));  // tables, counters, candidate lists, forced positions and descriptors are in place
    const double t_setup = ms_since(t_setup0);
    ACAV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fy_tile_multi), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)tile_smem));
    // the staged form of k_fy_part wants 64 KB of LDS beside the tile table (which grows with L: 64 KB at 16 Mi candidates)
    const bool part_staged = part_smem_staged <= 96 * 1024 && !getenv("ACAV_FY_PART_DIRECT");
    if (part_staged) {
        part_smem = part_smem_staged;
        ACAV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fy_part_multi<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)part_smem));
    }
    const TileChunk *dcd = lead->chunk_desc.as<TileChunk>();
    const int sel_f = sel_mode(B, pmax, k);  // one mode for every chunk of the launch: sized for the largest P and D
    const size_t sel_smem = sel_layout(B, pmax, dmax, k, sel_f).total;
    if (sel_smem > 48 * 1024)
        ACAV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fy_gather_select_multi),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sel_smem));
    const auto t_loop0 = std::chrono::steady_clock::now();
    for (int64_t g0 = 0; g0 < iters_max; g0 += FY_GROUP) {
        const int64_t g1 = g0 + FY_GROUP < iters_max ? g0 + FY_GROUP : iters_max;
        const unsigned gz = (unsigned)(g1 - g0);
        const int ge = (int)((g0 / FY_GROUP) % FY_DEPTH);  // event pair and buffer set of this group
        const int64_t lt = lmax - g0 * dl;           // the longest list still in play bounds the grids
        // ---- st_fy: the positions of the group's iterations (the gathers of the group FY_DEPTH back are done with this buffer set)
        if (g0 >= FY_DEPTH * FY_GROUP) ACAV_HIP_TRY(hipStreamWaitEvent(sf, lead->ev_gather[ge], 0));
        for (int c = 0; c < nchunks; ++c) {
            int64_t nd = 0;
            for (int64_t it = g0; it < g1 && it < iters[(size_t)c]; ++it) {
                const int64_t Lc = L[c] - it * dl;
                nd += Lc > 1 ? Lc - 1 : 0;
            }
            if (nd == 0) continue;
            const unsigned *unused = nullptr;
            ACAV_TRY(streams[(size_t)c].acquire(r0[(size_t)c], nd, &unused));
            r0[(size_t)c] += nd;
        }
        hipLaunchKernelGGL(part_staged ? k_fy_part_multi<true> : k_fy_part_multi<false>,
                           dim3((unsigned)((lt + FYA_CH - 1) / FYA_CH), (unsigned)nchunks, gz), dim3(FYA_THREADS), part_smem, sf, dcd,
                           (int)g0, (int)dl);
        for (int c = 0; c < nchunks; ++c)
            if (g0 < iters[(size_t)c]) ACAV_TRY(streams[(size_t)c].release(r0[(size_t)c]));  // k_fy_part is the only reader of the draws
        hipLaunchKernelGGL(k_fy_tile_multi, dim3((unsigned)ntmax, (unsigned)nchunks, gz), dim3(FYT_THREADS), tile_smem, sf, dcd, (int)g0,
                           (int)dl);
        hipLaunchKernelGGL(k_fy_resolve_multi, dim3((unsigned)((lt + 255) / 256), (unsigned)nchunks, gz), dim3(256), 0, sf, dcd, (int)g0,
                           (int)dl, keep_unselected ? B - k : 0);
        ACAV_HIP_TRY(hipEventRecord(lead->ev_tile[ge], sf));
        // ---- main stream: the group's gathers (each with the selection of the iteration before it), back to back
        ACAV_HIP_TRY(hipStreamWaitEvent(st, lead->ev_tile[ge], 0));
        for (int64_t it = g0; it < g1; ++it) {
            const dim3 grid((unsigned)((lmax - it * dl + 256 * GS_EPT - 1) / (256 * GS_EPT)) + 1u, (unsigned)nchunks);
            hipLaunchKernelGGL(k_fy_gather_select_multi, grid, dim3(256), sel_smem, st, dcd, (int)it, (int)dl, B, k, sel_f, keep_unselected);
        }
        ACAV_HIP_TRY(hipGetLastError());
        ACAV_HIP_TRY(hipEventRecord(lead->ev_gather[ge], st));
    }
    if (iters_max > 0)  // the selection of the last iteration
        hipLaunchKernelGGL(k_fy_gather_select_multi, dim3(1u, (unsigned)nchunks), dim3(256), sel_smem, st, dcd, (int)iters_max, (int)dl, B, k,
                           sel_f, keep_unselected);
    ACAV_HIP_TRY(hipGetLastError());
    const auto t_loop1 = std::chrono::steady_clock::now();
    ACAV_HIP_TRY(hipStreamSynchronize(sf));
    if (timing) {
        ACAV_HIP_TRY(hipStreamSynchronize(st));
        const auto t_loop2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[acav] greedy loop: %d chunk(s), %lld iterations, host enqueue %.2f us/iteration, enqueue + drain %.2f "
                        "us/iteration (tiles %d, cap %d, lanes %d; streams replaced by the queue probe at create: %d)\n", nchunks, (long long)iters_max,
                std::chrono::duration<double, std::micro>(t_loop1 - t_loop0).count() / (double)(iters_max ? iters_max : 1),
                std::chrono::duration<double, std::micro>(t_loop2 - t_loop0).count() / (double)(iters_max ? iters_max : 1),
                plans[0].NT, plans[0].ecap, streams[0].W, mis[0]->queue_probe_replaced);
    }
    const auto t_tail0 = clk::now();
    for (int c = 0; c < nchunks; ++c) {
        acav_mi *mi = mis[c];
        const int64_t itc = iters[(size_t)c];
        const int64_t nsel = itc * k < subset[c] ? itc * k : subset[c];
        if (itc > 0) {
            ACAV_HIP_TRY(hipMemcpyAsync(S_out[c], mi->S.p, sizeof(long long) * (size_t)nsel, hipMemcpyDeviceToHost, st));
            ACAV_HIP_TRY(hipMemcpyAsync(GAIN_out[c], mi->G.p, sizeof(double) * (size_t)(itc * k), hipMemcpyDeviceToHost, st));
            if (ex.trace_pos)
                ACAV_HIP_TRY(hipMemcpyAsync(ex.trace_pos, mi->tr_pos.p, sizeof(int) * (size_t)(itc * k), hipMemcpyDeviceToHost, st));
            if (ex.trace_ids)
                ACAV_HIP_TRY(hipMemcpyAsync(ex.trace_ids, mi->tr_ids.p, sizeof(long long) * (size_t)(itc * B), hipMemcpyDeviceToHost, st));
            if (ex.trace_scores)
                ACAV_HIP_TRY(hipMemcpyAsync(ex.trace_scores, mi->tr_sc.p, sizeof(double) * (size_t)(itc * B), hipMemcpyDeviceToHost, st));
        }
        if (n_selected) n_selected[c] = nsel;
        if (n_iters) n_iters[c] = itc;
    }
    ACAV_HIP_TRY(hipStreamSynchronize(st));
    for (int c = 0; c < nchunks; ++c) {  // every generator continues on the host where its chunk stopped drawing
        ACAV_HIP_TRY(hipStreamSynchronize(streams[(size_t)c].smt ? streams[(size_t)c].smt : mis[c]->st_mt));
        unsigned mtbuf[625];
        int idx = 0;
        ACAV_TRY(acav_rng_get_state(rngs[c], mtbuf, &idx));
        ACAV_TRY(streams[(size_t)c].final_state(mtbuf, &idx));
        ACAV_TRY(acav_rng_set_state(rngs[c], mtbuf, idx));
        unsigned ferr = 0;
        ACAV_HIP_TRY(hipMemcpy(&ferr, mis[c]->fy_err.p, sizeof(ferr), hipMemcpyDeviceToHost));
        ACAV_REQUIRE(ferr == 0, ACAV_ESTATE, "tiled Fisher-Yates: a tile bucket of chunk %d overflowed (flags %u); re-run with "
                                             "ACAV_FY_LEGACY=1", c, ferr);
    }
    if (timing)
        fprintf(stderr, "[acav]   set-up %.1f ms (per-chunk work %.1f: add_samples %.1f, ids %.1f, tables %.1f, generator plan %.1f; then the sync), "
                        "read-back + generator states %.1f ms\n", t_setup, t_chunks, t_add, t_ids, t_fy, t_mt, ms_since(t_tail0));
    return ACAV_OK;
}

ACAV_EXPORT int acav_mi_run_greedy_multi(acav_mi **mis, int nchunks, const int64_t *const *candidates, const int64_t *L,
                                         const int64_t *const *start, const int *ns, const int64_t *subset, int B, int k,
                                         int keep_unselected, acav_rng **rngs, int64_t *const *S_out,
                                         double *const *GAIN_out, int64_t *n_selected, int64_t *n_iters)
{
    ACAV_REQUIRE(mis && nchunks > 0 && candidates && L && ns && subset && rngs && S_out && GAIN_out, ACAV_EINVAL,
                 "NULL argument");
    ACAV_REQUIRE(B > 0 && B <= SEL_MAXB && k > 0 && k <= B, ACAV_EINVAL, "batch_size %d / selection_size %d out of range", B,
                 k);
    acav_mi *lead = mis[0];
    ACAV_REQUIRE(lead, ACAV_EINVAL, "handle is NULL");
    ACAV_HIP_TRY(hipSetDevice(lead->ctx.device));
    {   // validation common to both evaluations, then the tiled one unless a list is too long for it (or ACAV_FY_LEGACY=1)
        const char *legacy = getenv("ACAV_FY_LEGACY");
        bool tiled = !(legacy && legacy[0] == '1');
        for (int c = 0; c < nchunks; ++c) {
            acav_mi *mi = mis[c];
            ACAV_REQUIRE(mi && candidates[c] && rngs[c] && S_out[c] && GAIN_out[c], ACAV_EINVAL, "chunk %d: NULL argument", c);
            ACAV_REQUIRE(mi->ctx.device == lead->ctx.device, ACAV_EINVAL, "chunk %d lives on another device", c);
            ACAV_REQUIRE(L[c] > 0 && L[c] <= mi->V && ns[c] >= 0 && (ns[c] == 0 || (start && start[c])) && subset[c] >= 0,
                         ACAV_EINVAL, "chunk %d: bad sizes", c);
            ACAV_REQUIRE((int64_t)B * mi->P <= SEL_MAXBP, ACAV_EINVAL, "chunk %d: B*P exceeds %d", c, SEL_MAXBP);
            for (int e = 0; e < c; ++e)
                ACAV_REQUIRE(mis[e] != mi && rngs[e] != rngs[c], ACAV_EINVAL, "chunks must not share a handle or a generator");
            tiled = tiled && L[c] <= FY_TILED_MAX;
        }
        if (tiled)
            return run_greedy_tiled(mis, nchunks, candidates, L, start, ns, subset, B, k, keep_unselected, rngs, S_out, GAIN_out,
                                    n_selected, n_iters, TiledExtras());
    }
    ACAV_TRY(mi_ensure_streams(lead));
    hipStream_t st = lead->ctx.stream, smt = lead->st_mt;
    const int64_t dl = B - (keep_unselected ? B - k : 0);
    std::vector<ChunkDesc> desc((size_t)nchunks);
    std::vector<int64_t> iters((size_t)nchunks, 0);
    int64_t iters_max = 0, lmax = 0;
    int pmax = 1, dmax = 1;
    for (int c = 0; c < nchunks; ++c) {
        acav_mi *mi = mis[c];
        ACAV_REQUIRE(mi && candidates[c] && rngs[c] && S_out[c] && GAIN_out[c], ACAV_EINVAL, "chunk %d: NULL argument", c);
        ACAV_REQUIRE(mi->ctx.device == lead->ctx.device, ACAV_EINVAL, "chunk %d lives on another device", c);
        ACAV_REQUIRE(L[c] > 0 && L[c] <= mi->V && ns[c] >= 0 && (ns[c] == 0 || (start && start[c])) && subset[c] >= 0,
                     ACAV_EINVAL, "chunk %d: bad sizes", c);
        ACAV_REQUIRE((int64_t)B * mi->P <= SEL_MAXBP, ACAV_EINVAL, "chunk %d: B*P exceeds %d", c, SEL_MAXBP);
        for (int e = 0; e < c; ++e) ACAV_REQUIRE(mis[e] != mi && rngs[e] != rngs[c], ACAV_EINVAL, "chunks must not share a handle or a generator");
        ACAV_HIP_TRY(hipStreamSynchronize(mi->ctx.stream));  // whatever the handle was doing on its own stream is over
        if (ns[c]) ACAV_TRY(acav_mi_add_samples(mi, start[c], ns[c]));
        int64_t nS = 0, l = L[c], itc = 0;
        while (nS < subset[c]) {
            ACAV_REQUIRE(l >= B, ACAV_ERANGE, "chunk %d: %lld candidates left < batch_size %d (batch.py:143-150)", c,
                         (long long)l, B);
            nS += k;
            l -= dl;
            ++itc;
        }
        iters[(size_t)c] = itc;
        iters_max = itc > iters_max ? itc : iters_max;
        lmax = L[c] > lmax ? L[c] : lmax;
        pmax = mi->P > pmax ? mi->P : pmax;
        dmax = mi->D > dmax ? mi->D : dmax;
        const size_t Lc = (size_t)L[c];
        hipStream_t sc = mi->ctx.stream;
        ACAV_TRY(mi->A0.ensure(sizeof(int) * (Lc + B)));
        ACAV_TRY(ids_to_device32(mi, candidates[c], L[c], mi->stage, mi->A0));
        ACAV_TRY(mi->A1.ensure(sizeof(int) * (Lc + B)));
        ACAV_TRY(mi->draws.ensure(sizeof(unsigned) * (Lc * MT_GROUP + MT_PAD)));
        ACAV_TRY(mi->draws2.ensure(sizeof(unsigned) * (Lc * MT_GROUP + MT_PAD)));
        ACAV_TRY(mi->h.ensure(sizeof(int) * Lc));
        ACAV_TRY(mi->head.ensure(sizeof(int) * Lc));
        ACAV_TRY(mi->next.ensure(sizeof(int) * Lc));
        ACAV_TRY(mi->g.ensure(sizeof(int) * Lc));
        ACAV_TRY(mi->head2.ensure(sizeof(int) * Lc));
        ACAV_TRY(mi->g2.ensure(sizeof(int) * Lc));
        ACAV_HIP_TRY(hipMemsetAsync(mi->head.p, 0xFF, sizeof(int) * Lc, sc));
        ACAV_HIP_TRY(hipMemsetAsync(mi->g.p, 0xFF, sizeof(int) * Lc, sc));
        ACAV_TRY(mi->mt.ensure(sizeof(unsigned) * 625));
        ACAV_TRY(mi->batch.ensure(sizeof(int) * SEL_MAXB));
        ACAV_TRY(mi->S.ensure(sizeof(long long) * (size_t)(itc * k + 1)));
        ACAV_TRY(mi->G.ensure(sizeof(double) * (size_t)(itc * k + 1)));
        unsigned mtbuf[625];
        int idx = 0;
        ACAV_TRY(acav_rng_get_state(rngs[c], mtbuf, &idx));
        mtbuf[624] = (unsigned)idx;
        ACAV_HIP_TRY(hipMemcpyAsync(mi->mt.p, mtbuf, sizeof(mtbuf), hipMemcpyHostToDevice, sc));
        ACAV_HIP_TRY(hipStreamSynchronize(sc));  // mtbuf is a local; the lead's streams take over from here
        ChunkDesc &d = desc[(size_t)c];
        d.asg = mi->asg.as<int>(), d.pairs = mi->pairs.as<int>();
        d.Nc = mi->Nc.as<int>(), d.ac = mi->ac.as<int>(), d.bc = mi->bc.as<int>();
        d.SN = mi->SN.as<double>(), d.Sa = mi->Sa.as<double>(), d.Sb = mi->Sb.as<double>();
        d.phi = mi->phi.as<double>(), d.sc = mi->scalars.as<MiScalars>();
        d.A[0] = mi->A0.as<int>(), d.A[1] = mi->A1.as<int>();
        d.draws[0] = mi->draws.as<unsigned>(), d.draws[1] = mi->draws2.as<unsigned>();
        d.h = mi->h.as<int>(), d.next = mi->next.as<int>();
        d.head[0] = mi->head.as<int>(), d.head[1] = mi->head2.as<int>();
        d.g[0] = mi->g.as<int>(), d.g[1] = mi->g2.as<int>();
        d.mt = mi->mt.as<unsigned>(), d.batch = mi->batch.as<int>();
        d.S = mi->S.as<long long>(), d.G = mi->G.as<double>();
        d.D = mi->D, d.C = mi->C, d.P = mi->P, d.L0 = (int)L[c], d.iters = (int)itc, d.pad = 0;
    }
    ACAV_TRY(lead->chunk_desc.ensure(sizeof(ChunkDesc) * (size_t)nchunks));
    ACAV_HIP_TRY(hipMemcpyAsync(lead->chunk_desc.p, desc.data(), sizeof(ChunkDesc) * (size_t)nchunks, hipMemcpyHostToDevice, st));
    ACAV_HIP_TRY(hipStreamSynchronize(st));  // desc is a local
    const ChunkDesc *dcd = lead->chunk_desc.as<ChunkDesc>();
    const int64_t ngroups = (iters_max + MT_GROUP - 1) / MT_GROUP;
    ACAV_HIP_TRY(hipEventRecord(lead->ev_used[0], st));
    ACAV_HIP_TRY(hipStreamWaitEvent(smt, lead->ev_used[0], 0));
    auto launch_mt = [&](int64_t g_) -> int {
        const int cur_ = (int)(g_ & 1);
        if (g_ >= 2) ACAV_HIP_TRY(hipStreamWaitEvent(smt, lead->ev_used[cur_], 0));  // the readers of group g_-2 are done
        hipLaunchKernelGGL(k_mt_generate_multi, dim3((unsigned)nchunks), dim3(MT_THREADS), 0, smt, dcd, (int)g_, (int)dl);
        ACAV_HIP_TRY(hipEventRecord(lead->ev_mt[cur_], smt));
        return ACAV_OK;
    };
    if (iters_max > 0) ACAV_TRY(launch_mt(0));
    const int sel_f = sel_mode(B, pmax, k);
    const size_t smem = sel_layout(B, pmax, dmax, k, sel_f).total;
    if (smem > 48 * 1024)
        ACAV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mi_select_multi), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)smem));
    for (int64_t it = 0; it < iters_max; ++it) {
        const int64_t grp = it / MT_GROUP;
        const int cur = (int)(grp & 1);
        if (it % MT_GROUP == 0) {
            if (grp + 1 < ngroups) ACAV_TRY(launch_mt(grp + 1));
            ACAV_HIP_TRY(hipStreamWaitEvent(st, lead->ev_mt[cur], 0));
        }
        const int64_t lt = lmax - it * dl;  // the longest list still in play bounds the grid
        const dim3 grid((unsigned)((lt + 255) / 256), (unsigned)nchunks);
        hipLaunchKernelGGL(k_fy_build_multi, grid, dim3(256), 0, st, dcd, (int)it, (int)dl);
        if (it % MT_GROUP == MT_GROUP - 1 || it + 1 == iters_max) ACAV_HIP_TRY(hipEventRecord(lead->ev_used[cur], st));
        hipLaunchKernelGGL(k_fy_apply_multi, grid, dim3(256), 0, st, dcd, (int)it, (int)dl, B);
        hipLaunchKernelGGL(k_mi_select_multi, dim3((unsigned)nchunks), dim3(256), smem, st, dcd, (int)it, (int)dl, B, k, sel_f,
                           keep_unselected);
    }
    ACAV_HIP_TRY(hipGetLastError());
    ACAV_HIP_TRY(hipStreamSynchronize(smt));
    for (int c = 0; c < nchunks; ++c) {
        acav_mi *mi = mis[c];
        const int64_t itc = iters[(size_t)c];
        const int64_t nsel = itc * k < subset[c] ? itc * k : subset[c];
        if (itc > 0) {
            ACAV_HIP_TRY(hipMemcpyAsync(S_out[c], mi->S.p, sizeof(long long) * (size_t)nsel, hipMemcpyDeviceToHost, st));
            ACAV_HIP_TRY(hipMemcpyAsync(GAIN_out[c], mi->G.p, sizeof(double) * (size_t)(itc * k), hipMemcpyDeviceToHost, st));
        }
        if (n_selected) n_selected[c] = nsel;
        if (n_iters) n_iters[c] = itc;
    }
    ACAV_HIP_TRY(hipStreamSynchronize(st));
    for (int c = 0; c < nchunks; ++c) {  // every generator continues on the host where its chunk stopped drawing
        unsigned mtbuf[625];
        ACAV_HIP_TRY(hipMemcpy(mtbuf, mis[c]->mt.p, sizeof(mtbuf), hipMemcpyDeviceToHost));
        ACAV_TRY(acav_rng_set_state(rngs[c], mtbuf, (int)mtbuf[624]));
    }
    return ACAV_OK;
}

// which score the exact greedy (acav_mi_run_exact) maximises: 0 = calc_MI ('mi', 'mem_mi'; mi.py:85-91), 1 = calc_AMI ('ami',
// mi.py:212-259), 2 = calc_NMI (EfficientNMI, mi.py:262-271), 3 = ConstantMeasure (mi.py:274-281).  The adjusted and normalised
// scores read two more host-built tables, ln k and ln k! for k <= V + 1.
ACAV_EXPORT int acav_mi_set_measure(acav_mi *mi, int measure)
{
    ACAV_REQUIRE(mi, ACAV_EINVAL, "handle is NULL");
    ACAV_REQUIRE(measure >= 0 && measure <= 3, ACAV_EINVAL, "unknown measure %d", measure);
    ACAV_REQUIRE(measure != 2 || (int64_t)mi->C <= mi->V + 1, ACAV_EINVAL, "nmi: ncentroids %d exceeds the ln k table (V + 1 = %lld)",
                 mi->C, (long long)(mi->V + 1));
    if ((measure == 1 || measure == 2) && !mi->lnk.p) {
        ACAV_HIP_TRY(hipSetDevice(mi->ctx.device));
        std::vector<double> lnk((size_t)mi->V + 2), lf((size_t)mi->V + 2);
        lnk[0] = 0.0, lf[0] = 0.0;
        for (int64_t k = 1; k < mi->V + 2; ++k) lnk[(size_t)k] = log((double)k), lf[(size_t)k] = lgamma((double)k + 1.0);
        ACAV_TRY(mi->lnk.ensure(sizeof(double) * lnk.size()));
        ACAV_TRY(mi->lf.ensure(sizeof(double) * lf.size()));
        ACAV_HIP_TRY(hipMemcpyAsync(mi->lnk.p, lnk.data(), sizeof(double) * lnk.size(), hipMemcpyHostToDevice, mi->ctx.stream));
        ACAV_HIP_TRY(hipMemcpyAsync(mi->lf.p, lf.data(), sizeof(double) * lf.size(), hipMemcpyHostToDevice, mi->ctx.stream));
        ACAV_HIP_TRY(hipStreamSynchronize(mi->ctx.stream));  // the vectors are locals
    }
    mi->measure = measure;
    return ACAV_OK;
}

ACAV_EXPORT int acav_mi_run_exact(acav_mi *mi, const int64_t *candidates, int64_t L, int ns, int64_t subset,
                                  int64_t *S_out, double *GAIN_out, int64_t *n_selected, const int64_t *forced_pos,
                                  double *trace_scores, int64_t *trace_argmax)
{
    ACAV_REQUIRE(mi && candidates && S_out && GAIN_out && n_selected, ACAV_EINVAL, "NULL argument");
    ACAV_REQUIRE(L > 0 && L < 0x7fffffff && ns >= 0, ACAV_EINVAL, "bad candidate count %lld", (long long)L);
    ACAV_HIP_TRY(hipSetDevice(mi->ctx.device));
    hipStream_t st = mi->ctx.stream;
    int64_t iters = subset - 1 - ns;  // range(len(start_indices), subset_size - 1)   (mi.py:161)
    if (iters < 0) iters = 0;
    if (iters > L) iters = L;
    *n_selected = iters;
    if (iters == 0) return ACAV_OK;
    for (int64_t i = 0; i < L; ++i)
        ACAV_REQUIRE(candidates[i] >= 0 && candidates[i] < mi->V, ACAV_EINVAL, "candidate id %lld outside [0, %lld)",
                     (long long)candidates[i], (long long)mi->V);
    if (forced_pos)
        for (int64_t i = 0; i < iters; ++i)
            ACAV_REQUIRE(forced_pos[i] >= 0 && forced_pos[i] < L, ACAV_EINVAL, "forced position out of range");
    const unsigned grid = (unsigned)((L + 255) / 256);
    ACAV_TRY(mi->A0.ensure(sizeof(int) * (size_t)L));
    ACAV_TRY(ids_to_device32(mi, candidates, L, mi->stage, mi->A0));
    ACAV_TRY(mi->removed.ensure((size_t)L));
    ACAV_TRY(mi->blockbest.ensure(sizeof(ExactBest) * grid));
    ACAV_TRY(mi->ticket.ensure(sizeof(unsigned)));
    ACAV_TRY(mi->S.ensure(sizeof(long long) * (size_t)iters));
    ACAV_TRY(mi->G.ensure(sizeof(double) * (size_t)iters));
    ACAV_HIP_TRY(hipMemsetAsync(mi->removed.p, 0, (size_t)L, st));
    ACAV_HIP_TRY(hipMemsetAsync(mi->ticket.p, 0, sizeof(unsigned), st));
    if (forced_pos) {
        std::vector<int> f32((size_t)iters);
        for (int64_t i = 0; i < iters; ++i) f32[(size_t)i] = (int)forced_pos[i];
        ACAV_TRY(mi->forced.ensure(sizeof(int) * (size_t)iters));
        ACAV_HIP_TRY(hipMemcpyAsync(mi->forced.p, f32.data(), sizeof(int) * (size_t)iters, hipMemcpyHostToDevice, st));
        ACAV_HIP_TRY(hipStreamSynchronize(st));  // f32 is a local
    }
    if (trace_scores) ACAV_TRY(mi->tr_sc.ensure(sizeof(double) * (size_t)iters * (size_t)L));
    if (trace_argmax) ACAV_TRY(mi->tr_am.ensure(sizeof(int) * (size_t)iters));
    for (int64_t it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(k_mi_exact_iter, dim3(grid), dim3(256), 0, st, mi->asg.as<int>(), mi->D, mi->C, mi->P,
                           mi->pairs.as<int>(), mi->A0.as<int>(), (int)L, mi->removed.as<unsigned char>(), mi->Nc.as<int>(),
                           mi->ac.as<int>(), mi->bc.as<int>(), mi->SN.as<double>(), mi->Sa.as<double>(), mi->Sb.as<double>(),
                           mi->phi.as<double>(), mi->scalars.as<MiScalars>(), mi->blockbest.as<ExactBest>(),
                           mi->ticket.as<unsigned>(), mi->S.as<long long>() + it, mi->G.as<double>() + it,
                           forced_pos ? mi->forced.as<int>() + it : nullptr,
                           trace_scores ? mi->tr_sc.as<double>() + (size_t)it * (size_t)L : nullptr,
                           trace_argmax ? mi->tr_am.as<int>() + it : nullptr, mi->measure, mi->lnk.as<double>(), mi->lf.as<double>());
    }
    ACAV_HIP_TRY(hipGetLastError());
    ACAV_HIP_TRY(hipMemcpyAsync(S_out, mi->S.p, sizeof(long long) * (size_t)iters, hipMemcpyDeviceToHost, st));
    ACAV_HIP_TRY(hipMemcpyAsync(GAIN_out, mi->G.p, sizeof(double) * (size_t)iters, hipMemcpyDeviceToHost, st));
    if (trace_scores)
        ACAV_HIP_TRY(hipMemcpyAsync(trace_scores, mi->tr_sc.p, sizeof(double) * (size_t)iters * (size_t)L,
                                    hipMemcpyDeviceToHost, st));
    std::vector<int> am;
    if (trace_argmax) {
        am.resize((size_t)iters);
        ACAV_HIP_TRY(hipMemcpyAsync(am.data(), mi->tr_am.p, sizeof(int) * (size_t)iters, hipMemcpyDeviceToHost, st));
    }
    ACAV_HIP_TRY(hipStreamSynchronize(st));
    if (trace_argmax)
        for (int64_t i = 0; i < iters; ++i) trace_argmax[i] = am[(size_t)i];
    return ACAV_OK;
}

ACAV_EXPORT int acav_mi_sync(acav_mi *mi)
{
    ACAV_REQUIRE(mi, ACAV_EINVAL, "handle is NULL");
    ACAV_HIP_TRY(hipStreamSynchronize(mi->ctx.stream));
    return ACAV_OK;
}
ACAV_EXPORT int acav_mi_timer_begin(acav_mi *mi)
{
    ACAV_REQUIRE(mi, ACAV_EINVAL, "handle is NULL");
    return mi->ctx.timer_begin();
}
ACAV_EXPORT int acav_mi_timer_end(acav_mi *mi, float *ms)
{
    ACAV_REQUIRE(mi && ms, ACAV_EINVAL, "NULL argument");
    return mi->ctx.timer_end(ms);
}

ACAV_EXPORT int acav_mi_add_samples(acav_mi *mi, const int64_t *ids, int64_t n)
{
    ACAV_REQUIRE(mi && (ids || n == 0) && n >= 0 && n < 0x7fffffff, ACAV_EINVAL, "bad argument");
    if (n == 0) return ACAV_OK;
    ACAV_HIP_TRY(hipSetDevice(mi->ctx.device));
    ACAV_TRY(ids_to_device32(mi, ids, n, mi->stage, mi->ids32));
    hipLaunchKernelGGL(k_mi_commit, dim3(1), dim3(256), 0, mi->ctx.stream, mi->asg.as<int>(), mi->D, mi->C, mi->P,
                       mi->pairs.as<int>(), mi->ids32.as<int>(), (int)n, mi->Nc.as<int>(), mi->ac.as<int>(),
                       mi->bc.as<int>(), mi->SN.as<double>(), mi->Sa.as<double>(), mi->Sb.as<double>(),
                       mi->phi.as<double>(), mi->scalars.as<MiScalars>());
    ACAV_HIP_TRY(hipGetLastError());
    ACAV_HIP_TRY(hipStreamSynchronize(mi->ctx.stream));
    return ACAV_OK;
}

static int launch_select(acav_mi *mi, const int *batch, int B, int k, double *scores_out, long long *S_out,
                         double *G_out, const int *forced_pos, int *trace_pos, long long *trace_ids,
                         double *trace_scores, int keep, int *requeue_out)
{
    const int mode = sel_mode(B, mi->P, k);
    const size_t smem = sel_layout(B, mi->P, mi->D, k, mode).total;
    if (smem > 48 * 1024)
        ACAV_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_mi_select), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)smem));
    hipLaunchKernelGGL(k_mi_select, dim3(1), dim3(256), smem, mi->ctx.stream, mi->asg.as<int>(), mi->D, mi->C, mi->P,
                       mi->pairs.as<int>(), batch, B, k, mode, mi->Nc.as<int>(), mi->ac.as<int>(), mi->bc.as<int>(),
                       mi->SN.as<double>(), mi->Sa.as<double>(), mi->Sb.as<double>(), mi->phi.as<double>(),
                       mi->scalars.as<MiScalars>(), scores_out, S_out, G_out, forced_pos, trace_pos, trace_ids,
                       trace_scores, keep, requeue_out);
    ACAV_HIP_TRY(hipGetLastError());
    return ACAV_OK;
}

ACAV_EXPORT int acav_mi_score_batch(acav_mi *mi, const int64_t *ids, int B, double *scores_host)
{
    ACAV_REQUIRE(mi && ids && scores_host, ACAV_EINVAL, "NULL argument");
    ACAV_REQUIRE(B > 0 && B <= SEL_MAXB && (int64_t)B * mi->P <= SEL_MAXBP, ACAV_EINVAL,
                 "batch %d x pairs %d outside the supported range (B<=%d, B*P<=%d)", B, mi->P, SEL_MAXB, SEL_MAXBP);
    ACAV_HIP_TRY(hipSetDevice(mi->ctx.device));
    ACAV_TRY(ids_to_device32(mi, ids, B, mi->stage, mi->ids32));
    ACAV_TRY(mi->scores.ensure(sizeof(double) * SEL_MAXB));
    ACAV_TRY(launch_select(mi, mi->ids32.as<int>(), B, 0, mi->scores.as<double>(), nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, 0, nullptr));
    ACAV_TRY(from_device(scores_host, mi->scores.p, sizeof(double) * (size_t)B, mi->ctx.stream));
    ACAV_HIP_TRY(hipStreamSynchronize(mi->ctx.stream));
    return ACAV_OK;
}

ACAV_EXPORT int acav_mi_get_counts(acav_mi *mi, int32_t *N, int32_t *a, int32_t *b, int64_t *n)
{
    ACAV_REQUIRE(mi, ACAV_EINVAL, "handle is NULL");
    ACAV_HIP_TRY(hipSetDevice(mi->ctx.device));
    hipStream_t st = mi->ctx.stream;
    const size_t cc = (size_t)mi->P * mi->C * mi->C, pc = (size_t)mi->P * mi->C;
    if (N) ACAV_TRY(from_device(N, mi->Nc.p, sizeof(int) * cc, st));
    if (a) ACAV_TRY(from_device(a, mi->ac.p, sizeof(int) * pc, st));
    if (b) ACAV_TRY(from_device(b, mi->bc.p, sizeof(int) * pc, st));
    MiScalars s{};
    ACAV_HIP_TRY(hipMemcpyAsync(&s, mi->scalars.p, sizeof(s), hipMemcpyDeviceToHost, st));
    ACAV_HIP_TRY(hipStreamSynchronize(st));
    if (n) *n = (int64_t)s.nc;
    return ACAV_OK;
}

ACAV_EXPORT int acav_mi_run_greedy(acav_mi *mi, const int64_t *candidates, int64_t L, const int64_t *start, int ns,
                                   int64_t subset, int B, int k, int keep_unselected, acav_rng *rng,
                                   int64_t *S_out, double *GAIN_out, int64_t *n_selected, int64_t *n_iters,
                                   int64_t *trace_ids, double *trace_scores, int32_t *trace_pos,
                                   const int32_t *forced_pos, int64_t max_iters)
{
    ACAV_REQUIRE(mi && candidates && rng && S_out && GAIN_out, ACAV_EINVAL, "NULL argument");
    ACAV_REQUIRE(L > 0 && L <= mi->V && ns >= 0 && (start || ns == 0) && subset >= 0, ACAV_EINVAL, "bad sizes");
    ACAV_REQUIRE(B > 0 && B <= SEL_MAXB && k > 0 && k <= B && (int64_t)B * mi->P <= SEL_MAXBP, ACAV_EINVAL,
                 "batch_size %d / selection_size %d / pairs %d outside the supported range (B<=%d, B*P<=%d)", B, k,
                 mi->P, SEL_MAXB, SEL_MAXBP);
    ACAV_HIP_TRY(hipSetDevice(mi->ctx.device));
    // the permutation of every iteration: tiled evaluation (all atomics in LDS; run_greedy_tiled) unless the list is too long
    // for its tile table, or ACAV_FY_LEGACY=1 asks for the global-atomic kernels (k_fy_build / k_fy_apply) below
    const char *legacy = getenv("ACAV_FY_LEGACY");
    if (L <= FY_TILED_MAX && !(legacy && legacy[0] == '1')) {
        TiledExtras ex;
        ex.trace_ids = trace_ids, ex.trace_scores = trace_scores, ex.trace_pos = trace_pos, ex.forced_pos = forced_pos;
        ex.max_iters = max_iters;
        return run_greedy_tiled(&mi, 1, &candidates, &L, &start, &ns, &subset, B, k, keep_unselected, &rng, &S_out, &GAIN_out,
                                n_selected, n_iters, ex);
    }
    hipStream_t st = mi->ctx.stream;
    if (ns) ACAV_TRY(acav_mi_add_samples(mi, start, ns));  // batch.py:215

    // plan: the number of iterations and every L_t are known on the host (no device feedback)
    int64_t iters = 0;
    {
        int64_t nS = 0, l = L;
        while (nS < subset && (max_iters < 0 || iters < max_iters)) {
            ACAV_REQUIRE(l >= B, ACAV_ERANGE,
                         "%lld candidates left < batch_size %d: the reference's topk(k=floor(B/k*B')) raises here "
                         "(batch.py:143-150)", (long long)l, B);
            nS += k;
            l = l - B + (keep_unselected ? B - k : 0);
            ++iters;
        }
    }
    const int64_t cap = iters * k + 1;
    ACAV_TRY(mi->A0.ensure(sizeof(int) * (size_t)(L + B)));  // before the conversion: ensure() does not copy
    ACAV_TRY(ids_to_device32(mi, candidates, L, mi->stage, mi->A0));
    ACAV_TRY(mi->A1.ensure(sizeof(int) * (size_t)(L + B)));
    ACAV_TRY(mi->h.ensure(sizeof(int) * (size_t)L));
    ACAV_TRY(mi->head.ensure(sizeof(int) * (size_t)L));
    ACAV_TRY(mi->next.ensure(sizeof(int) * (size_t)L));
    ACAV_TRY(mi->g.ensure(sizeof(int) * (size_t)L));
    ACAV_TRY(mi->head2.ensure(sizeof(int) * (size_t)L));
    ACAV_TRY(mi->g2.ensure(sizeof(int) * (size_t)L));
    ACAV_HIP_TRY(hipMemsetAsync(mi->head.p, 0xFF, sizeof(int) * (size_t)L, st));
    ACAV_HIP_TRY(hipMemsetAsync(mi->g.p, 0xFF, sizeof(int) * (size_t)L, st));
    ACAV_TRY(mi->mt.ensure(sizeof(unsigned) * 625));
    ACAV_TRY(mi->batch.ensure(sizeof(int) * SEL_MAXB));
    ACAV_TRY(mi->S.ensure(sizeof(long long) * (size_t)cap));
    ACAV_TRY(mi->G.ensure(sizeof(double) * (size_t)cap));
    if (trace_pos) ACAV_TRY(mi->tr_pos.ensure(sizeof(int) * (size_t)(iters * k + 1)));
    if (trace_ids) ACAV_TRY(mi->tr_ids.ensure(sizeof(long long) * (size_t)(iters * B + 1)));
    if (trace_scores) ACAV_TRY(mi->tr_sc.ensure(sizeof(double) * (size_t)(iters * B + 1)));
    if (forced_pos) {
        for (int64_t i = 0; i < iters * k; ++i)
            ACAV_REQUIRE(forced_pos[i] >= 0 && forced_pos[i] < B, ACAV_EINVAL, "forced position out of range");
        ACAV_TRY(mi->forced.ensure(sizeof(int) * (size_t)(iters * k + 1)));
        ACAV_HIP_TRY(hipMemcpyAsync(mi->forced.p, forced_pos, sizeof(int) * (size_t)(iters * k), hipMemcpyHostToDevice, st));
    }
    // hand the host MT19937 stream to the device: W lanes generate it superblock by superblock on their own stream
    // (MtStream); nothing they do depends on what gets selected (L shrinks by a fixed amount per iteration)
    unsigned mtbuf[625];
    int idx = 0;
    ACAV_TRY(acav_rng_get_state(rng, mtbuf, &idx));
    const int64_t dl = B - (keep_unselected ? B - k : 0);  // candidates consumed per iteration
    int64_t total_draws = 0;
    for (int64_t t = 0; t < iters; ++t) {
        const int64_t lt = L - t * dl;
        total_draws += lt > 1 ? lt - 1 : 0;
    }
    MtStream ms;
    ACAV_TRY(ms.plan(mi, st, mtbuf, idx, total_draws, L, L));

    int *Acur = mi->A0.as<int>(), *Anew = mi->A1.as<int>();
    int64_t l = L;
    int64_t r0 = 0;  // first draw of this iteration, counted from the first draw of the run
    for (int64_t it = 0; it < iters; ++it) {
        const int Li = (int)l;
        const int64_t nd = Li > 1 ? Li - 1 : 0;
        const unsigned grid = (unsigned)((Li + 255) / 256);
        const unsigned *draws = nullptr;
        ACAV_TRY(ms.acquire(r0, nd, &draws));
        int *hd = (it & 1) ? mi->head2.as<int>() : mi->head.as<int>();
        int *gg = (it & 1) ? mi->g2.as<int>() : mi->g.as<int>();
        int *hd_n = (it & 1) ? mi->head.as<int>() : mi->head2.as<int>();
        int *gg_n = (it & 1) ? mi->g.as<int>() : mi->g2.as<int>();
        hipLaunchKernelGGL(k_fy_build, dim3(grid), dim3(256), 0, st, draws, Li, mi->h.as<int>(), hd,
                           mi->next.as<int>(), gg);
        r0 += nd;
        ACAV_TRY(ms.release(r0));  // k_fy_build is the only reader of the draws
        hipLaunchKernelGGL(k_fy_apply, dim3(grid), dim3(256), 0, st, Acur, Li, B, mi->h.as<int>(), hd,
                           mi->next.as<int>(), gg, mi->batch.as<int>(), Anew, hd_n, gg_n);
        ACAV_HIP_TRY(hipGetLastError());
        ACAV_TRY(launch_select(mi, mi->batch.as<int>(), B, k, nullptr, mi->S.as<long long>() + it * k,
                               mi->G.as<double>() + it * k, forced_pos ? mi->forced.as<int>() + it * k : nullptr,
                               trace_pos ? mi->tr_pos.as<int>() + it * k : nullptr,
                               trace_ids ? mi->tr_ids.as<long long>() + it * B : nullptr,
                               trace_scores ? mi->tr_sc.as<double>() + it * B : nullptr, keep_unselected,
                               Anew + (Li - B)));
        l = l - B + (keep_unselected ? B - k : 0);
        int *t = Acur;
        Acur = Anew;
        Anew = t;
    }
    const int64_t nsel = iters * k < subset ? iters * k : subset;
    if (iters > 0) {
        ACAV_HIP_TRY(hipMemcpyAsync(S_out, mi->S.p, sizeof(long long) * (size_t)nsel, hipMemcpyDeviceToHost, st));
        ACAV_HIP_TRY(hipMemcpyAsync(GAIN_out, mi->G.p, sizeof(double) * (size_t)(iters * k), hipMemcpyDeviceToHost, st));
        if (trace_pos) ACAV_HIP_TRY(hipMemcpyAsync(trace_pos, mi->tr_pos.p, sizeof(int) * (size_t)(iters * k), hipMemcpyDeviceToHost, st));
        if (trace_ids) ACAV_HIP_TRY(hipMemcpyAsync(trace_ids, mi->tr_ids.p, sizeof(long long) * (size_t)(iters * B), hipMemcpyDeviceToHost, st));
        if (trace_scores) ACAV_HIP_TRY(hipMemcpyAsync(trace_scores, mi->tr_sc.p, sizeof(double) * (size_t)(iters * B), hipMemcpyDeviceToHost, st));
    }
    if (mi->st_mt) ACAV_HIP_TRY(hipStreamSynchronize(mi->st_mt));
    ACAV_HIP_TRY(hipStreamSynchronize(st));
    ACAV_TRY(ms.final_state(mtbuf, &idx));
    ACAV_TRY(acav_rng_set_state(rng, mtbuf, idx));  // the stream continues on the host
    if (n_selected) *n_selected = nsel;
    if (n_iters) *n_iters = iters;
    return ACAV_OK;
}
