// acav_contrastive.hip -- the reference's contrastive baseline measure on gfx950.
// Reference: subset_selection/code/measures/contrastive/module.py:9-98 (two linear projections, F.normalize, symmetric
// InfoNCE at temperature 0.1, infer = cosine of the aligned audio / visual pair) and contrastive.py:27-132 (AdamW with
// amsgrad, eps 1e-6, the default weight decay 0.01; the train loop never zeroes the gradients, so `.grad` accumulates over
// every batch of every epoch -- reproduced).  fp32 throughout; parity with the reference is 1e-4 relative (its GEMMs sum
// in MKL's order), pinned through oracle/contrastive_ref.py and tests/golden/contrastive_*.npz.
//
// One training batch (B <= 256 clips) is a chain of small dependent kernels -- three GEMM shapes, row statistics of the
// B x B logits, the normalisation's backward, one fused AdamW over the flat parameter buffer -- enqueued without any
// host synchronisation; losses and accuracies of all batches of a call come back in one copy.
#include <cmath>

#include "acav_common.h"

using namespace acav;

namespace {

constexpr int CT_MAXB = 256;
constexpr float CT_TEMPERATURE = 0.1f;

// C[m][n] (+)= alpha * sum_k A(m, k) * B(n, k) (+ bias[n]); A(m, k) = A[m * sam + k * sak], B(n, k) = B[n * sbn + k * sbk]:
// one strided kernel serves X W^T (NT), G O (NN: sbn = 1, sbk = ldb) and dZ^T X (TN: sam = 1, sak = lda).
// 64 x 64 tile per workgroup, 16-deep LDS stages, 4 x 4 outputs per thread, k ascending (one fma chain per output).
__global__ __launch_bounds__(256) void k_ct_gemm(const float *__restrict__ A, long sam, long sak, const float *__restrict__ B, long sbn,
                                                 long sbk, float *__restrict__ C, long ldc, int M, int N, int K, float alpha,
                                                 const float *__restrict__ bias, int accumulate)
{
    __shared__ float sA[16][64 + 4], sB[16][64 + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4] = {};
    // split-K (gridDim.z > 1): slice z of the k range, raw sums into the z-th M x N plane of C (= the partial buffer); the
    // caller folds the planes in ascending z (k_ct_splitk_reduce): deterministic, unlike atomic adds
    const int kslice = gridDim.z > 1 ? ((K + (int)gridDim.z - 1) / (int)gridDim.z + 15) & ~15 : K;
    const int kbeg = (int)blockIdx.z * kslice, kend = min(K, kbeg + kslice);
    if (gridDim.z > 1) C += (size_t)blockIdx.z * (size_t)M * (size_t)ldc;
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        for (int e = tid; e < 64 * 16; e += 256) {
            const int r = e >> 4, kk = e & 15;  // consecutive threads walk k: contiguous for the NT operands
            const int m = m0 + r, n = n0 + r, k = k0 + kk;
            sA[kk][r] = (m < M && k < kend) ? A[m * sam + k * sak] : 0.f;
            sB[kk][r] = (n < N && k < kend) ? B[n * sbn + k * sbk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sA[kk][ty * 4 + i], b[i] = sB[kk][tx * 4 + i];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
            if (m < M && n < N) {
                float v = acc[i][j] * alpha;
                if (bias) v = v + bias[n];
                C[m * ldc + n] = accumulate ? C[m * ldc + n] + v : v;
            }
        }
}

// C[m][n] (+)= alpha * (P_0 + P_1 + ... + P_{S-1})[m][n] (+ bias[n]): the planes of a split-K product, folded in order
__global__ __launch_bounds__(256) void k_ct_splitk_reduce(const float *__restrict__ P, int S, float *__restrict__ C, long ldc, int M, int N,
                                                          float alpha, const float *__restrict__ bias, int accumulate)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)M * N) return;
    const int m = (int)(i / N), n = (int)(i % N);
    float v = P[i];
    for (int z = 1; z < S; ++z) v = v + P[(size_t)z * M * N + i];
    v = v * alpha;
    if (bias) v = v + bias[n];
    C[m * ldc + n] = accumulate ? C[m * ldc + n] + v : v;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) v = v + __shfl_xor(v, dlt);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) v = fmaxf(v, __shfl_xor(v, dlt));
    return v;
}

// F.normalize(z, dim=-1): o = z / max(||z||, 1e-12); one wave per row, both projections in one launch (blockIdx.y)
__global__ __launch_bounds__(256) void k_ct_normalize(const float *__restrict__ z1, const float *__restrict__ z2, float *__restrict__ o1,
                                                      float *__restrict__ o2, float *__restrict__ den1, float *__restrict__ den2,
                                                      int rows, int out)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float *z = (blockIdx.y ? z2 : z1) + (size_t)row * out;
    float *o = (blockIdx.y ? o2 : o1) + (size_t)row * out;
    float ss = 0.f;
    for (int j = lane; j < out; j += 64) ss = __builtin_fmaf(z[j], z[j], ss);
    ss = wave_sum(ss);
    const float den = fmaxf(__builtin_sqrtf(ss), 1e-12f);
    for (int j = lane; j < out; j += 64) o[j] = z[j] / den;
    if (lane == 0) (blockIdx.y ? den2 : den1)[row] = den;
}

// Row statistics of the logits L [B, B] in both directions: block i < B = row i (cross entropy of logits_ab), block B + j =
// column j (logits_ba = L^T).  stat = (max, sum exp), nll = logsumexp - L[i][i], hit = argmax is the diagonal (first
// maximum wins, like topk(1)).
__global__ __launch_bounds__(256) void k_ct_stats(const float *__restrict__ L, int B, float *__restrict__ mx, float *__restrict__ se,
                                                  float *__restrict__ nll, int *__restrict__ hit)
{
    __shared__ float sv[4];
    __shared__ int si[4];
    const int id = blockIdx.x, dir = id >= B, i = dir ? id - B : id;
    const long s0 = dir ? 1 : B, s1 = dir ? B : 1;  // element (i, t) of the direction
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float m = -INFINITY;
    int am = 0x7fffffff;
    for (int t = tid; t < B; t += 256) {
        const float v = L[i * s0 + t * s1];
        if (v > m) m = v, am = t;
    }
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
        const float om = __shfl_xor(m, dlt);
        const int oa = __shfl_xor(am, dlt);
        if (om > m || (om == m && oa < am)) m = om, am = oa;
    }
    if (lane == 0) sv[wv] = m, si[wv] = am;
    __syncthreads();
    m = sv[0], am = si[0];
    for (int q = 1; q < 4; ++q)
        if (sv[q] > m || (sv[q] == m && si[q] < am)) m = sv[q], am = si[q];
    __syncthreads();
    float s = 0.f;
    for (int t = tid; t < B; t += 256) s = s + expf(L[i * s0 + t * s1] - m);
    s = wave_sum(s);
    if (lane == 0) sv[wv] = s;
    __syncthreads();
    if (tid == 0) {
        const float tot = (sv[0] + sv[1]) + (sv[2] + sv[3]);
        mx[id] = m;
        se[id] = tot;
        nll[id] = (m + logf(tot)) - L[i * (long)B + i];
        hit[id] = am == i ? 1 : 0;
    }
}

// G = d loss / d L = ((softmax_rows - I) + (softmax_cols - I)) / (2 B); block 0 also folds the batch's loss / accuracy
__global__ __launch_bounds__(256) void k_ct_grad_logits(const float *__restrict__ L, int B, const float *__restrict__ mx,
                                                        const float *__restrict__ se, const float *__restrict__ nll,
                                                        const int *__restrict__ hit, float *__restrict__ G,
                                                        float *__restrict__ loss_out, float *__restrict__ acc_out)
{
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < B * B) {
        const int i = e / B, j = e - i * B;
        const float v = L[e];
        const float pa = expf(v - mx[i]) / se[i], pb = expf(v - mx[B + j]) / se[B + j];
        const float d = i == j ? 1.0f : 0.0f;
        G[e] = ((pa - d) + (pb - d)) / (float)(2 * B);
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {  // deterministic order: lane-strided partial sums, butterfly
        float s = 0.f;
        int h = 0;
        for (int t = threadIdx.x; t < 2 * B; t += 64) s = s + nll[t], h += hit[t];
        s = wave_sum(s);
        for (int dlt = 1; dlt < 64; dlt <<= 1) h += __shfl_xor(h, dlt);
        if (threadIdx.x == 0) {
            *loss_out = s / (float)(2 * B);
            *acc_out = (float)h / (float)(2 * B) * 100.0f;
        }
    }
}

// backward of F.normalize: dz = (do - o (o . do)) / den
__global__ __launch_bounds__(256) void k_ct_normalize_bwd(const float *__restrict__ o1, const float *__restrict__ o2,
                                                          const float *__restrict__ do1, const float *__restrict__ do2,
                                                          const float *__restrict__ den1, const float *__restrict__ den2,
                                                          float *__restrict__ dz1, float *__restrict__ dz2, int rows, int out)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const size_t off = (size_t)row * out;
    const float *o = (blockIdx.y ? o2 : o1) + off, *dd = (blockIdx.y ? do2 : do1) + off;
    float *dz = (blockIdx.y ? dz2 : dz1) + off;
    float dot = 0.f;
    for (int j = lane; j < out; j += 64) dot = __builtin_fmaf(o[j], dd[j], dot);
    dot = wave_sum(dot);
    const float den = (blockIdx.y ? den2 : den1)[row];
    for (int j = lane; j < out; j += 64) dz[j] = (dd[j] - o[j] * dot) / den;
}

// bias gradients: g[j] += sum_b dz[b][j] (b ascending)
__global__ __launch_bounds__(256) void k_ct_bias_grad(const float *__restrict__ dz1, const float *__restrict__ dz2, float *__restrict__ g1,
                                                      float *__restrict__ g2, int rows, int out)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= out) return;
    const float *dz = blockIdx.y ? dz2 : dz1;
    float s = 0.f;
    for (int b0 = 0; b0 < rows; b0 += 16) {  // sixteen loads in flight, summed in row order (a load per add was 28 us per batch)
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = b0 + q < rows ? dz[(size_t)(b0 + q) * out + j] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (b0 + q < rows) s = s + v[q];
    }
    float *g = blockIdx.y ? g2 : g1;
    g[j] = g[j] + s;
}

// torch.optim.AdamW, amsgrad=True, on the flat parameter buffer: decoupled decay, moments of the ACCUMULATED gradient,
// running max of the second moment, p -= lr / bc1 * m / (sqrt(vmax) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void k_ct_adamw(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                  float *__restrict__ v, float *__restrict__ vmax, long n, float decay,
                                                  float b1, float b2, float sqrt_bc2, float eps, float step_size)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    float pi = p[i] * decay;
    const float mi = m[i] * b1 + gi * (1.0f - b1);
    const float vi = v[i] * b2 + (gi * gi) * (1.0f - b2);
    const float vm = fmaxf(vmax[i], vi);
    m[i] = mi, v[i] = vi, vmax[i] = vm;
    const float denom = __builtin_sqrtf(vm) / sqrt_bc2 + eps;
    p[i] = pi - step_size * (mi / denom);
}

__global__ __launch_bounds__(256) void k_ct_scale(float *__restrict__ g, long n, float f)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) g[i] = g[i] * f;
}

__global__ __launch_bounds__(256) void k_ct_pair_dot(const float *__restrict__ o1, const float *__restrict__ o2, float *__restrict__ out_logits,
                                                     int rows, int out)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int j = lane; j < out; j += 64) s = __builtin_fmaf(o1[(size_t)row * out + j], o2[(size_t)row * out + j], s);
    s = wave_sum(s);
    if (lane == 0) out_logits[row] = s;
}

}  // namespace

struct acav_contrastive {
    StreamCtx ctx;
    int vis = 0, aud = 0, out = 0;
    int64_t step = 0;
    size_t nparam = 0;                      // vis*out + out + aud*out + out, in state_dict order
    DevBuf params, grads, m, v, vmax;       // flat [Wv | bv | Wa | ba]
    DevBuf stage_v, stage_a, z1, z2, o1, o2, den1, den2, L, mx, se, nll, hit, G, do1, do2, dz1, dz2, loss, acc, logits, splitk;
    acav_comm *comm = nullptr;              // set: the gradients are averaged over its ranks before every optimizer step
    int world = 1;
    float *Wv() { return params.as<float>(); }
    float *bv() { return Wv() + (size_t)vis * out; }
    float *Wa() { return bv() + out; }
    float *ba() { return Wa() + (size_t)aud * out; }
    float *g(float *p) { return grads.as<float>() + (p - params.as<float>()); }
};

static void ct_gemm(hipStream_t st, const float *A, long sam, long sak, const float *B, long sbn, long sbk, float *C, long ldc, int M,
                    int N, int K, float alpha, const float *bias, int accumulate, float *splitk = nullptr, size_t splitk_floats = 0)
{
    // a product of a few 64 x 64 tiles with a long k (the 128 x 128 x 2304 visual projection of a training batch: four
    // workgroups walking 144 stages each on a 256-CU chip) is cut along k into >= 128 workgroups and folded afterwards
    const int tiles = ((N + 63) / 64) * ((M + 63) / 64);
    int S = 1;
    if (splitk && tiles < 64 && K >= 256) {
        S = (128 + tiles - 1) / tiles;
        if (S > K / 64) S = K / 64;
        while (S > 1 && (size_t)S * M * N > splitk_floats) --S;
    }
    if (S > 1) {
        hipLaunchKernelGGL(k_ct_gemm, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)S), dim3(256), 0, st, A, sam,
                           sak, B, sbn, sbk, splitk, (long)N, M, N, K, 1.0f, nullptr, 0);
        hipLaunchKernelGGL(k_ct_splitk_reduce, dim3((unsigned)(((long)M * N + 255) / 256)), dim3(256), 0, st, splitk, S, C, ldc, M, N,
                           alpha, bias, accumulate);
        return;
    }
    hipLaunchKernelGGL(k_ct_gemm, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64)), dim3(256), 0, st, A, sam, sak, B, sbn,
                       sbk, C, ldc, M, N, K, alpha, bias, accumulate);
}

ACAV_EXPORT int acav_contrastive_destroy(acav_contrastive *c)
{
    if (!c) return ACAV_OK;
    (void)hipSetDevice(c->ctx.device);
    if (c->ctx.stream) (void)hipStreamSynchronize(c->ctx.stream);
    c->ctx.fini();
    delete c;
    return ACAV_OK;
}

// ContrastiveModule(visual_size, audio_size, out_size) with the given initial parameters (state_dict order:
// visual_linear.weight [out, vis], visual_linear.bias [out], audio_linear.weight [out, aud], audio_linear.bias [out])
ACAV_EXPORT int acav_contrastive_create(acav_contrastive **out_h, int device, int visual_size, int audio_size, int out_size,
                                        const float *params_host, void *stream)
{
    ACAV_REQUIRE(out_h && params_host, ACAV_EINVAL, "NULL argument");
    ACAV_REQUIRE(visual_size > 0 && audio_size > 0 && out_size > 0, ACAV_EINVAL, "bad sizes");
    acav_contrastive *c = new (std::nothrow) acav_contrastive;
    ACAV_REQUIRE(c, ACAV_ENOMEM, "out of host memory");
    int rc = c->ctx.init(device, stream);
    if (rc != ACAV_OK) {
        delete c;
        return rc;
    }
    c->vis = visual_size, c->aud = audio_size, c->out = out_size;
    c->nparam = (size_t)visual_size * out_size + out_size + (size_t)audio_size * out_size + out_size;
    const size_t bytes = sizeof(float) * c->nparam;
    auto body = [&]() -> int {
        for (DevBuf *b : {&c->params, &c->grads, &c->m, &c->v, &c->vmax}) ACAV_TRY(b->ensure(bytes));
        ACAV_HIP_TRY(hipMemcpyAsync(c->params.p, params_host, bytes, hipMemcpyHostToDevice, c->ctx.stream));
        for (DevBuf *b : {&c->grads, &c->m, &c->v, &c->vmax}) ACAV_HIP_TRY(hipMemsetAsync(b->p, 0, bytes, c->ctx.stream));
        ACAV_TRY(c->loss.ensure(sizeof(float)));
        ACAV_TRY(c->acc.ensure(sizeof(float)));
        ACAV_HIP_TRY(hipStreamSynchronize(c->ctx.stream));
        return ACAV_OK;
    };
    rc = body();
    if (rc != ACAV_OK) {
        acav_contrastive_destroy(c);
        return rc;
    }
    *out_h = c;
    return ACAV_OK;
}

// state_dict()['*'] / load_state_dict, plus the optimizer's position (`step`, gradients and moments travel too: the
// reference keeps one optimizer and never-zeroed .grad for the whole training)
ACAV_EXPORT int acav_contrastive_get_params(acav_contrastive *c, float *params_host, int64_t *step)
{
    ACAV_REQUIRE(c, ACAV_EINVAL, "handle is NULL");
    ACAV_HIP_TRY(hipSetDevice(c->ctx.device));
    if (params_host) {
        ACAV_HIP_TRY(hipMemcpyAsync(params_host, c->params.p, sizeof(float) * c->nparam, hipMemcpyDeviceToHost, c->ctx.stream));
        ACAV_HIP_TRY(hipStreamSynchronize(c->ctx.stream));
    }
    if (step) *step = c->step;
    return ACAV_OK;
}
ACAV_EXPORT int acav_contrastive_set_params(acav_contrastive *c, const float *params_host)
{
    ACAV_REQUIRE(c && params_host, ACAV_EINVAL, "NULL argument");
    ACAV_HIP_TRY(hipSetDevice(c->ctx.device));
    ACAV_HIP_TRY(hipMemcpyAsync(c->params.p, params_host, sizeof(float) * c->nparam, hipMemcpyHostToDevice, c->ctx.stream));
    ACAV_HIP_TRY(hipStreamSynchronize(c->ctx.stream));
    return ACAV_OK;
}

static int ct_forward(acav_contrastive *c, const float *dv, const float *da, int rows)
{
    hipStream_t st = c->ctx.stream;
    const size_t zo = sizeof(float) * (size_t)rows * c->out;
    for (DevBuf *b : {&c->z1, &c->z2, &c->o1, &c->o2}) ACAV_TRY(b->ensure(zo));
    ACAV_TRY(c->den1.ensure(sizeof(float) * rows));
    ACAV_TRY(c->den2.ensure(sizeof(float) * rows));
    const size_t skf = (size_t)64 * CT_MAXB * c->out;  // up to 64 planes of a training batch's projection
    if (rows <= CT_MAXB) ACAV_TRY(c->splitk.ensure(sizeof(float) * skf));
    float *sk = rows <= CT_MAXB ? c->splitk.as<float>() : nullptr;
    ct_gemm(st, dv, c->vis, 1, c->Wv(), c->vis, 1, c->z1.as<float>(), c->out, rows, c->out, c->vis, 1.0f, c->bv(), 0, sk, skf);
    ct_gemm(st, da, c->aud, 1, c->Wa(), c->aud, 1, c->z2.as<float>(), c->out, rows, c->out, c->aud, 1.0f, c->ba(), 0, sk, skf);
    hipLaunchKernelGGL(k_ct_normalize, dim3((unsigned)((rows + 3) / 4), 2), dim3(256), 0, st, c->z1.as<float>(), c->z2.as<float>(),
                       c->o1.as<float>(), c->o2.as<float>(), c->den1.as<float>(), c->den2.as<float>(), rows, c->out);
    ACAV_HIP_TRY(hipGetLastError());
    return ACAV_OK;
}

static int ct_ensure_train_buffers(acav_contrastive *c, int64_t nbatches)
{
    ACAV_TRY(c->loss.ensure(sizeof(float) * (size_t)(nbatches + 1)));
    ACAV_TRY(c->acc.ensure(sizeof(float) * (size_t)(nbatches + 1)));
    const size_t bb = sizeof(float) * CT_MAXB * CT_MAXB, bo = sizeof(float) * CT_MAXB * (size_t)c->out;
    ACAV_TRY(c->L.ensure(bb));
    ACAV_TRY(c->G.ensure(bb));
    for (DevBuf *b : {&c->mx, &c->se, &c->nll}) ACAV_TRY(b->ensure(sizeof(float) * 2 * CT_MAXB));
    ACAV_TRY(c->hit.ensure(sizeof(int) * 2 * CT_MAXB));
    for (DevBuf *b : {&c->do1, &c->do2, &c->dz1, &c->dz2}) ACAV_TRY(b->ensure(bo));
    return ACAV_OK;
}

// forward + backward of one batch of B clips (device rows): the gradients ACCUMULATE into .grad, loss / acc land in slot i
static int ct_backward(acav_contrastive *c, const float *xv, const float *xa, int B, int64_t i)
{
    hipStream_t st = c->ctx.stream;
    const float inv_t = 1.0f / CT_TEMPERATURE;
    ACAV_TRY(ct_forward(c, xv, xa, B));
    float *o1 = c->o1.as<float>(), *o2 = c->o2.as<float>(), *L = c->L.as<float>(), *G = c->G.as<float>();
    ct_gemm(st, o1, c->out, 1, o2, c->out, 1, L, B, B, B, c->out, inv_t, nullptr, 0);  // logits_ab = o1 o2^T / T
    hipLaunchKernelGGL(k_ct_stats, dim3((unsigned)(2 * B)), dim3(256), 0, st, L, B, c->mx.as<float>(), c->se.as<float>(),
                       c->nll.as<float>(), c->hit.as<int>());
    hipLaunchKernelGGL(k_ct_grad_logits, dim3((unsigned)((B * B + 255) / 256)), dim3(256), 0, st, L, B, c->mx.as<float>(),
                       c->se.as<float>(), c->nll.as<float>(), c->hit.as<int>(), G, c->loss.as<float>() + i, c->acc.as<float>() + i);
    // d o1 = G o2 / T (NN), d o2 = G^T o1 / T (TN)
    ct_gemm(st, G, B, 1, o2, 1, c->out, c->do1.as<float>(), c->out, B, c->out, B, inv_t, nullptr, 0);
    ct_gemm(st, G, 1, B, o1, 1, c->out, c->do2.as<float>(), c->out, B, c->out, B, inv_t, nullptr, 0);
    hipLaunchKernelGGL(k_ct_normalize_bwd, dim3((unsigned)((B + 3) / 4), 2), dim3(256), 0, st, o1, o2, c->do1.as<float>(),
                       c->do2.as<float>(), c->den1.as<float>(), c->den2.as<float>(), c->dz1.as<float>(), c->dz2.as<float>(), B,
                       c->out);
    // weight gradients (TN, accumulated into .grad): dW [out, in] += dz^T x
    ct_gemm(st, c->dz1.as<float>(), 1, c->out, xv, 1, c->vis, c->g(c->Wv()), c->vis, c->out, c->vis, B, 1.0f, nullptr, 1);
    ct_gemm(st, c->dz2.as<float>(), 1, c->out, xa, 1, c->aud, c->g(c->Wa()), c->aud, c->out, c->aud, B, 1.0f, nullptr, 1);
    hipLaunchKernelGGL(k_ct_bias_grad, dim3((unsigned)((c->out + 255) / 256), 2), dim3(256), 0, st, c->dz1.as<float>(),
                       c->dz2.as<float>(), c->g(c->bv()), c->g(c->ba()), B, c->out);
    ACAV_HIP_TRY(hipGetLastError());
    return ACAV_OK;
}

// ContrastiveModule.average_gradient (module.py:96-101): all-reduce SUM of every .grad, then / world -- on the flat buffer
static int ct_average_grads(acav_contrastive *c)
{
    if (!c->comm || c->world <= 1) return ACAV_OK;
    ACAV_HIP_TRY(hipStreamSynchronize(c->ctx.stream));  // the communicator has its own stream
    ACAV_TRY(acav_comm_allreduce_f32(c->comm, c->grads.as<float>(), (int64_t)c->nparam));
    ACAV_TRY(acav_comm_sync(c->comm));
    hipLaunchKernelGGL(k_ct_scale, dim3((unsigned)((c->nparam + 255) / 256)), dim3(256), 0, c->ctx.stream, c->grads.as<float>(),
                       (long)c->nparam, 1.0f / (float)c->world);
    ACAV_HIP_TRY(hipGetLastError());
    return ACAV_OK;
}

static int ct_step(acav_contrastive *c, double lr)
{
    c->step += 1;
    const double b1 = 0.9, b2 = 0.999, bc1 = 1.0 - pow(b1, (double)c->step), bc2 = 1.0 - pow(b2, (double)c->step);
    hipLaunchKernelGGL(k_ct_adamw, dim3((unsigned)((c->nparam + 255) / 256)), dim3(256), 0, c->ctx.stream, c->params.as<float>(),
                       c->grads.as<float>(), c->m.as<float>(), c->v.as<float>(), c->vmax.as<float>(), (long)c->nparam,
                       (float)(1.0 - lr * 0.01), (float)b1, (float)b2, (float)sqrt(bc2), 1e-6f, (float)(lr / bc1));
    ACAV_HIP_TRY(hipGetLastError());
    return ACAV_OK;
}

// Contrastive.train's inner loop over `nbatches` batches (contrastive.py:92-101,117-132): batch i = rows
// [offsets[i], offsets[i+1]) of visual / audio; forward, backward (gradients accumulate), [with a communicator: average of
// the gradients over the ranks,] AdamW step with this epoch's lr.
// losses / accs [nbatches] = the reference's loss.item() / acc.item().  Host or device inputs.
ACAV_EXPORT int acav_contrastive_train(acav_contrastive *c, const float *visual, const float *audio, const int64_t *offsets,
                                       int64_t nbatches, double lr, float *losses, float *accs)
{
    ACAV_REQUIRE(c && visual && audio && offsets && nbatches >= 0, ACAV_EINVAL, "NULL argument");
    ACAV_HIP_TRY(hipSetDevice(c->ctx.device));
    hipStream_t st = c->ctx.stream;
    const int64_t n = offsets[nbatches];
    for (int64_t i = 0; i < nbatches; ++i)
        ACAV_REQUIRE(offsets[i + 1] > offsets[i] && offsets[i + 1] - offsets[i] <= CT_MAXB, ACAV_EINVAL,
                     "batch %lld has %lld rows (1..%d supported)", (long long)i, (long long)(offsets[i + 1] - offsets[i]), CT_MAXB);
    const void *dv = nullptr, *da = nullptr;
    ACAV_TRY(to_device(visual, sizeof(float) * (size_t)n * c->vis, c->stage_v, st, &dv));
    ACAV_TRY(to_device(audio, sizeof(float) * (size_t)n * c->aud, c->stage_a, st, &da));
    ACAV_TRY(ct_ensure_train_buffers(c, nbatches));
    for (int64_t i = 0; i < nbatches; ++i) {
        const int B = (int)(offsets[i + 1] - offsets[i]);
        ACAV_TRY(ct_backward(c, static_cast<const float *>(dv) + (size_t)offsets[i] * c->vis,
                             static_cast<const float *>(da) + (size_t)offsets[i] * c->aud, B, i));
        ACAV_TRY(ct_average_grads(c));
        ACAV_TRY(ct_step(c, lr));
    }
    if (nbatches) {
        if (losses) ACAV_HIP_TRY(hipMemcpyAsync(losses, c->loss.p, sizeof(float) * (size_t)nbatches, hipMemcpyDeviceToHost, st));
        if (accs) ACAV_HIP_TRY(hipMemcpyAsync(accs, c->acc.p, sizeof(float) * (size_t)nbatches, hipMemcpyDeviceToHost, st));
    }
    ACAV_HIP_TRY(hipStreamSynchronize(st));
    return ACAV_OK;
}

// The distributed form of the loop (run_contrastive.py:118-168, contrastive.py:92-101 with distributed=True), in pieces.
// acav_contrastive_set_comm: with an RCCL communicator of more than one rank, acav_contrastive_train averages the gradients
// itself.  Without RCCL (the ranks share a GPU, or a host-side process group is all there is) the caller runs
// backward -> get_grads -> its own all-reduce / world -> set_grads -> step per batch.
ACAV_EXPORT int acav_contrastive_set_comm(acav_contrastive *c, acav_comm *comm)
{
    ACAV_REQUIRE(c, ACAV_EINVAL, "handle is NULL");
    c->comm = comm;
    c->world = 1;
    if (comm) {
        int rank = 0;
        ACAV_TRY(acav_comm_info(comm, &rank, &c->world));
    }
    return ACAV_OK;
}
ACAV_EXPORT int acav_contrastive_backward(acav_contrastive *c, const float *visual, const float *audio, int64_t n, float *loss,
                                          float *acc)
{
    ACAV_REQUIRE(c && visual && audio && n >= 1 && n <= CT_MAXB, ACAV_EINVAL, "bad argument (1..%d rows)", CT_MAXB);
    ACAV_HIP_TRY(hipSetDevice(c->ctx.device));
    hipStream_t st = c->ctx.stream;
    const void *dv = nullptr, *da = nullptr;
    ACAV_TRY(to_device(visual, sizeof(float) * (size_t)n * c->vis, c->stage_v, st, &dv));
    ACAV_TRY(to_device(audio, sizeof(float) * (size_t)n * c->aud, c->stage_a, st, &da));
    ACAV_TRY(ct_ensure_train_buffers(c, 1));
    ACAV_TRY(ct_backward(c, static_cast<const float *>(dv), static_cast<const float *>(da), (int)n, 0));
    if (loss) ACAV_HIP_TRY(hipMemcpyAsync(loss, c->loss.p, sizeof(float), hipMemcpyDeviceToHost, st));
    if (acc) ACAV_HIP_TRY(hipMemcpyAsync(acc, c->acc.p, sizeof(float), hipMemcpyDeviceToHost, st));
    ACAV_HIP_TRY(hipStreamSynchronize(st));
    return ACAV_OK;
}
ACAV_EXPORT int acav_contrastive_get_grads(acav_contrastive *c, float *grads_host)
{
    ACAV_REQUIRE(c && grads_host, ACAV_EINVAL, "NULL argument");
    ACAV_HIP_TRY(hipSetDevice(c->ctx.device));
    ACAV_HIP_TRY(hipMemcpyAsync(grads_host, c->grads.p, sizeof(float) * c->nparam, hipMemcpyDeviceToHost, c->ctx.stream));
    ACAV_HIP_TRY(hipStreamSynchronize(c->ctx.stream));
    return ACAV_OK;
}
ACAV_EXPORT int acav_contrastive_set_grads(acav_contrastive *c, const float *grads_host)
{
    ACAV_REQUIRE(c && grads_host, ACAV_EINVAL, "NULL argument");
    ACAV_HIP_TRY(hipSetDevice(c->ctx.device));
    ACAV_HIP_TRY(hipMemcpyAsync(c->grads.p, grads_host, sizeof(float) * c->nparam, hipMemcpyHostToDevice, c->ctx.stream));
    ACAV_HIP_TRY(hipStreamSynchronize(c->ctx.stream));
    return ACAV_OK;
}
ACAV_EXPORT int acav_contrastive_step(acav_contrastive *c, double lr)
{
    ACAV_REQUIRE(c, ACAV_EINVAL, "handle is NULL");
    ACAV_HIP_TRY(hipSetDevice(c->ctx.device));
    ACAV_TRY(ct_average_grads(c));
    ACAV_TRY(ct_step(c, lr));
    ACAV_HIP_TRY(hipStreamSynchronize(c->ctx.stream));
    return ACAV_OK;
}

// ContrastiveModule.infer (module.py:87-92): logits[i] = <normalize(visual_i W_v^T + b_v), normalize(audio_i W_a^T + b_a)>
ACAV_EXPORT int acav_contrastive_infer(acav_contrastive *c, const float *visual, const float *audio, int64_t n, float *logits_host)
{
    ACAV_REQUIRE(c && (n == 0 || (visual && audio && logits_host)) && n >= 0, ACAV_EINVAL, "bad argument");
    if (n == 0) return ACAV_OK;
    ACAV_HIP_TRY(hipSetDevice(c->ctx.device));
    hipStream_t st = c->ctx.stream;
    const void *dv = nullptr, *da = nullptr;
    ACAV_TRY(to_device(visual, sizeof(float) * (size_t)n * c->vis, c->stage_v, st, &dv));
    ACAV_TRY(to_device(audio, sizeof(float) * (size_t)n * c->aud, c->stage_a, st, &da));
    ACAV_TRY(c->logits.ensure(sizeof(float) * (size_t)n));
    const int64_t chunk = 8192;
    for (int64_t r0 = 0; r0 < n; r0 += chunk) {
        const int rows = (int)(n - r0 < chunk ? n - r0 : chunk);
        ACAV_TRY(ct_forward(c, static_cast<const float *>(dv) + (size_t)r0 * c->vis, static_cast<const float *>(da) + (size_t)r0 * c->aud,
                            rows));
        hipLaunchKernelGGL(k_ct_pair_dot, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, c->o1.as<float>(), c->o2.as<float>(),
                           c->logits.as<float>() + r0, rows, c->out);
        ACAV_HIP_TRY(hipGetLastError());
    }
    ACAV_HIP_TRY(hipMemcpyAsync(logits_host, c->logits.p, sizeof(float) * (size_t)n,
                                is_device_ptr(logits_host) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, st));
    ACAV_HIP_TRY(hipStreamSynchronize(st));
    return ACAV_OK;
}
