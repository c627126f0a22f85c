// acav_comm.hip -- the collectives of the path through RCCL, behind the C ABI (SURVEY 8(b) minimum export set:
// comm_init(rank, world, unique_id), allreduce_init(comm)).  Replaces what the reference does through torch.distributed
// in KMeans.initialize / KMeans.add (clustering/code/sgd_clustering.py:88-92, 97, 115, 126; mps/distributed.py:139-155).
//
// What the path really exchanges between GPUs (one process per GPU, xGMI):
//   initialize()      one all-reduce of [centres | counts]                       (K d + K floats, once)
//   multi-GPU training  the ROWS of the steps ahead, in bulk, to the one rank that runs a clustering's SGD chain -- which rows
//                     form a step's global batch is a plan (acav_kmeans_train_plan_multi: the reference's own N-GPU batch
//                     stream, the one-GPU stream over partitioned rows, or the large-batch layout of acav_kmeans_train_dp:
//                     rank-major concatenation of every rank's rows [t b, (t+1) b)); the chain itself is device-resident
//                     -- no collective on the 4-7 us step path
//   view-parallel     broadcast of a clustering's state from the rank that trained it (K d + K floats per epoch)
// RCCL is resolved at run time (dlopen of the librccl the process already has -- torch's -- else the ROCm one): the
// library loads and every other entry point works on a box without RCCL; only acav_comm_* report ACAV_ESTATE there.
#include <dlfcn.h>
#include <vector>
#include <rccl/rccl.h>

#include "acav_common.h"
#include <chrono>
#include <thread>

using namespace acav;

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    bool p2p = false;  // Send / Recv / Group* resolved: rows can be gathered to the one rank that consumes them
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *nm : names) {  // the copy already in the process first (torch bundles its own)
        r.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
        if (r.lib) break;
    }
    for (const char *nm : names) {
        if (r.lib) break;
        r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    }
    if (!r.lib) return r;
#define ACAV_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name))
    ACAV_SYM(GetUniqueId, "ncclGetUniqueId");
    ACAV_SYM(CommInitRank, "ncclCommInitRank");
    ACAV_SYM(CommDestroy, "ncclCommDestroy");
    ACAV_SYM(AllReduce, "ncclAllReduce");
    ACAV_SYM(AllGather, "ncclAllGather");
    ACAV_SYM(Broadcast, "ncclBroadcast");
    ACAV_SYM(GetErrorString, "ncclGetErrorString");
    ACAV_SYM(Send, "ncclSend");
    ACAV_SYM(Recv, "ncclRecv");
    ACAV_SYM(GroupStart, "ncclGroupStart");
    ACAV_SYM(GroupEnd, "ncclGroupEnd");
#undef ACAV_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.AllGather && r.Broadcast;
    r.p2p = r.Send && r.Recv && r.GroupStart && r.GroupEnd;
    return r;
}

#define ACAV_NCCL_TRY(expr)                                                                                  \
    do {                                                                                                     \
        ncclResult_t _r = (expr);                                                                            \
        if (_r != ncclSuccess) {                                                                             \
            acav::set_error("%s failed: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(_r) : "?"); \
            return ACAV_EHIP;                                                                                \
        }                                                                                                    \
    } while (0)

// [rank][step][row][:] -> [step][rank][row][:]: the all-gathered rows of `steps` steps become the global batches
__global__ __launch_bounds__(256) void k_interleave_rows(const float4 *__restrict__ in, float4 *__restrict__ out, int world,
                                                         int steps, int bl, int d4)
{
    const int64_t total = (int64_t)world * steps * bl * d4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % d4);
        int64_t r = i / d4;  // output row: (t * world + rank) * bl + row
        const int row = (int)(r % bl);
        r /= bl;
        const int rank = (int)(r % world), t = (int)(r / world);
        out[i] = in[(((int64_t)rank * steps + t) * bl + row) * d4 + c];
    }
}

// The rows of a chunk reach the rank(s) that consume them: with a known trainer (`root` >= 0) every rank SENDS its rows
// to it and only it receives (one grouped send / receive round: (W-1)/W of the all-gather's bytes arrive at ONE rank
// instead of at every rank -- at W = 8 an eighth of the xGMI traffic); root < 0: every rank trains, all-gather.
static int rows_to_trainers(const Rccl &R, const void *send, void *recv, size_t bytes, int root, int rank, int world,
                            ncclComm_t comm, hipStream_t st)
{
    if (root >= world) return ACAV_OK;  // the trainer is outside this communicator (single-rank tests): nothing to move
    if (root < 0 || !R.p2p) {
        ACAV_NCCL_TRY(R.AllGather(send, recv, bytes, ncclInt8, comm, st));
        return ACAV_OK;
    }
    ACAV_NCCL_TRY(R.GroupStart());
    ncclResult_t r1 = R.Send(send, bytes, ncclInt8, root, comm, st), r2 = ncclSuccess;
    if (rank == root)
        for (int r = 0; r < world && r2 == ncclSuccess; ++r)
            r2 = R.Recv(static_cast<char *>(recv) + (size_t)r * bytes, bytes, ncclInt8, r, comm, st);
    ncclResult_t r3 = R.GroupEnd();
    ACAV_NCCL_TRY(r1);
    ACAV_NCCL_TRY(r2);
    ACAV_NCCL_TRY(r3);
    return ACAV_OK;
}

// acav_kmeans_train_plan_multi: a destination row finds its source row through a sorted table of row ranges ("pieces").
//   PLACE = false  packing on the sending side: destination row R of the packed block <- local row src + (R - key)
//                  of the piece with the largest key <= R (one group)
//   PLACE = true   the trainer's global batches [step][slot][row]: destination row R = (t, q, j) reads position
//                  key = t * lb + j of slot q's stream within this chunk; group q holds that slot's pieces, whose `src`
//                  is a row of the landing buffer (the ranks' packed blocks one after the other)
struct PlanPiece {
    int64_t key, src, len;
};

template <bool PLACE>
__global__ __launch_bounds__(256) void k_rows_by_table(const PlanPiece *__restrict__ pieces, const int *__restrict__ group_begin,
                                                       const float4 *__restrict__ src, float4 *__restrict__ dst, int64_t rows,
                                                       int d4, int slots, int lb, int lanes_per_row)
{
    const int rpb = 256 / lanes_per_row, sub = (int)threadIdx.x / lanes_per_row, lane = (int)threadIdx.x % lanes_per_row;
    for (int64_t R = (int64_t)blockIdx.x * rpb + sub; R < rows; R += (int64_t)gridDim.x * rpb) {
        int g = 0;
        int64_t key = R;
        if (PLACE) {
            const int64_t bg = (int64_t)slots * lb, t = R / bg;
            const int rem = (int)(R - t * bg);
            g = rem / lb;
            key = t * lb + (rem - g * lb);
        }
        int lo = group_begin[g], hi = group_begin[g + 1] - 1;  // the last piece of the group whose key is <= `key`
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (pieces[mid].key <= key) lo = mid;
            else hi = mid - 1;
        }
        const PlanPiece p = pieces[lo];
        const float4 *s = src + (p.src + (key - p.key)) * d4;
        float4 *o = dst + R * d4;
        for (int c = lane; c < d4; c += lanes_per_row) o[c] = s[c];
    }
}

static void launch_rows_by_table(bool place, const PlanPiece *pieces, const int *group_begin, const float *src, float *dst,
                                 int64_t rows, int d, int slots, int lb, hipStream_t st)
{
    if (rows <= 0) return;
    const int d4 = d / 4;
    const int lpr = d4 >= 256 ? 256 : d4 >= 128 ? 128 : d4 >= 64 ? 64 : 32;
    const int rpb = 256 / lpr;
    int64_t blocks = (rows + rpb - 1) / rpb;
    if (blocks > 65536) blocks = 65536;
    if (place)
        hipLaunchKernelGGL(k_rows_by_table<true>, dim3((unsigned)blocks), dim3(256), 0, st, pieces, group_begin,
                           reinterpret_cast<const float4 *>(src), reinterpret_cast<float4 *>(dst), rows, d4, slots, lb, lpr);
    else
        hipLaunchKernelGGL(k_rows_by_table<false>, dim3((unsigned)blocks), dim3(256), 0, st, pieces, group_begin,
                           reinterpret_cast<const float4 *>(src), reinterpret_cast<float4 *>(dst), rows, d4, slots, lb, lpr);
}

// every rank's packed rows of a chunk -> the trainer `root`: one grouped send / receive round; a rank without rows in
// this chunk sends nothing and the root posts no receive for it (both sides read that off the same plan)
static int packed_rows_to_root(const Rccl &R, const void *send, int64_t my_rows, void *recv, const int64_t *from, size_t row_bytes,
                               int root, int rank, int world, ncclComm_t comm, hipStream_t st)
{
    if (root >= world) return ACAV_OK;  // the trainer is outside this communicator (single-rank tests): nothing to move
    ACAV_NCCL_TRY(R.GroupStart());
    ncclResult_t r1 = ncclSuccess, r2 = ncclSuccess;
    if (my_rows > 0) r1 = R.Send(send, (size_t)my_rows * row_bytes, ncclInt8, root, comm, st);
    if (rank == root) {
        size_t off = 0;
        for (int r = 0; r < world && r2 == ncclSuccess; ++r) {
            if (from[r] > 0) r2 = R.Recv(static_cast<char *>(recv) + off, (size_t)from[r] * row_bytes, ncclInt8, r, comm, st);
            off += (size_t)from[r] * row_bytes;
        }
    }
    ncclResult_t r3 = R.GroupEnd();
    ACAV_NCCL_TRY(r1);
    ACAV_NCCL_TRY(r2);
    ACAV_NCCL_TRY(r3);
    return ACAV_OK;
}

__global__ void k_scale_f32(float *__restrict__ v, int64_t n, float s)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = v[i] * s;
}

}  // namespace

struct acav_comm {
    StreamCtx ctx;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    DevBuf gather[2], batches[2], pack[2], flat;
    DevBuf plan_pieces, plan_groups;  // acav_kmeans_train_plan_multi: the epoch's piece tables (read-only once uploaded)
    hipEvent_t ev_gathered[2] = {nullptr, nullptr}, ev_trained[2] = {nullptr, nullptr};
};

ACAV_EXPORT int acav_comm_unique_id(uint8_t *id128)
{
    ACAV_REQUIRE(id128, ACAV_EINVAL, "NULL argument");
    ACAV_REQUIRE(rccl().ok, ACAV_ESTATE, "RCCL is not available in this process (librccl.so not found)");
    ncclUniqueId id;
    ACAV_NCCL_TRY(rccl().GetUniqueId(&id));
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return ACAV_OK;
}

ACAV_EXPORT int acav_comm_destroy(acav_comm *c)
{
    if (!c) return ACAV_OK;
    (void)hipSetDevice(c->ctx.device);
    if (c->ctx.stream) (void)hipStreamSynchronize(c->ctx.stream);
    if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
    for (int q = 0; q < 2; ++q) {
        if (c->ev_gathered[q]) (void)hipEventDestroy(c->ev_gathered[q]);
        if (c->ev_trained[q]) (void)hipEventDestroy(c->ev_trained[q]);
    }
    c->ctx.fini();
    delete c;
    return ACAV_OK;
}

ACAV_EXPORT int acav_comm_init(acav_comm **out, int device, int rank, int world, const uint8_t *id128, void *stream)
{
    ACAV_REQUIRE(out && id128, ACAV_EINVAL, "NULL argument");
    ACAV_REQUIRE(world >= 1 && rank >= 0 && rank < world, ACAV_EINVAL, "bad rank %d / world %d", rank, world);
    ACAV_REQUIRE(rccl().ok, ACAV_ESTATE, "RCCL is not available in this process (librccl.so not found)");
    acav_comm *c = new (std::nothrow) acav_comm;
    ACAV_REQUIRE(c, ACAV_ENOMEM, "out of host memory");
    int rc = c->ctx.init(device, stream);
    if (rc != ACAV_OK) {
        delete c;
        return rc;
    }
    c->rank = rank, c->world = world;
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);
    bool ok = r == ncclSuccess;
    if (!ok) set_error("ncclCommInitRank failed: %s", rccl().GetErrorString ? rccl().GetErrorString(r) : "?");
    for (int q = 0; q < 2 && ok; ++q) {
        ok = hipEventCreateWithFlags(&c->ev_gathered[q], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_trained[q], hipEventDisableTiming) == hipSuccess;
        if (!ok) set_error("could not create events");
    }
    if (!ok) {
        c->comm = r == ncclSuccess ? c->comm : nullptr;
        acav_comm_destroy(c);
        return ACAV_EHIP;
    }
    *out = c;
    return ACAV_OK;
}

ACAV_EXPORT int acav_comm_info(const acav_comm *c, int *rank, int *world)
{
    ACAV_REQUIRE(c, ACAV_EINVAL, "handle is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return ACAV_OK;
}

ACAV_EXPORT int acav_comm_sync(acav_comm *c)
{
    ACAV_REQUIRE(c, ACAV_EINVAL, "handle is NULL");
    ACAV_HIP_TRY(hipStreamSynchronize(c->ctx.stream));
    return ACAV_OK;
}

// in-place sum over the ranks of n floats (device memory); rank order of the summation is RCCL's
ACAV_EXPORT int acav_comm_allreduce_f32(acav_comm *c, float *buf_dev, int64_t n)
{
    ACAV_REQUIRE(c && buf_dev && n >= 0, ACAV_EINVAL, "bad argument");
    ACAV_REQUIRE(is_device_ptr(buf_dev), ACAV_EINVAL, "buffer must be device memory");
    ACAV_HIP_TRY(hipSetDevice(c->ctx.device));
    ACAV_NCCL_TRY(rccl().AllReduce(buf_dev, buf_dev, (size_t)n, ncclFloat32, ncclSum, c->comm, c->ctx.stream));
    return ACAV_OK;
}

// recv[rank r] = send of rank r (bytes each), device memory on both sides
ACAV_EXPORT int acav_comm_allgather(acav_comm *c, const void *send_dev, void *recv_dev, int64_t bytes)
{
    ACAV_REQUIRE(c && send_dev && recv_dev && bytes >= 0, ACAV_EINVAL, "bad argument");
    ACAV_REQUIRE(is_device_ptr(send_dev) && is_device_ptr(recv_dev), ACAV_EINVAL, "buffers must be device memory");
    ACAV_HIP_TRY(hipSetDevice(c->ctx.device));
    ACAV_NCCL_TRY(rccl().AllGather(send_dev, recv_dev, (size_t)bytes, ncclInt8, c->comm, c->ctx.stream));
    return ACAV_OK;
}

ACAV_EXPORT int acav_comm_broadcast(acav_comm *c, void *buf_dev, int64_t bytes, int root)
{
    ACAV_REQUIRE(c && buf_dev && bytes >= 0 && root >= 0 && root < c->world, ACAV_EINVAL, "bad argument");
    ACAV_REQUIRE(is_device_ptr(buf_dev), ACAV_EINVAL, "buffer must be device memory");
    ACAV_HIP_TRY(hipSetDevice(c->ctx.device));
    ACAV_NCCL_TRY(rccl().Broadcast(buf_dev, buf_dev, (size_t)bytes, ncclInt8, root, c->comm, c->ctx.stream));
    return ACAV_OK;
}

// KMeans.initialize() (sgd_clustering.py:88-92): all-reduce(SUM) of centres and counts, times 1/world
ACAV_EXPORT int acav_kmeans_allreduce_init(acav_kmeans *km, acav_comm *c)
{
    ACAV_REQUIRE(km && c, ACAV_EINVAL, "NULL argument");
    int K = 0, d = 0;
    ACAV_TRY(acav_kmeans_shape(km, &K, &d));
    const int64_t nc = (int64_t)K * d, n = nc + K;
    ACAV_HIP_TRY(hipSetDevice(c->ctx.device));
    ACAV_TRY(c->flat.ensure(sizeof(float) * (size_t)n));
    float *flat = c->flat.as<float>();
    int64_t count = 0, fb = 0;
    ACAV_TRY(acav_kmeans_get_state(km, flat, flat + nc, &count, &fb));  // device destination: device-to-device copies
    ACAV_TRY(acav_comm_allreduce_f32(c, flat, n));
    if (c->world > 1) {
        hipLaunchKernelGGL(k_scale_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->ctx.stream, flat, n,
                           1.0f / (float)c->world);
        ACAV_HIP_TRY(hipGetLastError());
    }
    ACAV_HIP_TRY(hipStreamSynchronize(c->ctx.stream));
    return acav_kmeans_set_state(km, flat, flat + nc, count, fb);
}

// One epoch of the reference's multi-GPU add() loop (sgd_clustering.py:94-129 under is_distributed): every rank feeds
// b_local rows per step, the global batch is the rank-major concatenation.  The rows of `chunk_steps` steps are
// all-gathered at a time on the communicator's stream (double-buffered: the gather of chunk c+1 overlaps the training
// of chunk c), re-ordered into global batches by k_interleave_rows, and trained by acav_kmeans_train on every rank --
// identical state everywhere, no collective on the step path.  warm_global [n_warm, world * b_local]: the labels of the
// warm-up steps, already in global-batch order (drawn per rank, exchanged once by the caller).
// flags: bit 0 (ACAV_DP_TRAIN) -- this rank trains; without it the rank only takes part in the row exchange (another
// rank trains this clustering and broadcasts its state afterwards, acav_kmeans_broadcast_state): with several
// clusterings over the same rows the replicated chains are dealt out over the ranks instead of every rank running all
// of them.  bit 1 (ACAV_DP_NOWAIT) -- return once everything is enqueued (collectives on the communicator's stream,
// training on the clustering's): with one communicator per clustering, different ranks then train different
// clusterings AT THE SAME TIME while all of them keep feeding every exchange.
ACAV_EXPORT int acav_kmeans_train_dp(acav_kmeans *km, acav_comm *c, const float *x_local_dev, int64_t n_local, int64_t b_local,
                                     double lr, const int64_t *warm_global, int64_t n_warm, int64_t chunk_steps, int flags)
{
    const bool train_here = (flags & 1) != 0, nowait = (flags & 2) != 0;
    const int root = (flags & 4) ? (flags >> 8) : -1;  // ACAV_DP_ROOTED: the one rank that trains (the same on every rank)
    ACAV_REQUIRE(root < 0 || (c && train_here == (c->rank == root)), ACAV_EINVAL,
                 "ACAV_DP_ROOTED: root %d does not match ACAV_DP_TRAIN on rank %d", root, c ? c->rank : -1);
    ACAV_REQUIRE(km && c && (x_local_dev || n_local == 0), ACAV_EINVAL, "NULL argument");
    ACAV_REQUIRE(n_local >= 0 && b_local > 0 && chunk_steps > 0 && n_warm >= 0, ACAV_EINVAL, "bad sizes");
    int K = 0, d = 0;
    ACAV_TRY(acav_kmeans_shape(km, &K, &d));
    ACAV_REQUIRE((d & 3) == 0, ACAV_EINVAL, "d = %d must be a multiple of 4 for the bulk exchange", d);
    const int64_t steps = n_local / b_local;
    if (steps == 0) return ACAV_OK;
    ACAV_REQUIRE(is_device_ptr(x_local_dev), ACAV_EINVAL, "x_local must be device memory");
    const int w = c->world;
    const int64_t bg = (int64_t)w * b_local;
    ACAV_HIP_TRY(hipSetDevice(c->ctx.device));
    hipStream_t sc = c->ctx.stream;
    void *st_train = nullptr;
    ACAV_TRY(acav_kmeans_stream(km, &st_train));
    const size_t chunk_bytes = sizeof(float) * (size_t)chunk_steps * (size_t)bg * (size_t)d;
    for (int q = 0; q < 2; ++q) {  // a rank that only sends needs no landing buffers
        if (train_here || root < 0 || !rccl().p2p) ACAV_TRY(c->gather[q].ensure(chunk_bytes));
        if (train_here) ACAV_TRY(c->batches[q].ensure(chunk_bytes));
    }
    int64_t warm_done = 0;
    auto gather = [&](int64_t c0, int par) -> int {  // rows of steps [c0, c0 + s) of every rank -> batches[par]
        const int64_t s = steps - c0 < chunk_steps ? steps - c0 : chunk_steps;
        const size_t bytes = sizeof(float) * (size_t)s * (size_t)b_local * (size_t)d;
        if (train_here) ACAV_HIP_TRY(hipStreamWaitEvent(sc, c->ev_trained[par], 0));  // the chunk trained from this buffer is done
        ACAV_TRY(rows_to_trainers(rccl(), x_local_dev + (size_t)c0 * b_local * d, c->gather[par].p, bytes, root, c->rank, w, c->comm, sc));
        if (train_here) {
            const int64_t total4 = (int64_t)w * s * b_local * (d / 4);
            const unsigned grid = (unsigned)(total4 / 256 + 1 < 4096 ? total4 / 256 + 1 : 4096);
            hipLaunchKernelGGL(k_interleave_rows, dim3(grid), dim3(256), 0, sc, c->gather[par].as<float4>(),
                               c->batches[par].as<float4>(), w, (int)s, (int)b_local, d / 4);
            ACAV_HIP_TRY(hipGetLastError());
        }
        ACAV_HIP_TRY(hipEventRecord(c->ev_gathered[par], sc));
        return ACAV_OK;
    };
    for (int q = 0; q < 2; ++q) ACAV_HIP_TRY(hipEventRecord(c->ev_trained[q], (hipStream_t)st_train));  // both buffers free
    ACAV_TRY(gather(0, 0));
    int par = 0;
    for (int64_t c0 = 0; c0 < steps; c0 += chunk_steps, par ^= 1) {
        const int64_t s = steps - c0 < chunk_steps ? steps - c0 : chunk_steps;
        if (c0 + s < steps) ACAV_TRY(gather(c0 + s, par ^ 1));  // next chunk travels while this one trains
        if (train_here) ACAV_HIP_TRY(hipStreamWaitEvent((hipStream_t)st_train, c->ev_gathered[par], 0));
        int64_t nw = n_warm - warm_done;
        nw = nw < 0 ? 0 : (nw > s ? s : nw);
        if (train_here)
            ACAV_TRY(acav_kmeans_train(km, c->batches[par].as<float>(), s * bg, bg, lr, nw ? warm_global + warm_done * bg : nullptr, nw));
        warm_done += nw;
        if (train_here) ACAV_HIP_TRY(hipEventRecord(c->ev_trained[par], (hipStream_t)st_train));
    }
    if (nowait) return ACAV_OK;
    ACAV_HIP_TRY(hipStreamSynchronize(sc));
    return acav_kmeans_sync(km);
}

// The same epoch for SEVERAL clusterings over the same local rows (the views of one batch stream), chunk by chunk across
// all of them: per chunk every rank first enqueues the NEXT chunk's row exchange of every clustering (one communicator --
// and stream -- per clustering; the same order on every rank), then trains the current chunk of the clusterings dealt to
// it (train_here[v] != 0) side by side (acav_kmeans_train_multi).  A rank that blocks in its own clustering's chunk has
// therefore already fed every other clustering's exchange: different ranks train different clusterings at the same time.
// (Calling acav_kmeans_train_dp once per clustering serialises them: the training call of a chunk is synchronous, so the
// trainer of clustering 0 joins the exchanges of clustering 1 only after its whole epoch.)
ACAV_EXPORT int acav_kmeans_train_dp_multi(acav_kmeans *const *kms, acav_comm *const *comms, int count,
                                           const float *const *x_local_dev, int64_t n_local, int64_t b_local, double lr,
                                           const int64_t *const *warm_global, const int64_t *n_warm, int64_t chunk_steps,
                                           const int *train_here)
{
    ACAV_REQUIRE(kms && comms && x_local_dev && train_here && count > 0 && count <= 64, ACAV_EINVAL, "bad argument");
    ACAV_REQUIRE(n_local >= 0 && b_local > 0 && chunk_steps > 0, ACAV_EINVAL, "bad sizes");
    const int64_t steps = n_local / b_local;
    if (steps == 0) return ACAV_OK;
    std::vector<int> dv((size_t)count, 0), here((size_t)count, 0), roots((size_t)count, -1);
    std::vector<void *> st_train((size_t)count, nullptr);
    for (int v = 0; v < count; ++v) {  // train_here[v]: ACAV_DP_TRAIN | ACAV_DP_ROOTED | root << 8 (as `flags` of train_dp)
        here[(size_t)v] = (train_here[v] & 1) != 0;
        roots[(size_t)v] = (train_here[v] & 4) ? (train_here[v] >> 8) : -1;
    }
    const int w = comms[0] ? comms[0]->world : 1;
    const int64_t bg = (int64_t)w * b_local;
    for (int v = 0; v < count; ++v) {
        ACAV_REQUIRE(kms[v] && comms[v] && x_local_dev[v], ACAV_EINVAL, "clustering %d: NULL argument", v);
        ACAV_REQUIRE(comms[v]->world == w, ACAV_EINVAL, "communicators of different sizes");
        ACAV_REQUIRE(roots[(size_t)v] < 0 || here[(size_t)v] == (comms[v]->rank == roots[(size_t)v]), ACAV_EINVAL,
                     "clustering %d: ACAV_DP_ROOTED root %d does not match ACAV_DP_TRAIN on rank %d", v, roots[(size_t)v], comms[v]->rank);
        ACAV_REQUIRE(is_device_ptr(x_local_dev[v]), ACAV_EINVAL, "x_local must be device memory");
        ACAV_REQUIRE(!n_warm || n_warm[v] == 0 || (warm_global && warm_global[v]), ACAV_EINVAL, "clustering %d: warm-up labels missing", v);
        for (int e = 0; e < v; ++e) ACAV_REQUIRE(comms[e] != comms[v] && kms[e] != kms[v], ACAV_EINVAL, "one communicator and one handle per clustering");
        int K = 0;
        ACAV_TRY(acav_kmeans_shape(kms[v], &K, &dv[(size_t)v]));
        ACAV_REQUIRE((dv[(size_t)v] & 3) == 0, ACAV_EINVAL, "d = %d must be a multiple of 4 for the bulk exchange", dv[(size_t)v]);
        ACAV_TRY(acav_kmeans_stream(kms[v], &st_train[(size_t)v]));
        const size_t chunk_bytes = sizeof(float) * (size_t)chunk_steps * (size_t)bg * (size_t)dv[(size_t)v];
        ACAV_HIP_TRY(hipSetDevice(comms[v]->ctx.device));
        for (int q = 0; q < 2; ++q) {  // a rank that only sends needs no landing buffers
            if (here[(size_t)v] || roots[(size_t)v] < 0 || !rccl().p2p) ACAV_TRY(comms[v]->gather[q].ensure(chunk_bytes));
            if (here[(size_t)v]) ACAV_TRY(comms[v]->batches[q].ensure(chunk_bytes));
        }
    }
    auto gather = [&](int v, int64_t c0, int par) -> int {  // rows of steps [c0, c0 + s) of every rank -> batches[par] of clustering v
        acav_comm *c = comms[v];
        const int d = dv[(size_t)v];
        hipStream_t sc = c->ctx.stream;
        const int64_t s = steps - c0 < chunk_steps ? steps - c0 : chunk_steps;
        const size_t bytes = sizeof(float) * (size_t)s * (size_t)b_local * (size_t)d;
        if (here[(size_t)v]) ACAV_HIP_TRY(hipStreamWaitEvent(sc, c->ev_trained[par], 0));  // the chunk trained from this buffer is done
        ACAV_TRY(rows_to_trainers(rccl(), x_local_dev[v] + (size_t)c0 * b_local * d, c->gather[par].p, bytes, roots[(size_t)v], c->rank, w, c->comm, sc));
        if (here[(size_t)v]) {
            const int64_t total4 = (int64_t)w * s * b_local * (d / 4);
            const unsigned grid = (unsigned)(total4 / 256 + 1 < 4096 ? total4 / 256 + 1 : 4096);
            hipLaunchKernelGGL(k_interleave_rows, dim3(grid), dim3(256), 0, sc, c->gather[par].as<float4>(),
                               c->batches[par].as<float4>(), w, (int)s, (int)b_local, d / 4);
            ACAV_HIP_TRY(hipGetLastError());
        }
        ACAV_HIP_TRY(hipEventRecord(c->ev_gathered[par], sc));
        return ACAV_OK;
    };
    for (int v = 0; v < count; ++v) {
        if (here[(size_t)v])
            for (int q = 0; q < 2; ++q) ACAV_HIP_TRY(hipEventRecord(comms[v]->ev_trained[q], (hipStream_t)st_train[(size_t)v]));  // both buffers free
        ACAV_TRY(gather(v, 0, 0));
    }
    std::vector<int64_t> warm_done((size_t)count, 0);
    std::vector<acav_kmeans *> lk;
    std::vector<const float *> lx;
    std::vector<int64_t> ln, lnw;
    std::vector<const int64_t *> lw;
    int par = 0;
    for (int64_t c0 = 0; c0 < steps; c0 += chunk_steps, par ^= 1) {
        const int64_t s = steps - c0 < chunk_steps ? steps - c0 : chunk_steps;
        if (c0 + s < steps)
            for (int v = 0; v < count; ++v) ACAV_TRY(gather(v, c0 + s, par ^ 1));  // next chunk travels while this one trains
        lk.clear(), lx.clear(), ln.clear(), lnw.clear(), lw.clear();
        for (int v = 0; v < count; ++v) {
            if (!here[(size_t)v]) continue;
            ACAV_HIP_TRY(hipStreamWaitEvent((hipStream_t)st_train[(size_t)v], comms[v]->ev_gathered[par], 0));
            int64_t nw = (n_warm ? n_warm[v] : 0) - warm_done[(size_t)v];
            nw = nw < 0 ? 0 : (nw > s ? s : nw);
            lk.push_back(kms[v]);
            lx.push_back(comms[v]->batches[par].as<float>());
            ln.push_back(s * bg);
            lnw.push_back(nw);
            lw.push_back(nw ? warm_global[v] + warm_done[(size_t)v] * bg : nullptr);
            warm_done[(size_t)v] += nw;
        }
        if (!lk.empty()) {
            ACAV_TRY(acav_kmeans_train_multi(lk.data(), (int)lk.size(), lx.data(), ln.data(), bg, lr, lw.data(), lnw.data()));
            for (int v = 0; v < count; ++v)
                if (here[(size_t)v]) ACAV_HIP_TRY(hipEventRecord(comms[v]->ev_trained[par], (hipStream_t)st_train[(size_t)v]));
        }
    }
    for (int v = 0; v < count; ++v) {
        ACAV_HIP_TRY(hipStreamSynchronize(comms[v]->ctx.stream));
        ACAV_TRY(acav_kmeans_sync(kms[v]));
    }
    return ACAV_OK;
}

// One training epoch whose global batches are put together from rows that live on DIFFERENT ranks, by a plan
// (acav100m_amd/parallel/row_plan.py): step t's batch = for slot q = 0 .. slots-1 the rows [t lb, (t+1) lb) of slot q's
// stream, and a stream is a list of extents (owner rank, first local row, rows).  The reference's own N-GPU stream
// (mps/distributed.py:433-437 rotated shard order, data/clustering.py:25 per-rank batch), the one-GPU stream over
// partitioned rows and the large-batch mode of acav_kmeans_train_dp_multi are three plans.  Mechanics as there: per chunk
// of `chunk_steps` steps every rank packs the rows it owns (k_rows_by_table), sends them to the ONE rank that runs the
// clustering's chain (grouped ncclSend / ncclRecv on the clustering's own communicator and stream, a chunk ahead of the
// training), which places them into global batches and trains (acav_kmeans_train_multi: its clusterings side by side).
// ext: int64 [n_ext][4] = (slot, owner, first row, rows), the extents of a slot in stream order.  The same `ext`, `steps`,
// `chunk_steps` and `train_here` roots on every rank.  train_here[v]: ACAV_DP_TRAIN | ACAV_DP_ROOTED | root << 8.
ACAV_EXPORT int acav_kmeans_train_plan_multi(acav_kmeans *const *kms, acav_comm *const *comms, int count,
                                             const float *const *x_local_dev, int64_t n_local, int slots, int64_t lb,
                                             const int64_t *ext, int64_t n_ext, int64_t steps, double lr,
                                             const int64_t *const *warm_global, const int64_t *n_warm, int64_t chunk_steps,
                                             const int *train_here)
{
    ACAV_REQUIRE(kms && comms && x_local_dev && train_here && count > 0 && count <= 64, ACAV_EINVAL, "bad argument");
    ACAV_REQUIRE(n_local >= 0 && slots > 0 && slots <= 4096 && lb > 0 && lb * slots <= (1 << 20) && chunk_steps > 0 && steps >= 0 &&
                     n_ext >= 0 && (ext || n_ext == 0),
                 ACAV_EINVAL, "bad sizes");
    ACAV_REQUIRE(rccl().p2p, ACAV_ESTATE, "this RCCL has no ncclSend / ncclRecv: the plan-driven row exchange needs them");
    if (steps == 0) return ACAV_OK;
    const int w = comms[0] ? comms[0]->world : 1, me = comms[0] ? comms[0]->rank : 0;
    const int64_t bg = (int64_t)slots * lb;
    std::vector<int> dv((size_t)count, 0), here((size_t)count, 0), roots((size_t)count, -1);
    std::vector<void *> st_train((size_t)count, nullptr);
    for (int v = 0; v < count; ++v) {
        here[(size_t)v] = (train_here[v] & 1) != 0;
        ACAV_REQUIRE(train_here[v] & 4, ACAV_EINVAL, "clustering %d: the plan-driven epoch has exactly one trainer (ACAV_DP_ROOTED)", v);
        roots[(size_t)v] = train_here[v] >> 8;
        ACAV_REQUIRE(kms[v] && comms[v] && (x_local_dev[v] || n_local == 0), ACAV_EINVAL, "clustering %d: NULL argument", v);
        ACAV_REQUIRE(comms[v]->world == w && comms[v]->rank == me, ACAV_EINVAL, "communicators of different shapes");
        ACAV_REQUIRE(here[(size_t)v] == (me == roots[(size_t)v]), ACAV_EINVAL,
                     "clustering %d: ACAV_DP_ROOTED root %d does not match ACAV_DP_TRAIN on rank %d", v, roots[(size_t)v], me);
        ACAV_REQUIRE(n_local == 0 || is_device_ptr(x_local_dev[v]), ACAV_EINVAL, "x_local must be device memory");
        ACAV_REQUIRE(!n_warm || n_warm[v] == 0 || (warm_global && warm_global[v]), ACAV_EINVAL, "clustering %d: warm-up labels missing", v);
        for (int e = 0; e < v; ++e) ACAV_REQUIRE(comms[e] != comms[v] && kms[e] != kms[v], ACAV_EINVAL, "one communicator and one handle per clustering");
        int K = 0;
        ACAV_TRY(acav_kmeans_shape(kms[v], &K, &dv[(size_t)v]));
        ACAV_REQUIRE((dv[(size_t)v] & 3) == 0, ACAV_EINVAL, "d = %d must be a multiple of 4 for the bulk exchange", dv[(size_t)v]);
        ACAV_TRY(acav_kmeans_stream(kms[v], &st_train[(size_t)v]));
    }
    // ---- the plan: per slot its extents with their stream offsets
    struct Ext { int owner; int64_t first, rows, pos; };
    std::vector<std::vector<Ext>> streams((size_t)slots);
    for (int64_t i = 0; i < n_ext; ++i) {
        const int64_t q = ext[4 * i], owner = ext[4 * i + 1], first = ext[4 * i + 2], rows = ext[4 * i + 3];
        ACAV_REQUIRE(q >= 0 && q < slots && owner >= 0 && first >= 0 && rows >= 0, ACAV_EINVAL, "extent %lld is malformed", (long long)i);
        ACAV_REQUIRE(owner < w, ACAV_EINVAL, "extent %lld: owner %lld outside the communicator", (long long)i, (long long)owner);
        ACAV_REQUIRE(owner != me || first + rows <= n_local, ACAV_EINVAL, "extent %lld reaches past this rank's %lld rows", (long long)i, (long long)n_local);
        if (rows == 0) continue;
        auto &sv = streams[(size_t)q];
        sv.push_back({(int)owner, first, rows, sv.empty() ? 0 : sv.back().pos + sv.back().rows});
    }
    for (int q = 0; q < slots; ++q) {
        const int64_t have = streams[(size_t)q].empty() ? 0 : streams[(size_t)q].back().pos + streams[(size_t)q].back().rows;
        ACAV_REQUIRE(have >= steps * lb, ACAV_EINVAL, "slot %d's stream holds %lld rows, the epoch needs %lld", q, (long long)have, (long long)(steps * lb));
    }
    // ---- per chunk: the pieces every rank packs (in (slot, position) order) and where they land at the trainer
    struct Chunk { int64_t s, pack_off, pack_n, place_off, gb_off, my_rows, all_rows; std::vector<int64_t> from; };
    const int64_t n_chunks = (steps + chunk_steps - 1) / chunk_steps;
    std::vector<Chunk> chunks((size_t)n_chunks);
    std::vector<PlanPiece> table;
    std::vector<int> groups;
    std::vector<size_t> cursor((size_t)slots, 0);
    struct Pc { int slot, owner; int64_t rel, first, rows; };
    std::vector<Pc> pcs;
    int64_t max_my = 0, max_all = 0;
    for (int64_t ci = 0; ci < n_chunks; ++ci) {
        Chunk &ch = chunks[(size_t)ci];
        const int64_t c0 = ci * chunk_steps;
        ch.s = steps - c0 < chunk_steps ? steps - c0 : chunk_steps;
        const int64_t lo = c0 * lb, hi = (c0 + ch.s) * lb;
        pcs.clear();
        for (int q = 0; q < slots; ++q) {
            const auto &sv = streams[(size_t)q];
            size_t &e = cursor[(size_t)q];
            while (e < sv.size() && sv[e].pos + sv[e].rows <= lo) ++e;
            for (size_t k = e; k < sv.size() && sv[k].pos < hi; ++k) {
                const int64_t a = sv[k].pos > lo ? sv[k].pos : lo, b = sv[k].pos + sv[k].rows < hi ? sv[k].pos + sv[k].rows : hi;
                if (a < b) pcs.push_back({q, sv[k].owner, a - lo, sv[k].first + (a - sv[k].pos), b - a});
            }
        }
        ch.from.assign((size_t)w, 0);
        for (const Pc &p : pcs) ch.from[(size_t)p.owner] += p.rows;
        std::vector<int64_t> block((size_t)ch.from.size(), 0), fill((size_t)ch.from.size(), 0);
        for (size_t r = 1; r < block.size(); ++r) block[r] = block[r - 1] + ch.from[r - 1];
        ch.all_rows = ch.s * bg;
        ch.my_rows = ch.from[(size_t)me];
        // packing table of THIS rank (one group): key = row of the packed block
        ch.pack_off = (int64_t)table.size();
        int64_t filled = 0;
        for (const Pc &p : pcs)
            if (p.owner == me) {
                table.push_back({filled, p.first, p.rows});
                filled += p.rows;
            }
        ch.pack_n = (int64_t)table.size() - ch.pack_off;
        // placement table (groups = slots): key = position within the slot's chunk, src = row of the landing buffer
        ch.place_off = (int64_t)table.size();
        ch.gb_off = (int64_t)groups.size();
        int cur = -1;
        for (const Pc &p : pcs) {
            while (cur < p.slot) groups.push_back((int)((int64_t)table.size() - ch.place_off)), ++cur;
            const size_t r = (size_t)p.owner;
            table.push_back({p.rel, block[r] + fill[r], p.rows});
            fill[r] += p.rows;
        }
        while (cur < slots) groups.push_back((int)((int64_t)table.size() - ch.place_off)), ++cur;  // slots + 1 entries
        max_my = ch.my_rows > max_my ? ch.my_rows : max_my;
        max_all = ch.all_rows > max_all ? ch.all_rows : max_all;
    }
    // the tables of the whole epoch go to the device once (a few hundred KB), before anything is enqueued
    acav_comm *c0 = comms[0];
    ACAV_HIP_TRY(hipSetDevice(c0->ctx.device));
    ACAV_TRY(c0->plan_pieces.ensure(sizeof(PlanPiece) * (table.size() + 1)));
    // groups: per chunk [slots + 1] placement offsets, then per chunk [2] = {0, pack_n} for the packing launch
    const size_t gb_pack0 = groups.size();
    for (const Chunk &ch : chunks) groups.push_back(0), groups.push_back((int)ch.pack_n);
    ACAV_TRY(c0->plan_groups.ensure(sizeof(int) * (groups.size() + 1)));
    if (!table.empty()) ACAV_HIP_TRY(hipMemcpy(c0->plan_pieces.p, table.data(), sizeof(PlanPiece) * table.size(), hipMemcpyHostToDevice));
    ACAV_HIP_TRY(hipMemcpy(c0->plan_groups.p, groups.data(), sizeof(int) * groups.size(), hipMemcpyHostToDevice));
    const PlanPiece *tab_dev = c0->plan_pieces.as<PlanPiece>();
    const int *grp_dev = c0->plan_groups.as<int>();
    for (int v = 0; v < count; ++v) {
        const size_t row_bytes = sizeof(float) * (size_t)dv[(size_t)v];
        for (int q = 0; q < 2; ++q) {
            ACAV_TRY(comms[v]->pack[q].ensure(row_bytes * (size_t)(max_my > 0 ? max_my : 1)));
            if (here[(size_t)v]) {  // a rank that only sends needs no landing buffers
                ACAV_TRY(comms[v]->gather[q].ensure(row_bytes * (size_t)max_all));
                ACAV_TRY(comms[v]->batches[q].ensure(row_bytes * (size_t)max_all));
            }
        }
    }
    // Watchdog (round 6; first contact with a real multi-GPU node is the driver's): a rank never waits for a peer for ever.  Every
    // rank -- the ones that only feed a clustering's exchange too -- waits on the host for chunk ci's exchange before it goes on
    // (the next chunk is already enqueued, so nothing is serialised); a wait longer than ACAV_COMM_TIMEOUT_S seconds (default 120;
    // 0 = wait for ever) ends the call with the chunk, the clustering and the rank its chain runs on in acav_last_error().
    double timeout_s = 120.0;
    if (const char *vt = getenv("ACAV_COMM_TIMEOUT_S")) timeout_s = atof(vt);
    auto wait_exchange = [&](int v, int64_t ci, int par) -> int {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            const hipError_t e = hipEventQuery(comms[v]->ev_gathered[par]);
            if (e == hipSuccess) return ACAV_OK;
            (void)hipGetLastError();  // hipErrorNotReady stays behind as the thread's last error otherwise
            if (e != hipErrorNotReady) {
                acav::set_error("hipEventQuery failed while waiting for the row exchange: %s", hipGetErrorString(e));
                return ACAV_EHIP;
            }
            if (spins > 256) std::this_thread::sleep_for(std::chrono::microseconds(50));
            if (timeout_s > 0 && (spins & 1023) == 1023) {
                const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (waited > timeout_s) {
                    acav::set_error("rank %d of %d waited %.0f s for the rows of plan chunk %lld / %lld (steps %lld ..) of clustering %d, whose chain runs "
                                    "on rank %d: a peer never fed the exchange (crashed, blocked, or running another plan?) -- ACAV_COMM_TIMEOUT_S",
                                    me, w, waited, (long long)ci, (long long)n_chunks, (long long)(ci * chunk_steps), v, roots[(size_t)v]);
                    return ACAV_ETIMEOUT;
                }
            }
        }
    };
    auto gather = [&](int v, int64_t ci, int par) -> int {  // chunk ci of every owner -> batches[par] of clustering v
        acav_comm *c = comms[v];
        const Chunk &ch = chunks[(size_t)ci];
        const int d = dv[(size_t)v];
        hipStream_t sc = c->ctx.stream;
        if (here[(size_t)v]) ACAV_HIP_TRY(hipStreamWaitEvent(sc, c->ev_trained[par], 0));  // the chunk trained from this buffer is done
        launch_rows_by_table(false, tab_dev + ch.pack_off, grp_dev + gb_pack0 + 2 * (size_t)ci, x_local_dev[v], c->pack[par].as<float>(),
                             ch.my_rows, d, 1, 1, sc);
        ACAV_HIP_TRY(hipGetLastError());
        ACAV_TRY(packed_rows_to_root(rccl(), c->pack[par].p, ch.my_rows, c->gather[par].p, ch.from.data(), sizeof(float) * (size_t)d,
                                     roots[(size_t)v], me, w, c->comm, sc));
        if (here[(size_t)v]) {
            launch_rows_by_table(true, tab_dev + ch.place_off, grp_dev + ch.gb_off, c->gather[par].as<float>(), c->batches[par].as<float>(),
                                 ch.all_rows, d, slots, (int)lb, sc);
            ACAV_HIP_TRY(hipGetLastError());
        }
        ACAV_HIP_TRY(hipEventRecord(c->ev_gathered[par], sc));
        return ACAV_OK;
    };
    for (int v = 0; v < count; ++v) {
        ACAV_HIP_TRY(hipSetDevice(comms[v]->ctx.device));
        if (here[(size_t)v])
            for (int q = 0; q < 2; ++q) ACAV_HIP_TRY(hipEventRecord(comms[v]->ev_trained[q], (hipStream_t)st_train[(size_t)v]));  // both buffers free
        ACAV_TRY(gather(v, 0, 0));
    }
    std::vector<int64_t> warm_done((size_t)count, 0);
    std::vector<acav_kmeans *> lk;
    std::vector<const float *> lx;
    std::vector<int64_t> ln, lnw;
    std::vector<const int64_t *> lw;
    int par = 0;
    for (int64_t ci = 0; ci < n_chunks; ++ci, par ^= 1) {
        const int64_t s = chunks[(size_t)ci].s;
        if (ci + 1 < n_chunks)
            for (int v = 0; v < count; ++v) ACAV_TRY(gather(v, ci + 1, par ^ 1));  // next chunk travels while this one trains
        lk.clear(), lx.clear(), ln.clear(), lnw.clear(), lw.clear();
        for (int v = 0; v < count; ++v) ACAV_TRY(wait_exchange(v, ci, par));  // (bounded: see wait_exchange)
        for (int v = 0; v < count; ++v) {
            if (!here[(size_t)v]) continue;
            ACAV_HIP_TRY(hipStreamWaitEvent((hipStream_t)st_train[(size_t)v], comms[v]->ev_gathered[par], 0));
            int64_t nw = (n_warm ? n_warm[v] : 0) - warm_done[(size_t)v];
            nw = nw < 0 ? 0 : (nw > s ? s : nw);
            lk.push_back(kms[v]);
            lx.push_back(comms[v]->batches[par].as<float>());
            ln.push_back(s * bg);
            lnw.push_back(nw);
            lw.push_back(nw ? warm_global[v] + warm_done[(size_t)v] * bg : nullptr);
            warm_done[(size_t)v] += nw;
        }
        if (!lk.empty()) {
            ACAV_TRY(acav_kmeans_train_multi(lk.data(), (int)lk.size(), lx.data(), ln.data(), bg, lr, lw.data(), lnw.data()));
            for (int v = 0; v < count; ++v)
                if (here[(size_t)v]) ACAV_HIP_TRY(hipEventRecord(comms[v]->ev_trained[par], (hipStream_t)st_train[(size_t)v]));
        }
    }
    for (int v = 0; v < count; ++v) {
        ACAV_HIP_TRY(hipStreamSynchronize(comms[v]->ctx.stream));
        ACAV_TRY(acav_kmeans_sync(kms[v]));
    }
    return ACAV_OK;
}

// every rank's clustering state <- rank `root`'s (centres, usage counts, count, fallback): one RCCL broadcast of
// K d + K floats and 16 bytes
ACAV_EXPORT int acav_kmeans_broadcast_state(acav_kmeans *km, acav_comm *c, int root)
{
    ACAV_REQUIRE(km && c && root >= 0 && root < c->world, ACAV_EINVAL, "bad argument");
    int K = 0, d = 0;
    ACAV_TRY(acav_kmeans_shape(km, &K, &d));
    const int64_t nc = (int64_t)K * d, n = nc + K;
    ACAV_HIP_TRY(hipSetDevice(c->ctx.device));
    ACAV_TRY(c->flat.ensure(sizeof(float) * (size_t)(n + 4)));
    float *flat = c->flat.as<float>();
    int64_t sc[2] = {0, 0};
    ACAV_TRY(acav_kmeans_get_state(km, flat, flat + nc, &sc[0], &sc[1]));  // device destination; synchronises km's stream
    ACAV_HIP_TRY(hipMemcpyAsync(flat + n, sc, sizeof(sc), hipMemcpyHostToDevice, c->ctx.stream));
    ACAV_TRY(acav_comm_broadcast(c, flat, (int64_t)sizeof(float) * (n + 4), root));
    ACAV_HIP_TRY(hipMemcpyAsync(sc, flat + n, sizeof(sc), hipMemcpyDeviceToHost, c->ctx.stream));
    ACAV_HIP_TRY(hipStreamSynchronize(c->ctx.stream));
    if (c->rank == root) return ACAV_OK;
    return acav_kmeans_set_state(km, flat, flat + nc, sc[0], sc[1]);
}
