// acav_common.h -- shared plumbing of libacav_hip.so (error model, device buffers, pointer kinds).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/acav_hip.h"

#define ACAV_EXPORT extern "C" __attribute__((visibility("default")))

// The timing-only ablation switches of the experiment harnesses (tools/exp) compile kernels that produce WRONG results on
// purpose (no MFMA, no DMA, sequential stand-ins for scattered stores).  They must never reach a product library through a
// stray -D (ACAV_EXTRA_HIPCC_FLAGS in __graft_entry__.build()): a build that defines one of them has to say so.
#if (defined(ACAV_ABL_NOAFRAG) || defined(ACAV_ABL_NOCDMA) || defined(ACAV_ABL_NOMFMA) || defined(ACAV_ABL_NOXDMA) || \
     defined(ACAV_ABL_NOXDMA_G123) || defined(ACAV_ABL_XHALF_G123) || \
     defined(ACAV_DBG_HALF_CAND_WINDOW) || defined(ACAV_FY_ABL_BKSEQ) || defined(ACAV_FY_ABL_NOWALK) || defined(ACAV_FY_ABL_SRCSEQ) || defined(ACAV_MI_ABL_EMPTY)) && \
    !defined(ACAV_EXPERIMENT_BUILD)
#error "ACAV_ABL_* / ACAV_FY_ABL_* / ACAV_MI_ABL_* select timing-only kernels with wrong results: experiment harnesses only (-DACAV_EXPERIMENT_BUILD)"
#endif

namespace acav {

void set_error(const char *fmt, ...);

#define ACAV_HIP_TRY(expr)                                                                         \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            acav::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return ACAV_EHIP;                                                                      \
        }                                                                                          \
    } while (0)

#define ACAV_REQUIRE(cond, code, ...)      \
    do {                                   \
        if (!(cond)) {                     \
            acav::set_error(__VA_ARGS__);  \
            return (code);                 \
        }                                  \
    } while (0)

#define ACAV_TRY(expr)             \
    do {                           \
        int _rc = (expr);          \
        if (_rc != ACAV_OK) return _rc; \
    } while (0)

// true when p is a device (or managed) pointer usable from kernels
bool is_device_ptr(const void *p);

// Blocks of up to 256 MB that a DevBuf gives back are parked (up to 16 GB / 16 384 blocks per process: ACAV_PARK_MAX_MB,
// ACAV_PARK_MAX_BLOCKS; acav_trim_device_cache() returns them) instead of hipFree'd, and handed
// out again to the next request of a similar size on the same device: hipMalloc / hipFree are synchronous and cost
// 50-200 us each -- a KMeans handle created per pass (bench.py, one CLI run per group) paid ~150 us of allocations inside
// its first assign sweep.  The owner of a DevBuf synchronises its stream before the buffer goes (all *_destroy do).
int devbuf_alloc(void **p, size_t *bytes, size_t want);
void devbuf_free(void *p, size_t bytes);
size_t devbuf_trim();

// Pinned host staging for host -> device copies (round 6).  hipMemcpyAsync from PAGEABLE memory pins the pages on the fly: 3.9 ms per
// 800 KB on this runtime (the candidate ids of a 100 k chunk; ten per lockstep group = 40 of a group's 50 ms of set-up), and pinning /
// unpinning from a helper thread updates the GPU's page tables under the kernels another thread has in flight.  A HostPinned block
// is hipHostMalloc'd once, parked when its owner goes (as device blocks are: hostpin_alloc / hostpin_free) and guarded by an event:
// the host may overwrite it only after the last copy that read it has completed (wait()).
int hostpin_alloc(void **p, size_t *bytes, size_t want);
void hostpin_free(void *p, size_t bytes);
struct HostPinned {
    void *p = nullptr;
    size_t bytes = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
    ~HostPinned() { release(); }
    void wait()
    {
        if (pending && ev) (void)hipEventSynchronize(ev);
        pending = false;
    }
    void release()
    {
        wait();
        if (p) hostpin_free(p, bytes);
        if (ev) (void)hipEventDestroy(ev);
        p = nullptr, bytes = 0, ev = nullptr;
    }
    int ensure(size_t n)
    {
        if (n <= bytes) return ACAV_OK;
        wait();
        if (p) hostpin_free(p, bytes);
        p = nullptr, bytes = 0;
        return hostpin_alloc(&p, &bytes, n);
    }
    int mark(hipStream_t s)  // a copy out of the block was just enqueued on s
    {
        if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            ev = nullptr;
            return hipStreamSynchronize(s) == hipSuccess ? ACAV_OK : ACAV_EHIP;
        }
        if (hipEventRecord(ev, s) != hipSuccess) return ACAV_EHIP;
        pending = true;
        return ACAV_OK;
    }
};
constexpr size_t HOSTPIN_MIN = 32u << 10, HOSTPIN_MAX = 64u << 20;  // copies of this size range go through a pinned shadow

// A device allocation that frees itself.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    HostPinned shadow;  // to_device(): the pinned staging of host data on its way into this buffer
    hipStream_t owner = nullptr;  // bind(): the ONE stream every use of this buffer is ordered on (k-means handles)
    bool bound = false;
    ~DevBuf() { release(); }
    void bind(hipStream_t s) { owner = s, bound = true; }
    void release()
    {
        if (p) devbuf_free(p, bytes);
        p = nullptr;
        bytes = 0;
    }
    int ensure(size_t n)
    {
        if (n <= bytes) return ACAV_OK;
        // Growing a buffer that kernels already in flight may still read or write (an asynchronous assign / step enqueued
        // with the old block): hipFree used to synchronise the device implicitly, a PARKED block is handed to the next
        // request at once -- possibly another handle on another stream.  Growth is rare (first use of a larger shape), so
        // it pays for the synchronisation the free no longer does: of the owning stream when the buffer is bound to one
        // (a k-means handle's buffers: growing clustering B's staging must not wait for clustering A's persistent epoch or a
        // pending send / receive of another communicator), else of the device.  Destructors run after the owner's *_destroy
        // has synchronised its stream.
        if (p) (void)(bound ? hipStreamSynchronize(owner) : hipDeviceSynchronize());
        release();
        return devbuf_alloc(&p, &bytes, n);
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// Returns a device pointer for `src` (n bytes): src itself when it already is one, otherwise a
// staged copy in `stage` (async on stream).
int to_device(const void *src, size_t bytes, DevBuf &stage, hipStream_t stream, const void **out);
// Copies a device result to `dst` (host or device), async on stream.
int from_device(void *dst, const void *src_dev, size_t bytes, hipStream_t stream);

// acav_mtjump.hip: g(t) = t^J mod phi(t) of MT19937 as 624 words (bit i = coefficient of t^i), cached; and the host
// evaluation window[0..624) <- stream words J ahead
const uint32_t *mt_jump_poly(int64_t J);
int mt_jump_window_host(uint32_t *window, int64_t J);

struct StreamCtx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // priority_class: 0 = default stream priority, +1 = the device's highest, -1 = its lowest (own streams only)
    int init(int dev, void *user_stream, int priority_class = 0);
    void fini();
    int timer_begin();
    int timer_end(float *ms);
};

}  // namespace acav
