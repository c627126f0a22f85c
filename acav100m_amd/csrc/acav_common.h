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
// Streams and events a handle no longer needs: destroyed LATER, in bulk (when ACAV_RETIRE_MAX = 512 have gathered, at
// acav_trim_device_cache(), never in between).  hipStreamDestroy / hipEventDestroy take the runtime's locks for milliseconds: the ten handles of a lockstep group cost
// 45-90 ms to destroy, on the launching thread between two groups or -- from a helper thread -- under the next group's loop, which
// then runs at 35-46 us per lockstep iteration instead of 26.7 (tools/exp/NOTES_r06.md section 10).  A retired stream is idle
// (its owner synchronised it) and stays alive until the flush.
void retire_stream(hipStream_t s);
void retire_event(hipEvent_t e);
void flush_retired(bool force);
int devbuf_alloc(void **p, size_t *bytes, size_t want, int *dev = nullptr);
void devbuf_free(void *p, size_t bytes, int dev = -1);  // dev: the device the block was allocated on (-1: ask the runtime)
size_t devbuf_trim();

// A device allocation that frees itself.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int dev = -1;       // the device the block lives on (devbuf_free parks it without asking the runtime)
    hipStream_t owner = nullptr;  // bind(): the ONE stream every use of this buffer is ordered on (k-means handles)
    bool bound = false;
    ~DevBuf() { release(); }
    void bind(hipStream_t s) { owner = s, bound = true; }
    void release()
    {
        if (p) devbuf_free(p, bytes, dev);
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
        return devbuf_alloc(&p, &bytes, n, &dev);
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// Returns a device pointer for `src` (n bytes): src itself when it already is one, otherwise a
// staged copy in `stage` (async on stream).
int to_device(const void *src, size_t bytes, DevBuf &stage, hipStream_t stream, const void **out);
// dst <- host `src` (n bytes), async on stream (grows dst as needed)
int upload(DevBuf &dst, const void *src, size_t bytes, hipStream_t stream);
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
    void fini(bool retire = false);  // retire: the stream and events are destroyed later, in bulk (retire_stream)
    int timer_begin();
    int timer_end(float *ms);
};

}  // namespace acav
