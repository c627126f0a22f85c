// acav_common.hip -- error model, device/pointer helpers, library-level entry points, and the
// host MT19937 stream (torch's CPU generator) of libacav_hip.so.
#include <map>
#include <mutex>

#include "acav_common.h"

namespace acav {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

bool is_device_ptr(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // unregistered host memory: clear the sticky error
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged ||
           attr.type == hipMemoryTypeUnified;
}

// ---- parked device blocks (see acav_common.h)
namespace {
// (device, bytes) -> block: the smallest parked block that fits is a lower_bound.  Round 6: the pool used to hold 256 blocks / 2 GB and
// was searched linearly -- a lockstep group of ten MI handles gives back ~1 000 blocks (80 position buffers per handle), so three
// quarters of them went through hipFree (a device-wide synchronisation each, issued from bench.py's helper thread while the next
// group's loop was running) and came back as hipMalloc in the next group's set-up: the cfg5 slice's selection took 2.5-3.7 s
// depending on how those calls interleaved.  Caps: ACAV_PARK_MAX_BLOCKS (16 384), ACAV_PARK_MAX_MB (16 384 MB; an out-of-memory
// hipMalloc gives everything back and retries, acav_trim_device_cache() gives it back on request).
std::mutex g_park_mutex;
std::multimap<std::pair<int, size_t>, void *> g_parked;
size_t g_parked_bytes = 0;
constexpr size_t PARK_MAX_BLOCK = 256ull << 20;
size_t park_max_total()
{
    static const size_t v = [] {
        const char *e = getenv("ACAV_PARK_MAX_MB");
        const long long mb = e ? atoll(e) : 16384;
        return (size_t)(mb < 0 ? 0 : mb) << 20;
    }();
    return v;
}
size_t park_max_blocks()
{
    static const size_t v = [] {
        const char *e = getenv("ACAV_PARK_MAX_BLOCKS");
        const long long n = e ? atoll(e) : 16384;
        return (size_t)(n < 0 ? 0 : n);
    }();
    return v;
}
void drop_all_parked()
{
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lock(g_park_mutex);
        for (auto &kv : g_parked) drop.push_back(kv.second);
        g_parked.clear();
        g_parked_bytes = 0;
    }
    for (void *q : drop) (void)hipFree(q);
}
}  // namespace

namespace {
std::mutex g_retired_mutex;
std::vector<hipStream_t> g_retired_streams;
std::vector<hipEvent_t> g_retired_events;
}  // namespace

void flush_retired(bool force)
{
    std::vector<hipStream_t> ss;
    std::vector<hipEvent_t> es;
    {
        std::lock_guard<std::mutex> lock(g_retired_mutex);
        static const size_t cap = [] { const char *e = getenv("ACAV_RETIRE_MAX"); return (size_t)(e && atoi(e) > 0 ? atoi(e) : 512); }();
        if (!force && g_retired_streams.size() + g_retired_events.size() < cap) return;
        ss.swap(g_retired_streams);
        es.swap(g_retired_events);
    }
    for (hipEvent_t e : es) (void)hipEventDestroy(e);
    for (hipStream_t q : ss) (void)hipStreamDestroy(q);
    (void)hipGetLastError();
}

void retire_stream(hipStream_t s)
{
    if (!s) return;
    {
        std::lock_guard<std::mutex> lock(g_retired_mutex);
        g_retired_streams.push_back(s);
    }
    flush_retired(false);
}

void retire_event(hipEvent_t e)
{
    if (!e) return;
    {
        std::lock_guard<std::mutex> lock(g_retired_mutex);
        g_retired_events.push_back(e);
    }
    flush_retired(false);
}

size_t devbuf_trim()
{
    flush_retired(true);
    size_t b;
    {
        std::lock_guard<std::mutex> lock(g_park_mutex);
        b = g_parked_bytes;
    }
    drop_all_parked();
    return b;
}

int devbuf_alloc(void **p, size_t *bytes, size_t want, int *dev_out)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev_out) *dev_out = dev;
    if (want <= PARK_MAX_BLOCK) {
        std::lock_guard<std::mutex> lock(g_park_mutex);
        auto it = g_parked.lower_bound({dev, want});  // smallest parked block of this device that fits ...
        if (it != g_parked.end() && it->first.first == dev && it->first.second <= 2 * want + 4096) {  // ... without wasting > 2x
            *p = it->second;
            *bytes = it->first.second;
            g_parked_bytes -= it->first.second;
            g_parked.erase(it);
            return ACAV_OK;
        }
    }
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) {  // out of memory: give the parked blocks back and try once more
        (void)hipGetLastError();
        drop_all_parked();
        e = hipMalloc(p, want);
    }
    if (e != hipSuccess) {
        *p = nullptr;
        *bytes = 0;
        set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return ACAV_ENOMEM;
    }
    *bytes = want;
    return ACAV_OK;
}

void devbuf_free(void *p, size_t bytes, int dev)
{
    if (!p) return;
    if (bytes <= PARK_MAX_BLOCK) {
        if (dev < 0) {
            hipPointerAttribute_t attr;
            if (hipPointerGetAttributes(&attr, p) == hipSuccess) dev = attr.device;
            else (void)hipGetLastError();
        }
        std::lock_guard<std::mutex> lock(g_park_mutex);
        if (dev >= 0 && g_parked_bytes + bytes <= park_max_total() && g_parked.size() < park_max_blocks()) {
            g_parked.emplace(std::make_pair(dev, bytes), p);
            g_parked_bytes += bytes;
            return;
        }
    }
    (void)hipFree(p);
}

int to_device(const void *src, size_t bytes, DevBuf &stage, hipStream_t stream, const void **out)
{
    if (bytes == 0 || is_device_ptr(src)) {
        *out = src;
        return ACAV_OK;
    }
    ACAV_TRY(stage.ensure(bytes));
    ACAV_HIP_TRY(hipMemcpyAsync(stage.p, src, bytes, hipMemcpyHostToDevice, stream));
    *out = stage.p;
    return ACAV_OK;
}

int upload(DevBuf &dst, const void *src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return ACAV_OK;
    ACAV_TRY(dst.ensure(bytes));
    ACAV_HIP_TRY(hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, stream));
    return ACAV_OK;
}

int from_device(void *dst, const void *src_dev, size_t bytes, hipStream_t stream)
{
    if (bytes == 0 || dst == src_dev) return ACAV_OK;
    hipMemcpyKind kind = is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    ACAV_HIP_TRY(hipMemcpyAsync(dst, src_dev, bytes, kind, stream));
    return ACAV_OK;
}

int StreamCtx::init(int dev, void *user_stream, int priority_class)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        (void)hipGetLastError();
        set_error("no HIP device available: this library does not fall back to the CPU");
        return ACAV_EHIP;
    }
    ACAV_REQUIRE(dev >= 0 && dev < n, ACAV_EINVAL, "device %d out of range (have %d)", dev, n);
    device = dev;
    ACAV_HIP_TRY(hipSetDevice(dev));
    if (user_stream) {
        stream = static_cast<hipStream_t>(user_stream);
        own_stream = false;
    } else {
        if (priority_class == 0) {
            ACAV_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        } else {  // (see acav_mi_create: a priority class has its own pool of hardware queues)
            int lo = 0, hi = 0;
            ACAV_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
            ACAV_HIP_TRY(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, priority_class > 0 ? hi : lo));
        }
        own_stream = true;
    }
    ACAV_HIP_TRY(hipEventCreate(&ev0));
    ACAV_HIP_TRY(hipEventCreate(&ev1));
    return ACAV_OK;
}

void StreamCtx::fini(bool retire)
{
    if (!retire) {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
        ev0 = ev1 = nullptr;
        stream = nullptr;
        return;
    }
    retire_event(ev0);
    retire_event(ev1);
    if (own_stream && stream) {
        (void)hipStreamSynchronize(stream);
        retire_stream(stream);
    }
    ev0 = ev1 = nullptr;
    stream = nullptr;
}

int StreamCtx::timer_begin()
{
    ACAV_HIP_TRY(hipEventRecord(ev0, stream));
    return ACAV_OK;
}

int StreamCtx::timer_end(float *ms)
{
    ACAV_HIP_TRY(hipEventRecord(ev1, stream));
    ACAV_HIP_TRY(hipEventSynchronize(ev1));
    ACAV_HIP_TRY(hipEventElapsedTime(ms, ev0, ev1));
    return ACAV_OK;
}

}  // namespace acav

using namespace acav;

ACAV_EXPORT const char *acav_last_error(void) { return g_err; }
ACAV_EXPORT int acav_version(void) { return 100; }

ACAV_EXPORT int acav_trim_device_cache(int64_t *freed_bytes)
{
    const size_t b = acav::devbuf_trim();
    if (freed_bytes) *freed_bytes = (int64_t)b;
    return ACAV_OK;
}

ACAV_EXPORT int acav_device_count(int *count)
{
    ACAV_REQUIRE(count, ACAV_EINVAL, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return ACAV_OK;
}

ACAV_EXPORT int acav_device_info(int device, char *name, int name_len, int *compute_units, int64_t *hbm_bytes)
{
    hipDeviceProp_t prop;
    ACAV_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return ACAV_OK;
}

// ------------------------------------------------------------------------------------ rng
// MT19937 exactly as at::mt19937 (torch.manual_seed -> init_genrand; random() = genrand_int32).
struct acav_rng {
    uint32_t mt[624];
    int idx;
    void seed(uint32_t s)
    {
        mt[0] = s;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    void refill()
    {
        for (int k = 0; k < 624; ++k) {
            uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        idx = 0;
    }
    inline uint32_t next()
    {
        if (idx >= 624) refill();
        uint32_t y = mt[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    inline float next_f32() { return (float)(next() & 0xFFFFFFu) * (1.0f / 16777216.0f); }
};

ACAV_EXPORT int acav_rng_create(acav_rng **out, uint32_t seed)
{
    ACAV_REQUIRE(out, ACAV_EINVAL, "out is NULL");
    acav_rng *r = new (std::nothrow) acav_rng;
    ACAV_REQUIRE(r, ACAV_ENOMEM, "out of host memory");
    r->seed(seed);
    *out = r;
    return ACAV_OK;
}
ACAV_EXPORT int acav_rng_destroy(acav_rng *rng)
{
    delete rng;
    return ACAV_OK;
}
ACAV_EXPORT int acav_rng_seed(acav_rng *rng, uint32_t seed)
{
    ACAV_REQUIRE(rng, ACAV_EINVAL, "rng is NULL");
    rng->seed(seed);
    return ACAV_OK;
}
ACAV_EXPORT int acav_rng_u32(acav_rng *rng, uint32_t *out)
{
    ACAV_REQUIRE(rng && out, ACAV_EINVAL, "NULL argument");
    *out = rng->next();
    return ACAV_OK;
}
ACAV_EXPORT int acav_rng_rand_f32(acav_rng *rng, float *out_host, int64_t n)
{
    ACAV_REQUIRE(rng && (out_host || n == 0) && n >= 0, ACAV_EINVAL, "bad argument");
    for (int64_t i = 0; i < n; ++i) out_host[i] = rng->next_f32();
    return ACAV_OK;
}
ACAV_EXPORT int acav_rng_randperm(acav_rng *rng, int64_t n, int64_t *out_host)
{
    ACAV_REQUIRE(rng && (out_host || n == 0) && n >= 0, ACAV_EINVAL, "bad argument");
    for (int64_t i = 0; i < n; ++i) out_host[i] = i;
    for (int64_t i = 0; i + 1 < n; ++i) {
        const int64_t z = (int64_t)(rng->next() % (uint64_t)(n - i));
        const int64_t t = out_host[i];
        out_host[i] = out_host[i + z];
        out_host[i + z] = t;
    }
    return ACAV_OK;
}
// Python's random.shuffle(list(range(n))) on this generator's stream (CPython Lib/random.py: for i = n-1 .. 1:
// j = _randbelow(i + 1), swap; _randbelow_with_getrandbits draws getrandbits(bit_length(i + 1)) =
// genrand_uint32() >> (32 - k) until it is below i + 1).  The selection stage shuffles its candidate list with
// Python's generator (run_greedy.py:40): a million-element list costs the interpreter 0.6 s, this loop 5 ms.
ACAV_EXPORT int acav_rng_py_shuffle(acav_rng *rng, int64_t n, int64_t *out_host)
{
    ACAV_REQUIRE(rng && (out_host || n == 0) && n >= 0 && n <= 0xffffffffll, ACAV_EINVAL, "bad argument");
    for (int64_t i = 0; i < n; ++i) out_host[i] = i;
    for (int64_t i = n - 1; i >= 1; --i) {
        const uint32_t m = (uint32_t)(i + 1);
        const int k = 32 - __builtin_clz(m);  // bit_length
        uint32_t r;
        do r = rng->next() >> (32 - k); while (r >= m);
        const int64_t t = out_host[i];
        out_host[i] = out_host[r];
        out_host[r] = t;
    }
    return ACAV_OK;
}
ACAV_EXPORT int acav_rng_get_state(const acav_rng *rng, uint32_t *mt624, int *idx)
{
    ACAV_REQUIRE(rng && mt624 && idx, ACAV_EINVAL, "NULL argument");
    memcpy(mt624, rng->mt, sizeof(rng->mt));
    *idx = rng->idx;
    return ACAV_OK;
}
ACAV_EXPORT int acav_rng_set_state(acav_rng *rng, const uint32_t *mt624, int idx)
{
    ACAV_REQUIRE(rng && mt624 && idx >= 0 && idx <= 624, ACAV_EINVAL, "bad argument");
    memcpy(rng->mt, mt624, sizeof(rng->mt));
    rng->idx = idx;
    return ACAV_OK;
}
ACAV_EXPORT int acav_rng_warmup_best(acav_rng *rng, int k, int64_t b, int64_t *best_host, float *mean_out)
{
    ACAV_REQUIRE(rng && best_host && k > 0 && b > 0, ACAV_EINVAL, "bad argument");
    // distances = torch.rand(k, b) row-major; min over axis 0, first index on ties
    std::vector<float> minv((size_t)b);
    for (int64_t i = 0; i < b; ++i) {
        minv[(size_t)i] = rng->next_f32();
        best_host[i] = 0;
    }
    for (int kk = 1; kk < k; ++kk) {
        for (int64_t i = 0; i < b; ++i) {
            const float t = rng->next_f32();
            if (t < minv[(size_t)i]) {
                minv[(size_t)i] = t;
                best_host[i] = kk;
            }
        }
    }
    if (mean_out) {
        double s = 0.0;
        for (int64_t i = 0; i < b; ++i) s += minv[(size_t)i];
        *mean_out = (float)(s / (double)b);
    }
    return ACAV_OK;
}
