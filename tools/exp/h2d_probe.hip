// Experiment (not product): what does a host -> device copy of PAGEABLE memory cost on this runtime, asynchronous on a stream
// (hipMemcpyAsync) against synchronous (hipMemcpy), per size?   ./h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
int main()
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    void *d;
    CK(hipMalloc(&d, 64 << 20));
    using clk = std::chrono::steady_clock;
    for (size_t bytes : {(size_t)64 << 10, (size_t)800 << 10, (size_t)8 << 20, (size_t)32 << 20}) {
        std::vector<char> h(bytes, 1);
        void *hp;
        CK(hipHostMalloc(&hp, bytes, hipHostMallocDefault));
        memset(hp, 1, bytes);
        double t_async = 1e9, t_sync = 1e9, t_pinned = 1e9, t_fresh = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            auto t0 = clk::now();
            CK(hipMemcpyAsync(d, h.data(), bytes, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            double ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
            t_async = ms < t_async ? ms : t_async;
            t0 = clk::now();
            CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
            ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
            t_sync = ms < t_sync ? ms : t_sync;
            t0 = clk::now();
            CK(hipMemcpyAsync(d, hp, bytes, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
            t_pinned = ms < t_pinned ? ms : t_pinned;
            std::vector<char> f(bytes, 2);  // a buffer the runtime has never seen
            t0 = clk::now();
            CK(hipMemcpyAsync(d, f.data(), bytes, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            ms = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
            t_fresh = ms < t_fresh ? ms : t_fresh;
        }
        printf("%8zu KB: hipMemcpyAsync pageable %.3f ms (a fresh buffer each time %.3f), hipMemcpy pageable %.3f ms, hipMemcpyAsync pinned %.3f ms\n", bytes >> 10,
               t_async, t_fresh, t_sync, t_pinned);
        CK(hipHostFree(hp));
    }
    return 0;
}
