// Which LDS sizes let a second kernel's workgroups become resident BESIDE a kernel that holds one 135 744-byte workgroup on every CU?
// Kernel A (256 x 256 threads, A_LDS bytes of dynamic LDS) spins for ~2 ms; kernel B (256 x 256 threads, Y bytes) spins ~2 ms on another
// stream.  Co-resident: both done after ~2 ms; not: ~4 ms.   usage: colds_probe [A_LDS [VGPRs of B: 240 | 128 | 0]]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>

template <int VG>
__global__ __launch_bounds__(256) void spin(long long ticks, int *sink)
{
    extern __shared__ int lds[];
    if (VG == 252) asm volatile("v_mov_b32 v251, 0" ::: "v251");  // the register footprints of the two training kernels
    if (VG == 240) asm volatile("v_mov_b32 v239, 0" ::: "v239");
    if (VG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x + 1) & 255] == -1) *sink = 1;
}

int main(int argc, char **argv)
{
    const int a_lds = argc > 1 ? atoi(argv[1]) : 135744;
    int *sink;
    hipMalloc(&sink, 4);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const int vb = argc > 2 ? atoi(argv[2]) : 240;
    auto ka = spin<252>;
    auto kb = vb == 240 ? spin<240> : vb == 128 ? spin<128> : spin<0>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(ka), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void *>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const long long ticks = 200000;  // 100 MHz wall clock: 2 ms
    for (int warm = 0; warm < 2; ++warm) {
        hipLaunchKernelGGL(ka, dim3(256), dim3(256), a_lds, s1, ticks, sink);
        hipDeviceSynchronize();
    }
    for (int y : {1024, 16384, 25344, 26624, 27136}) {
        if (a_lds + y > 160 * 1024) continue;
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(ka, dim3(256), dim3(256), a_lds, s1, ticks, sink);
        hipLaunchKernelGGL(kb, dim3(256), dim3(256), y, s2, ticks, sink);
        hipDeviceSynchronize();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("A (252 VGPRs) %d B + B (%d VGPRs) %d B = %d (%d left of 163840): %.2f ms -> %s\n", a_lds, vb, y, a_lds + y, 163840 - a_lds - y, ms,
               ms < 3.0 ? "side by side" : "one after the other");
    }
    return 0;
}
