// Experiment (not product): which XCD does a workgroup land on (a) by block index in a plain launch, (b) on a stream created with
// hipExtStreamCreateWithCUMask for single-bit and per-XCD masks.  HW_REG_XCC_ID = hwreg 20 (bits 3:0), HW_REG_HW_ID = hwreg 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__global__ void k_where(unsigned *out)
{
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hw;
    }
    // stay a little so that blocks spread
    const long long t0 = clock64();
    while (clock64() - t0 < 20000) {}
}
int main()
{
    unsigned *d; CK(hipMalloc(&d, 8 * 4096));
    std::vector<unsigned> h(2 * 4096);
    hipLaunchKernelGGL(k_where, dim3(64), dim3(64), 0, 0, d);
    CK(hipMemcpy(h.data(), d, 8 * 64, hipMemcpyDeviceToHost));
    printf("plain launch, block -> xcc:");
    for (int b = 0; b < 64; ++b) printf(" %u", h[2 * b]);
    printf("\n");
    for (int bit = 0; bit < 40; ++bit) {
        unsigned mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        mask[bit / 32] = 1u << (bit % 32);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("bit %d: stream create failed\n", bit); (void)hipGetLastError(); continue; }
        hipLaunchKernelGGL(k_where, dim3(4), dim3(64), 0, s, d);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, 8 * 4, hipMemcpyDeviceToHost));
        printf("mask bit %2d -> xcc %u %u %u %u  hw_id %08x\n", bit, h[0], h[2], h[4], h[6], h[1]);
        CK(hipStreamDestroy(s));
    }
    // candidate per-XCD masks: bits i with i % 8 == x  vs  bits [32 x, 32 x + 32)
    for (int mode = 0; mode < 2; ++mode)
        for (int x = 0; x < 8; x += 3) {
            unsigned mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < 256; ++i)
                if (mode == 0 ? (i % 8 == x) : (i / 32 == x)) mask[i / 32] |= 1u << (i % 32);
            hipStream_t s;
            CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
            hipLaunchKernelGGL(k_where, dim3(64), dim3(64), 0, s, d);
            CK(hipStreamSynchronize(s));
            CK(hipMemcpy(h.data(), d, 8 * 64, hipMemcpyDeviceToHost));
            int cnt[8] = {0};
            for (int b = 0; b < 64; ++b) cnt[h[2 * b] & 7]++;
            printf("mask %s x=%d: blocks per xcc:", mode == 0 ? "i%8==x" : "i/32==x", x);
            for (int q = 0; q < 8; ++q) printf(" %d", cnt[q]);
            printf("\n");
            CK(hipStreamDestroy(s));
        }
    return 0;
}
