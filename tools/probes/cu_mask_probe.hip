// Which XCDs does a CU-masked stream dispatch to?  (hipExtStreamCreateWithCUMask; bit k of the mask <-> which CU?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void where(unsigned* hist, long long spin) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) atomicAdd(&hist[v & 0xf], 1u);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); return; }
    unsigned* hist; CK(hipMalloc(&hist, 64)); CK(hipMemset(hist, 0, 64));
    hipLaunchKernelGGL(where, dim3(512), dim3(256), 0, s, hist, 2000LL);   // 20 us of spinning per workgroup
    CK(hipStreamSynchronize(s));
    unsigned h[16]; CK(hipMemcpy(h, hist, 64, hipMemcpyDeviceToHost));
    printf("%-44s XCD histogram of 512 workgroups:", name);
    for (int i = 0; i < 8; ++i) printf(" %3u", h[i]);
    printf("\n");
    CK(hipFree(hist)); CK(hipStreamDestroy(s));
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("%s: %d CUs\n", p.name, p.multiProcessorCount);
    const int words = 8;
    std::vector<uint32_t> all(words, 0xffffffffu), interleaved(words, 0u), first64(words, 0xffffffffu), last(words, 0u);
    for (int k = 0; k < 32 * words; ++k) if (k % 8 >= 2) interleaved[k / 32] |= 1u << (k % 32);
    first64[0] = 0; first64[1] = 0;
    for (int k = 0; k < 32 * words; ++k) if (k % 8 == 7) last[k / 32] |= 1u << (k % 32);
    run("all CUs", all);
    run("bits with k % 8 in {0, 1} cleared", interleaved);
    run("bits 0..63 cleared", first64);
    run("only bits with k % 8 == 7", last);
    return 0;
}
