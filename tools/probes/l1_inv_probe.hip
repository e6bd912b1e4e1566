// After a flag barrier inside one XCD: are PLAIN loads of data another CU rewrote correct (and fast) when preceded by a
// vector-L1 invalidate (buffer_inv sc1), instead of making every such load L1-bypassing (nt)?
//   32 workgroups on one XCD; iteration it: every worker rewrites its 1/32 share of a bucket with the value `it`,
//   barrier, every worker reads the WHOLE bucket and checks that all values equal `it`, barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 512, kWorkers = 32;
struct Ctl { unsigned tickets, error, mismatches, pad[29]; unsigned flags[64]; };
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }
__device__ __forceinline__ void barrier(Ctl* c, int w, unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) *(volatile unsigned*)&c->flags[w] = epoch;
    if (threadIdx.x < 64) {
        const long long t0 = wall_clock64();
        while (true) {
            unsigned f = epoch;
            if (threadIdx.x < kWorkers) f = __hip_atomic_load(&c->flags[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((int)(f >= epoch))) break;
            if (wall_clock64() - t0 > 200000000LL) { if (threadIdx.x == 0) c->error = 1; break; }
        }
    }
    __syncthreads();
}
// MODE 0: nt loads; 1: plain loads, no invalidate (expected WRONG); 2: plain loads after buffer_inv sc1; 3: after buffer_inv sc0 sc1
template <int MODE>
__global__ __launch_bounds__(kThreads) void k(float* w, long n4, int iters, Ctl* c, int target_xcc) {
    __shared__ int s_ticket;
    if (xcc_id() != (unsigned)target_xcc) return;
    if (threadIdx.x == 0) s_ticket = (int)atomicAdd(&c->tickets, 1u);
    __syncthreads();
    const int wk = s_ticket;
    if (wk >= kWorkers) return;
    unsigned epoch = 0, bad = 0;
    const long share = (n4 + kWorkers - 1) / kWorkers;
    f4* p = reinterpret_cast<f4*>(w);
    for (int it = 1; it <= iters; ++it) {
        for (long i = wk * share + threadIdx.x; i < (wk + 1) * share && i < n4; i += kThreads) { f4 v = {(float)it, (float)it, (float)it, (float)it}; p[i] = v; }
        barrier(c, wk, ++epoch);
        if (MODE == 2) asm volatile("buffer_inv sc1" ::: "memory");
        if (MODE == 3) asm volatile("buffer_inv sc0 sc1" ::: "memory");
        for (long i = threadIdx.x; i < n4; i += kThreads) {
            f4 v;
            if (MODE == 0) v = __builtin_nontemporal_load(p + i); else v = p[i];
            if (v[0] != (float)it || v[3] != (float)it) ++bad;
        }
        barrier(c, wk, ++epoch);
    }
    if (bad) atomicAdd(&c->mismatches, bad);
}
template <int MODE> void run(const char* name, float* w, long n4, int iters, Ctl* c) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipMemset(c, 0, sizeof(Ctl))); CK(hipMemset(w, 0, n4 * 16));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(kThreads), 100 * 1024, 0, w, n4, iters, c, 0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    Ctl h; CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
    printf("%-44s %7.2f us / iteration, stale or wrong values seen: %u, error %u\n", name, ms * 1e3 / iters, h.mismatches, h.error);
}
int main(int argc, char** argv) {
    const long bytes = argc > 1 ? atol(argv[1]) : 271 * 1024;
    const long n4 = bytes / 16; const int iters = 2000;
    float* w; Ctl* c; CK(hipMalloc(&w, n4 * 16)); CK(hipMalloc(&c, sizeof(Ctl)));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    printf("bucket %ld B, 32 workers on XCD 0, %d iterations (write share, barrier, read all, barrier)\n", bytes, iters);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("nt loads", w, n4, iters, c);
        run<1>("plain loads, no invalidate", w, n4, iters, c);
        run<2>("plain loads after buffer_inv sc1", w, n4, iters, c);
        run<3>("plain loads after buffer_inv sc0 sc1", w, n4, iters, c);
    }
    return 0;
}
