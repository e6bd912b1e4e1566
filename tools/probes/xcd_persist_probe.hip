// Feasibility probe for a single-XCD persistent update kernel (never shipped):
//   32 workgroups on ONE XCD (found by HW_REG_XCC_ID + tickets), phases separated by a flag barrier that lives in that XCD's L2:
//   phase A: every worker streams the whole "weight" bucket (BYTES) with L1-bypassing loads (nt),
//   phase B: every worker rewrites its 1/32 share of the bucket with plain stores (line stays in the XCD's L2).
// Reports us per iteration, and the same work as two kernels per iteration (launch boundaries as barriers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 512, kWorkers = 32;

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

struct Ctl { unsigned tickets; unsigned error; unsigned pad[30]; unsigned flags[64]; };

__device__ __forceinline__ bool barrier(Ctl* c, int w, unsigned epoch, long long budget) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        *(volatile unsigned*)&c->flags[w] = epoch;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    bool ok = true;
    if (threadIdx.x < 64) {
        const long long t0 = wall_clock64();
        while (true) {
            unsigned f = epoch;
            if (threadIdx.x < kWorkers) f = __hip_atomic_load(&c->flags[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool all = __all((int)(f >= epoch));
            if (all) break;
            if (wall_clock64() - t0 > budget) { ok = false; if (threadIdx.x == 0) c->error = 1; break; }
        }
    }
    __syncthreads();
    return ok;
}

template <bool NT>
__device__ __forceinline__ float stream_read(const float* w, long n4, int tid) {
    const f4* p = reinterpret_cast<const f4*>(w);
    f4 acc = {0, 0, 0, 0};
    for (long i = tid; i < n4; i += kThreads * 4) {
        f4 a, b = {0,0,0,0}, c = {0,0,0,0}, d = {0,0,0,0};
        if (NT) {
            a = __builtin_nontemporal_load(p + i);
            if (i + kThreads < n4) b = __builtin_nontemporal_load(p + i + kThreads);
            if (i + 2 * kThreads < n4) c = __builtin_nontemporal_load(p + i + 2 * kThreads);
            if (i + 3 * kThreads < n4) d = __builtin_nontemporal_load(p + i + 3 * kThreads);
        } else {
            a = p[i];
            if (i + kThreads < n4) b = p[i + kThreads];
            if (i + 2 * kThreads < n4) c = p[i + 2 * kThreads];
            if (i + 3 * kThreads < n4) d = p[i + 3 * kThreads];
        }
        acc += a + b + c + d;
    }
    return acc.x + acc.y + acc.z + acc.w;
}

__global__ __launch_bounds__(kThreads) void persistent(float* w, long n4, int iters, Ctl* c, float* sink, int target_xcc) {
    extern __shared__ float smem[];
    __shared__ int s_ticket;
    if (xcc_id() != (unsigned)target_xcc) return;
    if (threadIdx.x == 0) s_ticket = (int)atomicAdd(&c->tickets, 1u);
    __syncthreads();
    const int wk = s_ticket;
    if (wk >= kWorkers) return;
    float acc = 0.f;
    unsigned epoch = 0;
    const long share = (n4 + kWorkers - 1) / kWorkers;
    for (int it = 0; it < iters; ++it) {
        acc += stream_read<true>(w, n4, threadIdx.x);
        if (!barrier(c, wk, ++epoch, 200000000LL)) break;
        f4* p = reinterpret_cast<f4*>(w);
        for (long i = wk * share + threadIdx.x; i < (wk + 1) * share && i < n4; i += kThreads) {
            f4 v = {acc * 1e-30f + (float)it, 1.f, 2.f, 3.f};
            p[i] = v;
        }
        if (!barrier(c, wk, ++epoch, 200000000LL)) break;
    }
    if (acc == 1.2345f) sink[0] = acc;
    smem[threadIdx.x] = acc;
}

__global__ __launch_bounds__(kThreads) void phaseA(const float* w, long n4, float* sink) {
    extern __shared__ float smem[];
    const float acc = stream_read<false>(w, n4, threadIdx.x);
    if (acc == 1.2345f) sink[0] = acc;
    smem[threadIdx.x] = acc;
}
__global__ __launch_bounds__(kThreads) void phaseB(float* w, long n4, int it) {
    const long share = (n4 + gridDim.x - 1) / gridDim.x;
    f4* p = reinterpret_cast<f4*>(w);
    for (long i = blockIdx.x * share + threadIdx.x; i < (blockIdx.x + 1) * share && i < n4; i += kThreads) {
        f4 v = {(float)it, 1.f, 2.f, 3.f};
        p[i] = v;
    }
}

int main(int argc, char** argv) {
    const long bytes = argc > 1 ? atol(argv[1]) : 271 * 1024;
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    const long n4 = bytes / 16;
    float *w, *sink; Ctl* c;
    CK(hipMalloc(&w, n4 * 16)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&c, sizeof(Ctl)));
    CK(hipMemset(w, 0, n4 * 16));
    const size_t lds = 100 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(persistent), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(phaseA), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(c, 0, sizeof(Ctl)));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(persistent, dim3(256), dim3(kThreads), lds, 0, w, n4, iters, c, sink, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        Ctl h; CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
        printf("persistent (1 XCD, %d workers): %ld B bucket, %d iters: %.2f us / iteration (read all + barrier + rewrite + barrier); tickets %u error %u\n",
               kWorkers, bytes, iters, ms * 1e3 / iters, h.tickets, h.error);
    }
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; ++it) {
            hipLaunchKernelGGL(phaseA, dim3(32), dim3(kThreads), lds, 0, w, n4, sink);
            hipLaunchKernelGGL(phaseB, dim3(66), dim3(256), 0, 0, w, n4, it);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("two launches per iteration (32 readers, 66 writers): %.2f us / iteration\n", ms * 1e3 / iters);
    }
    return 0;
}
