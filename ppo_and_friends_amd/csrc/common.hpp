// Shared host/device helpers for libppoaf_hip.so (gfx950 only, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdarg>
#include <cstdio>

#include "../../include/ppoaf_hip.h"

namespace ppoaf {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return PPOAF_E_LAUNCH;
    }
    return PPOAF_OK;
}

#define PPOAF_REQUIRE(cond, ...)                     \
    do {                                             \
        if (!(cond)) {                               \
            ::ppoaf::set_error(__VA_ARGS__);         \
            return PPOAF_E_INVALID;                  \
        }                                            \
    } while (0)

// ---- wave64 / workgroup reductions --------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Sum over the workgroup; every thread gets the result.  `smem` holds >= 17
// elements of T (16 waves max + 1 broadcast slot).  blockDim.x multiple of 64.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* smem) {
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();                       // protect smem reuse across calls
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    if (wid == 0) {
        T x = (lane < nw) ? smem[lane] : T(0);
        x = wave_sum(x);
        if (lane == 0) smem[16] = x;
    }
    __syncthreads();
    return smem[16];
}

// Sum of n partials in a FIXED association, by one wave (all 64 lanes call it, all get the result): lane-strided
// ascending per lane, then the xor butterfly.  IEEE addition is commutative, so every lane, every wave, every run and
// every rank computes the bitwise identical total -- what clip coefficients of replicated optimisers must be built on.
__device__ __forceinline__ double ordered_partial_sum(const double* partials, unsigned n) {
    const unsigned lane = threadIdx.x & 63;
    double p = 0.0;
    for (unsigned b = lane; b < n; b += 64) p += __hip_atomic_load(&partials[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return wave_sum(p);
}

// ---- Philox4x32-10 (counter-based RNG; one 128-bit block per call) --------
struct Philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(uint64_t seed, uint64_t counter, uint32_t stream_id) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = stream_id, c3 = 0u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

// uniform in (0,1]: never 0 so log() is safe
__device__ __forceinline__ float u32_to_unit_open0(uint32_t u) {
    return ((float)(u >> 8) + 1.0f) * (1.0f / 16777216.0f);
}
// uniform in [0,1)
__device__ __forceinline__ float u32_to_unit(uint32_t u) {
    return (float)(u >> 8) * (1.0f / 16777216.0f);
}

}  // namespace ppoaf
