// Device building blocks shared by the fused update (ppo_update.hip) and the rollout step
// (policy_step.hip): network descriptors, activations, f32-MFMA forward tiles.
#pragma once
#include "common.hpp"
#include <cfloat>

namespace ppoaf {

constexpr int kRows = PPOAF_UPDATE_ROWS_PER_WG;   // 16 rows per workgroup = one MFMA M tile
constexpr int kMaxLayers = 8;
constexpr int kMaxOut = 16;
constexpr int kMaxAdvLds = 4096;
constexpr int kThreads = 256;

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct NetDev {
    int in_dim, H, depth, out_dim, act;
    long offset, size, log_std_off;
    long offW[kMaxLayers], offB[kMaxLayers];      // relative to this network's bucket
};

__device__ __forceinline__ float act_fwd(float z, int act) {
    if (act == PPOAF_ACT_RELU) return fmaxf(z, 0.f);
    if (act == PPOAF_ACT_LEAKY_RELU) return z > 0.f ? z : 0.01f * z;
    return tanhf(z);
}
// derivative from the POST-activation value (what autograd's backward kernels use as well)
__device__ __forceinline__ float act_bwd(float h, int act) {
    if (act == PPOAF_ACT_RELU) return h > 0.f ? 1.f : 0.f;
    if (act == PPOAF_ACT_LEAKY_RELU) return h > 0.f ? 1.f : 0.01f;
    return 1.f - h * h;
}

// ---- 16-lane (DPP "row") reductions.  `__shfl_xor` compiles to ds_bpermute_b32 -- a round trip through the LDS
//      hardware, ~100+ cycles, four of them dependent per sum (K15 carried 324 of them: a fifth of its wave cycles).  The
//      same xor butterfly on DPP row operations is four VALU instructions: xor 8 = row_ror:8 exactly; after it the values
//      are 8-periodic inside the row, so lane i ^ 4 holds what row_ror:4 delivers (and likewise 2 / 1, which are quad
//      permutes anyway): the same additions of the same operand pairs -- bitwise the __shfl_xor result.  Every lane of the
//      wave must be active at the call (all call sites are whole waves).
template <int CTRL> __device__ __forceinline__ float dpp_row(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
constexpr int kDppRor8 = 0x128, kDppRor4 = 0x124, kDppXor2 = 0x4E /* quad_perm [2,3,0,1] */, kDppXor1 = 0xB1 /* [1,0,3,2] */;
__device__ __forceinline__ float group16_sum(float v) {
    v += dpp_row<kDppRor8>(v); v += dpp_row<kDppRor4>(v);
    v += dpp_row<kDppXor2>(v); v += dpp_row<kDppXor1>(v);
    return v;
}
__device__ __forceinline__ float group16_max(float v) {
    v = fmaxf(v, dpp_row<kDppRor8>(v)); v = fmaxf(v, dpp_row<kDppRor4>(v));
    v = fmaxf(v, dpp_row<kDppXor2>(v)); v = fmaxf(v, dpp_row<kDppXor1>(v));
    return v;
}
// 4-lane groups (xor 1, xor 2): quad permutes
__device__ __forceinline__ float group4_sum(float v) { v += dpp_row<kDppXor1>(v); v += dpp_row<kDppXor2>(v); return v; }
__device__ __forceinline__ float group4_max(float v) { v = fmaxf(v, dpp_row<kDppXor1>(v)); return fmaxf(v, dpp_row<kDppXor2>(v)); }

__device__ __forceinline__ float clamp_prob_u(float n) {
    return fminf(fmaxf(n, FLT_EPSILON), 1.0f - FLT_EPSILON);
}
__device__ __forceinline__ float softplus_u(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// ---- MFMA building blocks (v_mfma_f32_16x16x4_f32: A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15],
//      C/D col = lane&15, row = 4*(lane>>4)+reg).  The k order inside a 16-chunk is permuted the
//      same way on both operands (lane slot s carries k = 4s+j at step j), which leaves the sum intact.

// B fragments of one forward tile: fr[c] = W[o][16c + 4*slot .. +3], o = n0 + (lane&15); W row-major [*, H]
// Loads of data that ANOTHER CU of the same launch may have rewritten (the persistent update kernels: weights after
// the Adam phase, slabs, activations, the gradient bucket, statistics).  A CU's vector L1 is never refreshed by
// another CU's stores, so these loads must not be served by it.  NT = true: AGENT-SCOPE loads (`sc1`), the form the
// AMDGPU memory model defines for reading other CUs' data -- 4-byte values as relaxed agent-scope atomic loads, 16-byte
// values as two 8-byte ones (MI355X_MICROARCH.md, inter-workgroup visibility, "8-B agent atomics" / `sc1` loads): they
// miss in the L1 by definition and are served by the L2 all workers of a network share (one XCD), whatever other loads
// have touched the same lines before.  (Rounds 1-2 used non-temporal loads here: a cache-policy HINT that happened to
// bypass the L1 as long as every access to those lines was non-temporal; measured the same speed)  The producers' side is `s_waitcnt vmcnt(0)` + workgroup barrier before the flag store: their write-through
// stores have been acknowledged by that same L2.  NT = false: plain loads (separate launches: the kernel boundary does
// the invalidation).
template <bool NT> __device__ __forceinline__ float ld1(const float* p) {
    if (NT) return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return *p;
}
template <bool NT> __device__ __forceinline__ double ld1(const double* p) {
    if (NT) return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                                     __HIP_MEMORY_SCOPE_AGENT));
    return *p;
}
// 16 bytes in ONE instruction: `buffer_load_dwordx4 ... sc1` (two 8-byte agent-scope loads measured 8-10 % slower per
// mini-batch at C4).  A buffer resource needs a wave-uniform base: the first active lane's address minus 2 GiB, each lane
// then carries its own 32-bit byte offset (the lanes of a wave read one array: far less than 2 GiB apart).  The compiler
// tracks it with vmcnt like any other load.
typedef unsigned ppoaf_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ float4 ld4(const float* p) {
    if (NT) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(p);
        const unsigned lo0 = __builtin_amdgcn_readfirstlane((unsigned)a), hi0 = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        const unsigned long long base = (((unsigned long long)hi0 << 32) | lo0) - 0x80000000ull;
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(base), 0, 0xFFFFFFFF, 0x00020000);
        const ppoaf_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (unsigned)(a - base), 0, 16 /* sc1 */);
        return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
    return *reinterpret_cast<const float4*>(p);
}

template <int HT, bool NT = false>
__device__ __forceinline__ void load_fwd_frags(const float* __restrict__ W, int n0, int lane, float4 (&fr)[HT]) {
    const float* w = W + (long)(n0 + (lane & 15)) * (16 * HT) + 4 * (lane >> 4);
    // chunks c and c + 1 lie in one 128-byte line of every row: even chunks first -- a load that asks for a line whose fill is
    // still pending stalls the CU's L1 (round 4, stamps of the 256-wide row pairs: a 128 KB set arrived in 18 k cycles instead of
    // 32 k; C2 +0.7 %)
#pragma unroll
    for (int c = 0; c < HT; c += 2) fr[c] = ld4<NT>(w + 16 * c);
    __builtin_amdgcn_sched_barrier(0);                    // (the scheduler would sort the loads by offset again)
#pragma unroll
    for (int c = 1; c < HT; c += 2) fr[c] = ld4<NT>(w + 16 * c);
}
// B fragments of one dgrad tile: fr[c] = { W[16c+4*slot+j][n0 + (lane&15)] }_j
template <int HT, bool NT = false>
__device__ __forceinline__ void load_dgrad_frags(const float* __restrict__ W, int n0, int lane, float4 (&fr)[HT]) {
    constexpr int H = 16 * HT;
    const float* w = W + (long)(4 * (lane >> 4)) * H + n0 + (lane & 15);
#pragma unroll
    for (int c = 0; c < HT; ++c) {
        const float* wp = w + (long)(16 * c) * H;
        fr[c] = make_float4(ld1<NT>(wp), ld1<NT>(wp + H), ld1<NT>(wp + 2 * H), ld1<NT>(wp + 3 * H));
    }
}
// The same fragments through a buffer resource on the (wave-uniform) weight matrix: every load of a set shares ONE
// per-lane byte offset, the chunk / row displacement rides in the instruction's immediate or scalar offset -- no 64-bit
// address pair per load (the 64 scalar loads of a 256-wide dgrad set cost 128 address VGPRs as global loads: with them
// two sets cannot be in flight at once).  Forward chunks c and c + 1 share a 128-byte line of every row: even chunks
// first -- a load that asks for a line whose fill is still pending stalls the CU's L1 (stamps: a 128 KB set arrived in
// 18 k cycles instead of 32 k).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t frag_rsrc(const float* W) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ float4 frag_u4(const ppoaf_u32x4 v) {
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
template <int HT>
__device__ __forceinline__ void load_fwd_frags_buf(const float* __restrict__ W, int n0, int lane, float4 (&fr)[HT]) {
    const __amdgpu_buffer_rsrc_t rs = frag_rsrc(W);
    const unsigned off = (unsigned)(((n0 + (lane & 15)) * (16 * HT) + 4 * (lane >> 4)) * 4);
#pragma unroll
    for (int c = 0; c < HT; c += 2) fr[c] = frag_u4(__builtin_amdgcn_raw_buffer_load_b128(rs, off + 64u * c, 0, 0));
    __builtin_amdgcn_sched_barrier(0);                    // (the scheduler would sort the loads by offset again)
#pragma unroll
    for (int c = 1; c < HT; c += 2) fr[c] = frag_u4(__builtin_amdgcn_raw_buffer_load_b128(rs, off + 64u * c, 0, 0));
}
template <int HT>
__device__ __forceinline__ void load_dgrad_frags_buf(const float* __restrict__ W, int n0, int lane, float4 (&fr)[HT]) {
    constexpr int H = 16 * HT;
    const __amdgpu_buffer_rsrc_t rs = frag_rsrc(W);
    const unsigned off = (unsigned)((4 * (lane >> 4) * H + n0 + (lane & 15)) * 4);
#pragma unroll
    for (int c = 0; c < HT; ++c) {
        const int row = 16 * c * H * 4;
        fr[c] = make_float4(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, row, 0)),
                            __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, row + H * 4, 0)),
                            __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, row + 2 * H * 4, 0)),
                            __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, row + 3 * H * 4, 0)));
    }
}

// acc[16 rows, 16 cols] = init + A[16, H] . frags ; A rows in LDS with stride HS.  Two accumulators
// (even / odd chunks) keep the matrix pipe issuing back to back instead of waiting on its own result.
template <int HT>
__device__ __forceinline__ f32x4 mfma_rows_x_frags(const float* __restrict__ A, int HS, int lane,
                                                   const float4 (&fr)[HT], float init) {
    const float* arow = A + (lane & 15) * HS + 4 * (lane >> 4);
    f32x4 acc0 = {init, init, init, init};
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < HT; ++c) {
        const float4 a4 = *reinterpret_cast<const float4*>(arow + 16 * c);
        if (c & 1) {
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, fr[c].x, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, fr[c].y, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, fr[c].z, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, fr[c].w, acc1, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, fr[c].x, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, fr[c].y, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, fr[c].z, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, fr[c].w, acc0, 0, 0, 0);
        }
    }
    return acc0 + acc1;
}

// ---- forward fragments as WHOLE LINES.  One CU's L1 delivers the fragment pattern above (an instruction = 16 rows x 64
// bytes: sixteen half lines) at 14 B/clk whether the lines hit the L2 or not -- 128 KB in 9.1 k cycles -- and 8 rows x 128
// bytes per instruction at 32 (L2 miss) to 45 B/clk (tools/probes/frag_stream_probe.hip: 4.0 k / 2.9 k cycles).  So a set
// is REQUESTED line by line (raw[2 q + h] = 16 bytes of row n0 + 8 h + lane / 8, line q, block lane % 8) and turned into
// fragment order when it is consumed: each line of the 16 rows goes through a wave-private LDS slot (rows 144 bytes apart:
// conflict-free 16-byte reads), two slots so that line q + 1 is written while line q's MFMAs issue.  The MFMA operands and
// their order are exactly mfma_rows_x_frags' -- bitwise the same sums.
constexpr int kLineRow = 36;                              // floats between the rows of a slot (128 B + 16 B)
constexpr int kLineSlot = 16 * kLineRow;                  // floats per slot; a wave owns two
template <int HT>
__device__ __forceinline__ void load_fwd_lines_buf(const float* __restrict__ W, int n0, int lane, float4 (&raw)[HT]) {
    static_assert(HT % 2 == 0, "whole lines: H a multiple of 32");
    const __amdgpu_buffer_rsrc_t rs = frag_rsrc(W);
    const unsigned off = (unsigned)(((n0 + (lane >> 3)) * (16 * HT) + 4 * (lane & 7)) * 4);
#pragma unroll
    for (int q = 0; q < HT / 2; ++q) {
        raw[2 * q] = frag_u4(__builtin_amdgcn_raw_buffer_load_b128(rs, off + 128u * q, 0, 0));
        raw[2 * q + 1] = frag_u4(__builtin_amdgcn_raw_buffer_load_b128(rs, off + (unsigned)(8 * 16 * HT * 4) + 128u * q, 0, 0));
    }
}
template <int HT>
__device__ __forceinline__ f32x4 mfma_rows_x_lines(const float* __restrict__ A, int HS, int lane, const float4 (&raw)[HT],
                                                   float init, float* __restrict__ scratch) {
    constexpr int NL = HT / 2;                            // lines per row
    const float* arow = A + (lane & 15) * HS + 4 * (lane >> 4);
    float* wr = scratch + (lane >> 3) * kLineRow + 4 * (lane & 7);
    const float* rd = scratch + (lane & 15) * kLineRow + 4 * (lane >> 4);
    f32x4 acc0 = {init, init, init, init};
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
    // three stages in flight: line q + 2 being written, line q + 1 being read back, line q in the matrix pipe
    *reinterpret_cast<float4*>(wr) = raw[0];
    *reinterpret_cast<float4*>(wr + 8 * kLineRow) = raw[1];
    if (NL > 1) {
        *reinterpret_cast<float4*>(wr + kLineSlot) = raw[2];
        *reinterpret_cast<float4*>(wr + kLineSlot + 8 * kLineRow) = raw[3];
    }
    __builtin_amdgcn_wave_barrier();                      // (LDS executes a wave's instructions in order: no wait between the lanes' write and read)
    float4 f0 = *reinterpret_cast<const float4*>(rd);
    float4 f1 = *reinterpret_cast<const float4*>(rd + 16);
#pragma unroll
    for (int q = 0; q < NL; ++q) {
        const int slot = (q & 1) * kLineSlot;
        float4 g0 = f0, g1 = f1;
        if (q + 1 < NL) {
            g0 = *reinterpret_cast<const float4*>(rd + (kLineSlot - slot));
            g1 = *reinterpret_cast<const float4*>(rd + (kLineSlot - slot) + 16);
        }
        const float4 a0 = *reinterpret_cast<const float4*>(arow + 32 * q);
        const float4 a1 = *reinterpret_cast<const float4*>(arow + 32 * q + 16);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, f0.x, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, f0.y, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, f0.z, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, f0.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, f1.x, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, f1.y, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, f1.z, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, f1.w, acc1, 0, 0, 0);
        __builtin_amdgcn_wave_barrier();
        if (q + 2 < NL) {                                 // this line's slot is free: its fragments were read an iteration ago
            *reinterpret_cast<float4*>(wr + slot) = raw[2 * q + 4];
            *reinterpret_cast<float4*>(wr + slot + 8 * kLineRow) = raw[2 * q + 5];
        }
        f0 = g0; f1 = g1;
    }
    return acc0 + acc1;
}

// Weight gradient of one 16-row block of outputs: dW[m0+.., i] = sum_s D[s][m0+..] * Bsrc[s][i] for
// i < n_valid (n tiles of 16 columns), K = the 16 rows of the workgroup.  Both operands come from LDS.
__device__ __forceinline__ void wgrad_mtile(const float* __restrict__ D, int HS, const float* __restrict__ Bsrc,
                                            int strideB, int m0, int ntiles, int n_valid, int lane,
                                            float* __restrict__ dstW, int ldw) {
    float a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = D[(4 * (lane >> 4) + j) * HS + m0 + (lane & 15)];
    const float* bcol = Bsrc + (4 * (lane >> 4)) * strideB + (lane & 15);
    float* drow = dstW + (long)(m0 + 4 * (lane >> 4)) * ldw + (lane & 15);
#pragma unroll 4
    for (int nt = 0; nt < ntiles; ++nt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bcol[j * strideB + 16 * nt], acc, 0, 0, 0);
        if (16 * nt + (lane & 15) < n_valid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) drow[(long)r * ldw + 16 * nt] = acc[r];
        }
    }
}

// The same for a full [*, 16 * NT] gradient block with a compile-time tile count: all B operands are read
// from LDS up front and the NT accumulation chains (4 dependent MFMAs each) are issued interleaved, so the
// matrix pipe is never waiting on an LDS read or on its own previous result.
template <int NT>
__device__ __forceinline__ void wgrad_mtile_full(const float* __restrict__ D, int HS, const float* __restrict__ Bsrc,
                                                 int strideB, int m0, int lane, float* __restrict__ dstW, int ldw) {
    float a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = D[(4 * (lane >> 4) + j) * HS + m0 + (lane & 15)];
    const float* bcol = Bsrc + (4 * (lane >> 4)) * strideB + (lane & 15);
    float b[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) b[nt][j] = bcol[j * strideB + 16 * nt];
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[nt][j], acc[nt], 0, 0, 0);
    float* drow = dstW + (long)(m0 + 4 * (lane >> 4)) * ldw + (lane & 15);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) drow[(long)r * ldw + 16 * nt] = acc[nt][r];
}

constexpr int kNW = 8;                    // waves per workgroup of the fused update kernels
constexpr int kThreadsU = 64 * kNW;

// ---- fragment loads with an explicit row stride (W row-major [*, ldw], columns [0, 16*HT) of W) ----
template <int HT, bool ALIGNED>
__device__ __forceinline__ void load_fwd_frags_ld(const float* __restrict__ W, long ldw, int n0, int lane,
                                                  float4 (&fr)[HT]) {
    const float* w = W + (long)(n0 + (lane & 15)) * ldw + 4 * (lane >> 4);
    if (ALIGNED) {       // even chunks first: chunks c and c + 1 share a 128-byte line of every row (see load_fwd_frags)
#pragma unroll
        for (int c = 0; c < HT; c += 2) fr[c] = *reinterpret_cast<const float4*>(w + 16 * c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 1; c < HT; c += 2) fr[c] = *reinterpret_cast<const float4*>(w + 16 * c);
    } else {
#pragma unroll
        for (int c = 0; c < HT; ++c) fr[c] = make_float4(w[16 * c], w[16 * c + 1], w[16 * c + 2], w[16 * c + 3]);
    }
}
template <int HT>
__device__ __forceinline__ void load_dgrad_frags_ld(const float* __restrict__ W, long ldw, int n0, int lane,
                                                    float4 (&fr)[HT]) {
    const float* w = W + (long)(4 * (lane >> 4)) * ldw + n0 + (lane & 15);
#pragma unroll
    for (int c = 0; c < HT; ++c) {
        const float* wp = w + (long)(16 * c) * ldw;
        fr[c] = make_float4(wp[0], wp[ldw], wp[2 * ldw], wp[3 * ldw]);
    }
}

// ---- one output tile per wave, REQUESTED a phase ahead of its use (K14's kernels: waves >= HT hold no tile).  Forward
// sets travel as whole lines when the rows are whole lines (ldw a multiple of 32 floats), in fragment order otherwise;
// dgrad sets as buffer loads that share one per-lane offset.  The consumers are mfma_rows_x_lines / mfma_rows_x_frags:
// the sums of layer_fwd / layer_dgrad below, bit for bit.
template <int HT>
__device__ __forceinline__ void load_fwd_lines_ld(const float* __restrict__ W, long ldw, int n0, int lane, float4 (&raw)[HT]) {
    static_assert(HT % 2 == 0, "whole lines: H a multiple of 32");
    const __amdgpu_buffer_rsrc_t rs = frag_rsrc(W);
    const unsigned off = (unsigned)(((long)(n0 + (lane >> 3)) * ldw + 4 * (lane & 7)) * 4);
    const unsigned half = (unsigned)(8 * ldw * 4);
#pragma unroll
    for (int q = 0; q < HT / 2; ++q) {
        raw[2 * q] = frag_u4(__builtin_amdgcn_raw_buffer_load_b128(rs, off + 128u * q, 0, 0));
        raw[2 * q + 1] = frag_u4(__builtin_amdgcn_raw_buffer_load_b128(rs, off + half + 128u * q, 0, 0));
    }
}
template <int HT>
__device__ __forceinline__ void load_dgrad_frags_buf_ld(const float* __restrict__ W, long ldw, int n0, int lane, float4 (&fr)[HT]) {
    const __amdgpu_buffer_rsrc_t rs = frag_rsrc(W);
    const unsigned off = (unsigned)(((long)(4 * (lane >> 4)) * ldw + n0 + (lane & 15)) * 4);
    const int row = (int)(ldw * 4);
#pragma unroll
    for (int c = 0; c < HT; ++c) {
        fr[c] = make_float4(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, (16 * c) * row, 0)),
                            __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, (16 * c + 1) * row, 0)),
                            __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, (16 * c + 2) * row, 0)),
                            __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, (16 * c + 3) * row, 0)));
    }
}
// epilogues of layer_fwd / layer_dgrad for this wave's tile
template <int HT>
__device__ __forceinline__ void fwd_tile_store(const f32x4 acc, float* __restrict__ out, int act, int wave, int lane) {
    constexpr int HS = 16 * HT + 4;
    const int o = wave * 16 + (lane & 15);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(4 * (lane >> 4) + r) * HS + o] = act >= 0 ? act_fwd(acc[r], act) : acc[r];
}
template <int HT>
__device__ __forceinline__ void dgrad_tile_store(const f32x4 acc, const float* __restrict__ Hin, int act, float* __restrict__ dst_lds,
                                                 float* __restrict__ dst_glob, int wave, int lane) {
    constexpr int H = 16 * HT, HS = H + 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int s = 4 * (lane >> 4) + r, i = wave * 16 + (lane & 15);
        float v = acc[r];
        if (Hin) v *= act_bwd(Hin[s * HS + i], act);
        if (dst_lds) dst_lds[s * HS + i] = v;
        else dst_glob[(long)s * H + i] = v;
    }
}

// out[s][o] = f(bias[o] + sum_k A[s][k] W[o][k]) for the H outputs; act < 0: linear
template <int HT, bool ALIGNED, int NW = 8>
__device__ __forceinline__ void layer_fwd(const float* __restrict__ W, long ldw, const float* __restrict__ bias,
                                          const float* __restrict__ A, float* __restrict__ out, int act,
                                          int wave, int lane) {
    constexpr int HS = 16 * HT + 4;
    for (int nt = wave; nt < HT; nt += NW) {
        float4 fr[HT];
        load_fwd_frags_ld<HT, ALIGNED>(W, ldw, nt * 16, lane, fr);
        const int o = nt * 16 + (lane & 15);
        const f32x4 acc = mfma_rows_x_frags<HT>(A, HS, lane, fr, bias[o]);
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(4 * (lane >> 4) + r) * HS + o] = act >= 0 ? act_fwd(acc[r], act) : acc[r];
    }
}

// dh[s][i] = (sum_o D[s][o] W[o][i]) * act'(Hin[s][i]) for the H columns [0, H) of W (offset the pointer
// for other column blocks); result to LDS (dst_lds) or to global rows (dst_glob, row stride H).
template <int HT, int NW = 8>
__device__ __forceinline__ void layer_dgrad(const float* __restrict__ W, long ldw, const float* __restrict__ D,
                                            const float* __restrict__ Hin, int act, float* __restrict__ dst_lds,
                                            float* __restrict__ dst_glob, int wave, int lane) {
    constexpr int H = 16 * HT, HS = H + 4;
    for (int nt = wave; nt < HT; nt += NW) {
        float4 fr[HT];
        load_dgrad_frags_ld<HT>(W, ldw, nt * 16, lane, fr);
        const f32x4 acc = mfma_rows_x_frags<HT>(D, HS, lane, fr, 0.f);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = 4 * (lane >> 4) + r, i = nt * 16 + (lane & 15);
            float v = acc[r];
            if (Hin) v *= act_bwd(Hin[s * HS + i], act);
            if (dst_lds) dst_lds[s * HS + i] = v;
            else dst_glob[(long)s * H + i] = v;
        }
    }
}

// dW[o][i] = sum_s D[s][o] Bsrc[s][i] (i < n_valid) and, when dstB, db[o] = sum_s D[s][o]
template <int HT, int NW = 8>
__device__ __forceinline__ void layer_wgrad(const float* __restrict__ D, const float* __restrict__ Bsrc,
                                            int strideB, int ntiles, int n_valid, float* __restrict__ dstW,
                                            int ldw, float* __restrict__ dstB, int wave, int lane, int tid) {
    constexpr int H = 16 * HT, HS = H + 4;
    for (int mt = wave; mt < HT; mt += NW)
        wgrad_mtile(D, HS, Bsrc, strideB, mt * 16, ntiles, n_valid, lane, dstW, ldw);
    if (dstB) {
        for (int o = tid; o < H; o += 64 * NW) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < kRows; ++s) acc += D[s * HS + o];
            dstB[o] = acc;
        }
    }
}

// host: validate a network descriptor and fill the device view
int fill_net(const ppoaf_mlp_desc_t& d, NetDev& n, const char* what);

}  // namespace ppoaf
