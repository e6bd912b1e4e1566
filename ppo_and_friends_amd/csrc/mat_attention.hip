// K9: multi-agent-transformer attention core on f32 MFMA.
// Replaces the middle of SelfAttention.forward (networks/attention.py:94-103):
//   att = softmax((q @ k^T) / sqrt(d) [masked_fill(tril == 0, -inf)]);  y = att @ v
// for sequences of L agents (L = 3 in the SimpleSpread configs; L <= 16), head size D (64).
//
// A 3x3 score matrix is far below an MFMA tile, so floor(16 / L) sequences are packed
// block-diagonally into one 16-row tile: one wave computes S = Q_tile . K_tile^T (16x16, K = D) on
// v_mfma_f32_16x16x4_f32, masks every cross-sequence (and, if causal, future) entry, does the row
// softmax with 16-lane shuffles, transposes P through LDS and computes Y = P . V_tile.  Backward
// recomputes nothing: it reuses the saved probabilities (dV = P^T dY, dP = dY V^T,
// dS = P * (dP - rowsum(dP * P)) / sqrt(d), dQ = dS K, dK = dS^T Q), all as 16x16 MFMA tiles.
// f32 MFMA keeps float32 parity (exact products, fmaf chains).
#include "mlp_device.hpp"

namespace ppoaf {

constexpr int kAttWaves = 4;      // tiles per workgroup

struct AttDev {
    const float* q; const float* k; const float* v; long n_seq; int L, D, masked;
    float* y; float* probs;
    const float* dy; float* dq; float* dk; float* dv;
    int per_tile; long n_tiles;
};

// C-layout accumulator element (reg r of lane l) -> tile row / col
__device__ __forceinline__ int c_row(int lane, int r) { return 4 * (lane >> 4) + r; }
__device__ __forceinline__ int c_col(int lane) { return lane & 15; }

// acc[16,16] = X_tile[16, D] . Y_tile[16, D]^T  (both row-major with row stride D; rows >= n_rows read as 0)
__device__ __forceinline__ f32x4 tile_xyT(const float* __restrict__ X, const float* __restrict__ Y, int D,
                                          int n_rows, int lane) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, slot = lane >> 4;
    const bool ok = r < n_rows;
    const float* xr = X + (long)r * D + 4 * slot;
    const float* yr = Y + (long)r * D + 4 * slot;
    for (int c = 0; c < D / 16; ++c) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (ok) { a = *reinterpret_cast<const float4*>(xr + 16 * c); b = *reinterpret_cast<const float4*>(yr + 16 * c); }
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    }
    return acc;
}

// out_tile[16, D] (+)= A[16,16] . Z_tile[16, D]; A is read from an LDS tile (stride 17) either as is
// (A[i][k] = T[i][k]) or transposed (A[i][k] = T[k][i]); rows >= n_rows of Z read as 0.
__device__ __forceinline__ void tile_a_times_rows(const float* __restrict__ T, bool transposed,
                                                  const float* __restrict__ Z, int D, int n_rows, int lane,
                                                  float scale, float* __restrict__ out) {
    const int i = lane & 15, slot = lane >> 4;
    float a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kk = 4 * slot + j;
        a[j] = transposed ? T[kk * 17 + i] : T[i * 17 + kk];
    }
    for (int nt = 0; nt < D / 16; ++nt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int kk = 4 * slot + j;
            const float b = kk < n_rows ? Z[(long)kk * D + 16 * nt + i] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = c_row(lane, r);
            if (row < n_rows) out[(long)row * D + 16 * nt + i] = acc[r] * scale;
        }
    }
}

__global__ __launch_bounds__(64 * kAttWaves) void mat_attention_fwd_kernel(AttDev u) {
    __shared__ float sP[kAttWaves][16 * 17];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long tile = (long)blockIdx.x * kAttWaves + w;
    if (tile >= u.n_tiles) return;                        // wave-uniform; no block barrier below
    const int L = u.L, D = u.D;
    const long seq0 = tile * u.per_tile;
    const long rem = u.n_seq - seq0;
    const int n_rows = (int)((rem < u.per_tile ? rem : u.per_tile) * L);
    const float* Q = u.q + seq0 * L * D;
    const float* Kp = u.k + seq0 * L * D;
    const float* V = u.v + seq0 * L * D;
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);

    f32x4 s = tile_xyT(Q, Kp, D, n_rows, lane);
    const int col = c_col(lane);
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = c_row(lane, r);
        const bool in_block = row < n_rows && col < n_rows && (row / L) == (col / L);
        bool ok = in_block;
        if (u.masked) ok = ok && (col % L) <= (row % L);
        const float x = ok ? s[r] * inv_sqrt_d : -INFINITY;
        // row softmax over the 16 columns held by the 16 lanes of this slot group
        float m = x;
        m = fmaxf(m, __shfl_xor(m, 8, 64)); m = fmaxf(m, __shfl_xor(m, 4, 64));
        m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 1, 64));
        const float e = ok ? expf(x - m) : 0.f;
        const float den = group16_sum(e);
        p[r] = den > 0.f ? e / den : 0.f;
        sP[w][row * 17 + col] = p[r];
        if (in_block && u.probs)        // masked entries of the block are stored as 0
            u.probs[(seq0 + row / L) * L * L + (long)(row % L) * L + (col % L)] = p[r];
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0): this wave's LDS writes are done
    __builtin_amdgcn_wave_barrier();
    tile_a_times_rows(sP[w], false, V, D, n_rows, lane, 1.0f, u.y + seq0 * L * D);
}

__global__ __launch_bounds__(64 * kAttWaves) void mat_attention_bwd_kernel(AttDev u) {
    __shared__ float sP[kAttWaves][16 * 17];
    __shared__ float sS[kAttWaves][16 * 17];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long tile = (long)blockIdx.x * kAttWaves + w;
    if (tile >= u.n_tiles) return;
    const int L = u.L, D = u.D;
    const long seq0 = tile * u.per_tile;
    const long rem = u.n_seq - seq0;
    const int n_rows = (int)((rem < u.per_tile ? rem : u.per_tile) * L);
    const long off = seq0 * L * D;
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);
    const int col = c_col(lane);

    // P tile from the saved probabilities (zero outside the diagonal blocks)
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = c_row(lane, r);
        const bool ok = row < n_rows && col < n_rows && (row / L) == (col / L);
        p[r] = ok ? u.probs[(seq0 + row / L) * L * L + (long)(row % L) * L + (col % L)] : 0.f;
        sP[w][row * 17 + col] = p[r];
    }
    // dP = dY . V^T
    const f32x4 dp = tile_xyT(u.dy + off, u.v + off, D, n_rows, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = c_row(lane, r);
        const float dot = group16_sum(dp[r] * p[r]);               // sum over the row's columns
        sS[w][row * 17 + col] = p[r] * (dp[r] - dot) * inv_sqrt_d; // dS (already / sqrt(d))
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
    tile_a_times_rows(sP[w], true, u.dy + off, D, n_rows, lane, 1.0f, u.dv + off);   // dV = P^T dY
    tile_a_times_rows(sS[w], false, u.k + off, D, n_rows, lane, 1.0f, u.dq + off);   // dQ = dS K
    tile_a_times_rows(sS[w], true, u.q + off, D, n_rows, lane, 1.0f, u.dk + off);    // dK = dS^T Q
}

static int att_common(AttDev& u, const float* q, const float* k, const float* v, int64_t n_seq, int32_t L,
                      int32_t D, const char* what) {
    PPOAF_REQUIRE(n_seq >= 0, "%s: negative n_seq", what);
    PPOAF_REQUIRE(L >= 1 && L <= 16, "%s: L=%d out of [1,16]", what, L);
    PPOAF_REQUIRE(D >= 16 && D <= 1024 && D % 16 == 0, "%s: D=%d must be a multiple of 16", what, D);
    PPOAF_REQUIRE(q && k && v, "%s: null pointer", what);
    PPOAF_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0, "%s: q/k/v must be 16-byte aligned", what);
    u.q = q; u.k = k; u.v = v; u.n_seq = n_seq; u.L = L; u.D = D;
    u.per_tile = 16 / L;
    u.n_tiles = (n_seq + u.per_tile - 1) / u.per_tile;
    PPOAF_REQUIRE((u.n_tiles + kAttWaves - 1) / kAttWaves <= 0x7fffffffL, "%s: too many sequences", what);
    return PPOAF_OK;
}

}  // namespace ppoaf

using namespace ppoaf;

extern "C" int ppoaf_mat_attention_fwd(const float* q, const float* k, const float* v, int64_t n_seq,
                                       int32_t L, int32_t D, int masked, float* y_out, float* probs_out,
                                       ppoaf_stream_t stream) {
    AttDev u{};
    int rc = att_common(u, q, k, v, n_seq, L, D, "mat_attention_fwd");
    if (rc) return rc;
    if (n_seq == 0) return PPOAF_OK;
    PPOAF_REQUIRE(y_out, "mat_attention_fwd: null output");
    u.masked = masked; u.y = y_out; u.probs = probs_out;
    hipLaunchKernelGGL(mat_attention_fwd_kernel, dim3((unsigned)((u.n_tiles + kAttWaves - 1) / kAttWaves)),
                       dim3(64 * kAttWaves), 0, (hipStream_t)stream, u);
    return check_launch("mat_attention_fwd");
}

extern "C" int ppoaf_mat_attention_bwd(const float* q, const float* k, const float* v, const float* probs,
                                       const float* dy, int64_t n_seq, int32_t L, int32_t D, float* dq,
                                       float* dk, float* dv, ppoaf_stream_t stream) {
    AttDev u{};
    int rc = att_common(u, q, k, v, n_seq, L, D, "mat_attention_bwd");
    if (rc) return rc;
    if (n_seq == 0) return PPOAF_OK;
    PPOAF_REQUIRE(probs && dy && dq && dk && dv, "mat_attention_bwd: null pointer");
    PPOAF_REQUIRE(((uintptr_t)dy & 15) == 0, "mat_attention_bwd: dy must be 16-byte aligned");
    u.probs = const_cast<float*>(probs); u.dy = dy; u.dq = dq; u.dk = dk; u.dv = dv;
    hipLaunchKernelGGL(mat_attention_bwd_kernel, dim3((unsigned)((u.n_tiles + kAttWaves - 1) / kAttWaves)),
                       dim3(64 * kAttWaves), 0, (hipStream_t)stream, u);
    return check_launch("mat_attention_bwd");
}
