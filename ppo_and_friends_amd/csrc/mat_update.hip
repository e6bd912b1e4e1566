// K15: fused multi-agent-transformer mini-batch update (see include/ppoaf_hip.h for the contract).
// One iteration of PPO._ppo_batch_train for a MATPolicy (ppo.py:2292-2469 with
// mat_policy.py:378-439,628-699; networks multi_agent_transformer.py:22-373, attention.py:13-257) =
//   mat_update_fwd_bwd_kernel   gather + token block + critic (encoder) forward + actor (decoder)
//                               forward + categorical head / PPO loss / value loss + the whole
//                               backward -> weight-gradient slab of this workgroup
//   mat_update_reduce_kernel    slabs -> gradient bucket (fixed order), loss partials -> totals, cursor++
//   then K11 (global-norm clip + Adam over the one actor_critic bucket; mat_policy.py:677-699)
//
// Work decomposition: attention couples only the L agents of one env, so floor(16/L) sequences
// (15 token rows for L = 3) form a self-contained 16-row tile: one workgroup of 4 waves owns a
// tile end to end.  Every 64x64 linear is 4 output tiles of v_mfma_f32_16x16x4_f32 (one per wave),
// the 3x3 attention blocks are packed block-diagonally into one 16x16 MFMA tile exactly as K9 does,
// LayerNorm / GELU / residuals are 16-lane-per-row VALU passes, and everything the backward needs
// (normalised activations, pre-GELU values, q/k/v, probabilities) stays in LDS: ~147 KB of the 160 KB.
// A mini-batch of 256 envs is 52 workgroups; the ~80 dependent phases are barrier + LDS latency
// bound, not MFMA bound -- the point is that they replace ~270 separate kernel launches.
#include "mlp_device.hpp"
#include "tail_sync.hpp"
#include <hip/hip_ext.h>
#include <cstdlib>

namespace ppoaf {

constexpr int kMD = 64;            // embedding width
constexpr int kMHS = kMD + 4;      // LDS row stride of a [16, 64] tile
constexpr int kMNW = 4;            // waves per workgroup
constexpr int kMT = 64 * kMNW;
constexpr int kMXS = 20;           // row stride of the action-token tile [16, 16 + 4]
constexpr int kMTile = kRows * kMHS;

// parameter tensors of MATActorCritic in module order (== bucket order); offsets in floats
enum MatP {
    A_ENC_W, A_LN_G, A_LN_B, A_LN1_G, A_LN1_B, A_LN2_G, A_LN2_B, A_LN3_G, A_LN3_B,
    A_K1_W, A_K1_B, A_Q1_W, A_Q1_B, A_V1_W, A_V1_B, A_P1_W, A_P1_B,
    A_K2_W, A_K2_B, A_Q2_W, A_Q2_B, A_V2_W, A_V2_B, A_P2_W, A_P2_B,
    A_M1_W, A_M1_B, A_M2_W, A_M2_B, A_H1_W, A_H1_B, A_HLN_G, A_HLN_B, A_H2_W, A_H2_B,
    C_OLN_G, C_OLN_B, C_ENC_W, C_ENC_B, C_LN_G, C_LN_B, C_LN1_G, C_LN1_B, C_LN2_G, C_LN2_B,
    C_K_W, C_K_B, C_Q_W, C_Q_B, C_V_W, C_V_B, C_P_W, C_P_B,
    C_M1_W, C_M1_B, C_M2_W, C_M2_B, C_H1_W, C_H1_B, C_HLN_G, C_HLN_B, C_H2_W, C_H2_B,
    MAT_NUM_PARAMS
};
static_assert(MAT_NUM_PARAMS == 63, "offset table layout");

struct MatDev {
    long off[64];
    int O, L, NA, Ain, per_tile, nT;
    long total;
    const float* params; float* grads; float* slabs;
    const float* obs; const int64_t* actions; const float* adv; const float* old_lp; const float* rtg; float* values;
    const int64_t* perm; const int32_t* row_map; long n_rows;
    int64_t* cursor; long B, batch_stride, mb_offset, cursor_advance;
    int normalize_values, n_ranks, normalize_adv, use_huber;
    float* vn_mean; float* vn_var; double* vn_count; const double* vn_records; const double* adv_records;
    float surr_clip, entropy_weight, kl_loss_weight, huber_delta;
    float* loss_partials; double* totals;
    double* norm_scratch; int64_t* step_count; int fuse_norm;
    int pregathered;     // obs / actions / adv / old_lp / rtg are per-epoch tables in shuffled order (entry i belongs to perm[i])
    // split-wgrad chain (args->split_workspace): fwd_bwd computes NO weight gradient of the 18 64x64 linears; it publishes
    // their inputs and dLoss/dz tiles into panels and the reduce launch becomes mat_update_wgrad_kernel
    int split;
    int R;               // rows of a panel = 16 * nT (tile g owns rows [16 g, +16), dead rows included: their dz is zero)
    float* xpanel;       // [MAT_NUM_XPANELS][R][64] inputs of the linears
    float* dpanel;       // [kMatLin][R][64]         dLoss / d(linear output)
    int goff[64];        // where parameter k's partial sits inside a workgroup's slab (split: compact layout of the small
                         // tensors, -1 for the 18 linears' W and b; else == off)
    long slab_stride;    // floats per workgroup slab (split: n_small4 * 4; else total)
    int n_small4;        // float4 columns of the compact slab
    int n_seg, seg_start[4], seg_dst[4];   // runs of small tensors: compact float offset -> bucket float offset
};

// the 18 linears whose weight gradients the wgrad launch forms, in the order the backward meets them, and the panel that
// holds each one's input (x1 feeds K2 / V2, x0 feeds Q1 / K1 / V1, H0 feeds Q / K / V, rep_enc feeds Q2 and the critic head)
enum MatX { X_X3, X_GAM, X_X2, X_AY2, X_ENC, X_X1, X_AY1, X_X0, X_GCM, X_H1, X_CY, X_H0, MAT_NUM_XPANELS };
constexpr int kMatLin = 18;
#define PPOAF_MAT_LIN_W { A_H1_W, A_M2_W, A_M1_W, A_P2_W, A_Q2_W, A_K2_W, A_V2_W, A_P1_W, A_Q1_W, A_K1_W, A_V1_W, \
                          C_H1_W, C_M2_W, C_M1_W, C_P_W, C_Q_W, C_K_W, C_V_W }
#define PPOAF_MAT_LIN_X { X_X3, X_GAM, X_X2, X_AY2, X_ENC, X_X1, X_X1, X_AY1, X_X0, X_X0, X_X0, \
                          X_ENC, X_GCM, X_H1, X_CY, X_H0, X_H0, X_H0 }
static const int kMatLinW_host[kMatLin] = PPOAF_MAT_LIN_W;
__constant__ int kMatLinW[kMatLin] = PPOAF_MAT_LIN_W;
__constant__ int kMatLinX[kMatLin] = PPOAF_MAT_LIN_X;
// indices into the tables above, named by the weight (call sites of the backward)
enum MatLinId { L_A_H1, L_A_M2, L_A_M1, L_A_P2, L_A_Q2, L_A_K2, L_A_V2, L_A_P1, L_A_Q1, L_A_K1, L_A_V1,
                L_C_H1, L_C_M2, L_C_M1, L_C_P, L_C_Q, L_C_K, L_C_V };

extern __shared__ __attribute__((aligned(16))) unsigned char mat_smem[];

// Diagnostic build only (-DPPOAF_MAT_STAMPS, tools/mat_stamps.py): s_memtime of workgroup 0 / thread 0 at the marked points
// of the update kernel, into a buffer nothing else reads.  The shipped library executes no stamp.
#ifdef PPOAF_MAT_STAMPS
static __device__ unsigned long long g_mat_stamps[64];
#define MAT_STAMP(k)                                                                     \
    do {                                                                                 \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                       \
            unsigned long long t_;                                                       \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");    \
            g_mat_stamps[k] = t_;                                                        \
        }                                                                                \
    } while (0)
#else
#define MAT_STAMP(k) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------
// element-wise passes over a [16, 64] tile: thread = (row = tid >> 4, lane16 = tid & 15), 4 columns each
// ------------------------------------------------------------------------------------------------
// Exact (erf) GELU, nn.GELU's default (multi_agent_transformer.py:51-60).  erf by Abramowitz & Stegun 7.1.26 --
// erf(x) = 1 - (a1 t + .. + a5 t^5) exp(-x^2), t = 1 / (1 + p x), |error| <= 1.5e-7 -- instead of the library erff:
// one exponential (the SAME exp(-z^2 / 2) the derivative's density term needs), one reciprocal and five FMAs against
// ~85 instructions per element; a GELU pass over a tile was 1400 of a phase's 1500 cycles (tools/mat_stamps.py) and 16 of
// the kernel's 71 phases contain one.  1.5e-7 absolute on erf is 25 x below the 1e-5 parity tolerance of every output.
__device__ __forceinline__ void gelu_parts(float z, float& cdf, float& pdf) {
    const float x = fabsf(z) * 0.70710678118654752440f;
    const float e = __expf(-x * x);                            // = exp(-z^2 / 2)
    const float t = __frcp_rn(fmaf(0.3275911f, x, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float erf_abs = 1.0f - poly * t * e;
    cdf = 0.5f * (1.0f + copysignf(erf_abs, z));
    pdf = e * 0.39894228040143267794f;
}
__device__ __forceinline__ float gelu_f(float z) { float cdf, pdf; gelu_parts(z, cdf, pdf); return z * cdf; }
__device__ __forceinline__ float gelu_d(float z) { float cdf, pdf; gelu_parts(z, cdf, pdf); return cdf + z * pdf; }

// out = gelu(Z)
__device__ __forceinline__ void tile_gelu(const float* __restrict__ Z, float* __restrict__ out, int tid) {
    const int row = tid >> 4, l = tid & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) out[row * kMHS + l + 16 * j] = gelu_f(Z[row * kMHS + l + 16 * j]);
}
// D *= gelu'(Z)
__device__ __forceinline__ void tile_gelu_bwd(float* __restrict__ D, const float* __restrict__ Z, int tid) {
    const int row = tid >> 4, l = tid & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) D[row * kMHS + l + 16 * j] *= gelu_d(Z[row * kMHS + l + 16 * j]);
}
// out = A + B
__device__ __forceinline__ void tile_add(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out, int tid) {
    const int row = tid >> 4, l = tid & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) out[row * kMHS + l + 16 * j] = A[row * kMHS + l + 16 * j] + B[row * kMHS + l + 16 * j];
}
// Y = Xhat * g + b   (LayerNorm output re-materialised for the backward)
__device__ __forceinline__ void tile_affine(const float* __restrict__ Xhat, const float* __restrict__ g,
                                            const float* __restrict__ b, float* __restrict__ Y, int tid) {
    const int row = tid >> 4, l = tid & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = l + 16 * j; Y[row * kMHS + c] = Xhat[row * kMHS + c] * g[c] + b[c]; }
}

// LayerNorm over W <= 64 columns (tile stride `ld`): Xhat, rstd[16], Y = Xhat * g + b.  X may alias Y.
__device__ __forceinline__ void tile_ln_fwd(const float* X, int ld, int W, const float* __restrict__ g,
                                            const float* __restrict__ b, float* __restrict__ Xhat,
                                            float* __restrict__ rstd, float* Y, int tid) {
    const int row = tid >> 4, l = tid & 15;
    float x[4], s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = l + 16 * j; x[j] = c < W ? X[row * ld + c] : 0.f; s += x[j]; }
    const float mean = group16_sum(s) / (float)W;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = l + 16 * j; x[j] = c < W ? x[j] - mean : 0.f; q += x[j] * x[j]; }
    const float rs = 1.0f / sqrtf(group16_sum(q) / (float)W + 1e-5f);
    if (l == 0) rstd[row] = rs;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = l + 16 * j;
        if (c < W) {
            const float xh = x[j] * rs;
            Xhat[row * ld + c] = xh;
            Y[row * ld + c] = xh * g[c] + b[c];
        }
    }
}
// LayerNorm backward: DX = rstd (g dy - mean(g dy) - xhat mean(g dy xhat)); d gamma / d beta column sums -> slab.
// DX must not alias DY (the column sums read DY while the rows are being written).  DX == nullptr: no input gradient.
__device__ __forceinline__ void tile_ln_bwd(const float* __restrict__ DY, int ld, int W, const float* __restrict__ Xhat,
                                            const float* __restrict__ rstd, const float* __restrict__ g,
                                            float* __restrict__ DX, float* __restrict__ slab_g, float* __restrict__ slab_b,
                                            int tid) {
    const int row = tid >> 4, l = tid & 15;
    if (DX) {
        float dxh[4], xh[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = l + 16 * j;
            xh[j] = c < W ? Xhat[row * ld + c] : 0.f;
            dxh[j] = c < W ? DY[row * ld + c] * g[c] : 0.f;
            s1 += dxh[j]; s2 += dxh[j] * xh[j];
        }
        const float m1 = group16_sum(s1) / (float)W, m2 = group16_sum(s2) / (float)W, rs = rstd[row];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int c = l + 16 * j; if (c < W) DX[row * ld + c] = (dxh[j] - m1 - xh[j] * m2) * rs; }
    }
    if (tid < 64) {
        if (tid < W) {
            float a = 0.f;
#pragma unroll
            for (int s = 0; s < kRows; ++s) a = fmaf(DY[s * ld + tid], Xhat[s * ld + tid], a);
            slab_g[tid] = a;
        }
    } else if (tid < 128) {
        const int c = tid - 64;
        if (c < W) {
            float a = 0.f;
#pragma unroll
            for (int s = 0; s < kRows; ++s) a += DY[s * ld + c];
            slab_b[c] = a;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// linears on [16, 64] tiles (one output tile of 16 columns per wave)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lin_fwd(const float* __restrict__ W, const float* __restrict__ bias,
                                        const float* __restrict__ A, float* __restrict__ out, int wave, int lane) {
    layer_fwd<4, true, kMNW>(W, kMD, bias, A, out, -1, wave, lane);
}
__device__ __forceinline__ void lin_wgrad(const float* __restrict__ Dt, const float* __restrict__ In,
                                          float* __restrict__ dW, float* __restrict__ dB, int wave, int lane, int tid) {
    layer_wgrad<4, kMNW>(Dt, In, kMHS, 4, kMD, dW, kMD, dB, wave, lane, tid);
}
// out (+)= Dt . W   (out must not alias Dt)
template <bool ACCUM>
__device__ __forceinline__ void lin_dgrad(const float* __restrict__ W, const float* __restrict__ Dt,
                                          float* __restrict__ out, int wave, int lane) {
    float4 fr[4];
    load_dgrad_frags_ld<4>(W, kMD, wave * 16, lane, fr);
    const f32x4 acc = mfma_rows_x_frags<4>(Dt, kMHS, lane, fr, 0.f);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int idx = (4 * (lane >> 4) + r) * kMHS + wave * 16 + (lane & 15);
        out[idx] = ACCUM ? out[idx] + acc[r] : acc[r];
    }
}

// split-wgrad chain: a settled [16, 64] LDS tile -> rows [16 g, +16) of a [R][64] panel (one float4 per thread)
__device__ __forceinline__ void publish_tile(const float* __restrict__ T, float* __restrict__ panel, int g, int tid) {
    const int r = tid >> 4, c4 = tid & 15;
    *reinterpret_cast<float4*>(panel + ((long)g * kRows + r) * kMD + 4 * c4) =
        *reinterpret_cast<const float4*>(T + r * kMHS + 4 * c4);
}

// The same linears with their weight fragments REQUESTED AHEAD: a wave's share of a 64x64 linear is 4 float4 per lane (+ its
// bias element).  A phase of this kernel is a dependent chain -- barrier, weight fragments from the L2 (~400 cycles), A
// from LDS, 16 MFMAs (32 cycles each), LDS stores, barrier: ~1800 cycles by in-kernel stamps, tools/mat_stamps.py -- and
// the 400 cycles of fragment latency sit at the head of 45 of the 71 phases.  The update kernel therefore requests the
// fragments of the NEXT linear at the start of the phase that runs the current one (they land behind its MFMAs and the
// barrier).  Not deeper: vmcnt retires in order, so with a whole pass's fragments in flight every small load issued
// after them (LayerNorm gains, biases) waited for all of them -- measured slower than no prefetch at all.
struct MatFr { float4 f[4]; float b; };
__device__ __forceinline__ void pf_fwd(const float* __restrict__ W, const float* __restrict__ bias, int wave, int lane, MatFr& r) {
    load_fwd_frags_ld<4, true>(W, kMD, wave * 16, lane, r.f);
    r.b = bias[wave * 16 + (lane & 15)];
}
__device__ __forceinline__ void pf_dgrad(const float* __restrict__ W, int wave, int lane, MatFr& r) {
    load_dgrad_frags_ld<4>(W, kMD, wave * 16, lane, r.f);
    r.b = 0.f;
}
__device__ __forceinline__ void lin_fwd_r(const MatFr& r, const float* __restrict__ A, float* __restrict__ out, int wave, int lane) {
    const int o = wave * 16 + (lane & 15);
    const f32x4 acc = mfma_rows_x_frags<4>(A, kMHS, lane, r.f, r.b);
#pragma unroll
    for (int q = 0; q < 4; ++q) out[(4 * (lane >> 4) + q) * kMHS + o] = acc[q];
}
template <bool ACCUM>
__device__ __forceinline__ void lin_dgrad_r(const MatFr& r, const float* __restrict__ Dt, float* __restrict__ out, int wave, int lane) {
    const f32x4 acc = mfma_rows_x_frags<4>(Dt, kMHS, lane, r.f, 0.f);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = (4 * (lane >> 4) + q) * kMHS + wave * 16 + (lane & 15);
        out[idx] = ACCUM ? out[idx] + acc[q] : acc[q];
    }
}

// ------------------------------------------------------------------------------------------------
// attention core on LDS tiles (K9's block-diagonal packing).  Q, K, V, Y, dY ... are [16, 64] tiles.
// ------------------------------------------------------------------------------------------------
// out[:, 16 nt ..] = A[16,16] . Z[:, 16 nt ..]; A from the 17-stride tile T, as is or transposed
__device__ __forceinline__ void att_a_times(const float* __restrict__ T, bool transposed, const float* __restrict__ Z,
                                            float* __restrict__ out, int nt, int lane) {
    const int i = lane & 15, slot = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int kk = 4 * slot + j;
        const float a = transposed ? T[kk * 17 + i] : T[i * 17 + kk];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Z[kk * kMHS + 16 * nt + i], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(4 * slot + r) * kMHS + 16 * nt + i] = acc[r];
}
// One 16-wide K chunk of X Y^T (X, Y: [16, 64] tiles): the wave's share of a score tile
__device__ __forceinline__ f32x4 att_xyT_chunk(const float* __restrict__ X, const float* __restrict__ Y, int chunk, int lane) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float4 a = *reinterpret_cast<const float4*>(X + (lane & 15) * kMHS + 16 * chunk + 4 * (lane >> 4));
    const float4 b = *reinterpret_cast<const float4*>(Y + (lane & 15) * kMHS + 16 * chunk + 4 * (lane >> 4));
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    return acc;
}
// forward: probabilities -> sP (kept for the backward), Y = P V.  Three barriers inside, none after.
// The 16 x 16 score tile is ONE MFMA output tile: instead of one wave running its 16 dependent MFMAs and the softmax of all
// 16 rows while three waves wait (stamps: 3.6 k cycles per attention), every wave takes one 16-wide K chunk (4 MFMAs), the
// four partial tiles meet in LDS (the Y tile is free until the last phase), and every wave runs the softmax of 4 rows, one
// row per 16 lanes.
__device__ __forceinline__ void att_fwd(const float* Q, const float* K, const float* V, float* Y, float* sP,
                                        int L, int n_rows, bool masked, int wave, int lane) {
    {
        const f32x4 s = att_xyT_chunk(Q, K, wave, lane);
        const int col = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[wave * 272 + (4 * (lane >> 4) + r) * 17 + col] = s[r];
    }
    __syncthreads();
    {
        const int row = 4 * wave + (lane >> 4), col = lane & 15;
        const float sc = (Y[row * 17 + col] + Y[272 + row * 17 + col]) + (Y[544 + row * 17 + col] + Y[816 + row * 17 + col]);
        bool ok = row < n_rows && col < n_rows && (row / L) == (col / L);
        if (masked) ok = ok && (col % L) <= (row % L);
        const float x = ok ? sc * 0.125f : -INFINITY;                       // 1 / sqrt(64)
        const float m = group16_max(x);
        const float e = ok ? expf(x - m) : 0.f;
        const float den = group16_sum(e);
        sP[row * 17 + col] = den > 0.f ? e / den : 0.f;
    }
    __syncthreads();
    att_a_times(sP, false, V, Y, wave, lane);
}
// backward from dY: dQ, dK, dV (three distinct output tiles); sS is a 17-stride scratch tile; the partial tiles of
// dP = dY V^T meet in the dQ tile (written only in the last phase)
__device__ __forceinline__ void att_bwd(const float* Q, const float* K, const float* V, const float* sP, const float* dY,
                                        float* dQ, float* dK, float* dV, float* sS, int wave, int lane) {
    {
        const f32x4 dp = att_xyT_chunk(dY, V, wave, lane);
        const int col = lane & 15;
#pragma unroll
        for (int r = 0; r < 4; ++r) dQ[wave * 272 + (4 * (lane >> 4) + r) * 17 + col] = dp[r];
    }
    __syncthreads();
    {
        const int row = 4 * wave + (lane >> 4), col = lane & 15;
        const float dp = (dQ[row * 17 + col] + dQ[272 + row * 17 + col]) + (dQ[544 + row * 17 + col] + dQ[816 + row * 17 + col]);
        const float p = sP[row * 17 + col];
        const float dot = group16_sum(dp * p);
        sS[row * 17 + col] = p * (dp - dot) * 0.125f;
    }
    __syncthreads();
    att_a_times(sP, true, dY, dV, wave, lane);      // dV = P^T dY
    att_a_times(sS, false, K, dQ, wave, lane);      // dQ = dS K
    att_a_times(sS, true, Q, dK, wave, lane);       // dK = dS^T Q
}

// ------------------------------------------------------------------------------------------------
// narrow first layers (observation encoder K = O, action encoder K = Ain): weights [64, K] row-major
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void narrow_fwd(const float* __restrict__ W, int Kdim, const float* __restrict__ bias,
                                           const float* __restrict__ X, int ldx, float* __restrict__ out, int wave, int lane) {
    const int o = wave * 16 + (lane & 15);
    const float bv = bias ? bias[o] : 0.f;
    f32x4 acc = {bv, bv, bv, bv};
    const float* w = W + (long)o * Kdim;
    const float* arow = X + (lane & 15) * ldx;
    for (int k0 = 0; k0 < Kdim; k0 += 16) {
        float bq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int k = k0 + 4 * j + (lane >> 4); bq[j] = k < Kdim ? w[k] : 0.f; }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[k0 + 4 * j + (lane >> 4)], bq[j], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(4 * (lane >> 4) + r) * kMHS + o] = acc[r];
}
// dX[16, Kdim] = D[16,64] . W[64, Kdim]  (tile stride ldx; columns >= Kdim untouched)
__device__ __forceinline__ void narrow_dgrad(const float* __restrict__ W, int Kdim, const float* __restrict__ Dt,
                                             float* __restrict__ dX, int ldx, int wave, int lane) {
    for (int nt = wave; nt * 16 < Kdim; nt += kMNW) {
        const int i = nt * 16 + (lane & 15);
        float4 fr[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float* wp = W + (long)(16 * c + 4 * (lane >> 4)) * Kdim + i;
            fr[c] = i < Kdim ? make_float4(wp[0], wp[Kdim], wp[2 * Kdim], wp[3 * Kdim]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const f32x4 acc = mfma_rows_x_frags<4>(Dt, kMHS, lane, fr, 0.f);
        if (i < Kdim) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dX[(4 * (lane >> 4) + r) * ldx + i] = acc[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// narrow output layers (64 -> n_out <= 8) on the VALU
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void head_out_fwd(const float* __restrict__ W, const float* __restrict__ bias, int n_out,
                                             const float* __restrict__ Hh, float* __restrict__ sOut, int tid) {
    const int s = tid >> 4, part = tid & 15;
    for (int k = 0; k < n_out; ++k) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = fmaf(Hh[s * kMHS + part + 16 * i], W[k * kMD + part + 16 * i], acc);
        acc = group16_sum(acc);
        if (part == 0) sOut[s * 8 + k] = acc + bias[k];
    }
}
// dW[k][i], db[k] -> slab; dH[s][i] = sum_k dOut[s][k] W[k][i]
__device__ __forceinline__ void head_out_bwd(const float* __restrict__ W, int n_out, const float* __restrict__ Hh,
                                             const float* __restrict__ sDOut, float* __restrict__ dH,
                                             float* __restrict__ slabW, float* __restrict__ slabB, int tid) {
    if (tid < kMD) {
        float h[kRows];
#pragma unroll
        for (int s = 0; s < kRows; ++s) h[s] = Hh[s * kMHS + tid];
        for (int k = 0; k < n_out; ++k) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < kRows; ++s) acc = fmaf(sDOut[s * 8 + k], h[s], acc);
            slabW[k * kMD + tid] = acc;
        }
    } else if (tid < kMD + 8) {
        const int k = tid - kMD;
        float acc = 0.f;
        if (k < n_out)
#pragma unroll
            for (int s = 0; s < kRows; ++s) acc += sDOut[s * 8 + k];
        if (k < ((n_out + 3) & ~3)) slabB[k] = acc;
    }
    const int s = tid >> 4, part = tid & 15;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < n_out; ++k) {
        const float d = sDOut[s * 8 + k];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = fmaf(d, W[k * kMD + part + 16 * i], acc[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) dH[s * kMHS + part + 16 * i] = acc[i];
}

// Barriers separate phases whose readers use a different thread <-> element mapping than the writers (MFMA
// tiles vs the 16-lanes-per-row passes).  Row-pass -> row-pass sequences (add -> LayerNorm, GELU -> LayerNorm,
// LayerNorm backward -> GELU backward) and accumulating dgrads into one tile touch only a thread's own elements
// and run without one.
#define MAT_SYNC() __syncthreads()

// LDS carve shared by the update kernel (K15) and the rollout step kernel (K16)
struct MatCtx {
    const float* P; const long* off;
    int O, L, NA, Ain, OS, n_rows;
    int* sRow; int* sAct; float* sMisc; float* sRowF; float* sRstd;
    float *sOutC, *sOutA, *sDOutC, *sDOutA, *sP0, *sP1, *sP2, *sS, *sXA, *sXO, *sYO, *T;
    __device__ __forceinline__ float* tile(int k) const { return T + (long)k * kMTile; }
    __device__ __forceinline__ float* rstd(int k) const { return sRstd + 16 * k; }
    __device__ __forceinline__ const float* W(int k) const { return P + off[k]; }
    __device__ __forceinline__ void carve(float* sm, int O_) {
        O = O_; OS = 16 * ((O_ + 15) >> 4) + 4;
        sRow = reinterpret_cast<int*>(sm);                 // [16] buffer row of the token's env (-1: padding)
        sAct = reinterpret_cast<int*>(sm) + 16;            // [16] action of the token
        sMisc = sm + 32;                                   // [8] adv mean/std, vn mean/var
        sRowF = sm + 48;                                   // [3][16] adv, old log-prob, rewards-to-go
        sRstd = sm + 96;                                   // [10][16]
        sOutC = sm + 256;                                  // [16][8] critic output (col 0)
        sOutA = sOutC + 128;                               // [16][8] logits
        sDOutC = sOutA + 128;
        sDOutA = sDOutC + 128;
        sP0 = sDOutA + 128;                                // 3 probability tiles + 1 dS scratch, [16][17] each
        sP1 = sP0 + 272;
        sP2 = sP1 + 272;
        sS = sP2 + 272;
        sXA = sS + 272;                                    // [16][kMXS] action tokens
        sXO = sXA + kRows * kMXS;                          // [16][OS] xhat of the observation LayerNorm
        sYO = sXO + kRows * OS;                            // [16][OS] its output (input of the encoder linear)
        T = sYO + kRows * OS;                              // [16][68] tiles from here on
    }
};
// tile indices: saved activations of the critic (0-11) and the actor (12-27), scratch (28-33)
enum { cZ1_ = 0, cN0_, cQ_, cK_, cV_, cY_, cN1_, cZM_, cN2_, cENC_, cZH_, cNH_,
       aZ_, aN0_, aQ1_, aK1_, aV1_, aY1_, aN1_, aK2_, aV2_, aQ2_, aY2_, aN2_, aZM_, aN3_, aZH_, aNH_,
       S0_, S1_, S2_, S3_, S4_, DENC_, MAT_NUM_TILES };

#define MAT_TILES(c)                                                                                              \
    float *cZ1 = c.tile(cZ1_), *cN0 = c.tile(cN0_), *cQ = c.tile(cQ_), *cK = c.tile(cK_), *cV = c.tile(cV_),       \
          *cY = c.tile(cY_), *cN1 = c.tile(cN1_), *cZM = c.tile(cZM_), *cN2 = c.tile(cN2_), *cENC = c.tile(cENC_), \
          *cZH = c.tile(cZH_), *cNH = c.tile(cNH_);                                                               \
    float *aZ = c.tile(aZ_), *aN0 = c.tile(aN0_), *aQ1 = c.tile(aQ1_), *aK1 = c.tile(aK1_), *aV1 = c.tile(aV1_),   \
          *aY1 = c.tile(aY1_), *aN1 = c.tile(aN1_), *aK2 = c.tile(aK2_), *aV2 = c.tile(aV2_), *aQ2 = c.tile(aQ2_), \
          *aY2 = c.tile(aY2_), *aN2 = c.tile(aN2_), *aZM = c.tile(aZM_), *aN3 = c.tile(aN3_), *aZH = c.tile(aZH_), \
          *aNH = c.tile(aNH_);                                                                                    \
    float *S0 = c.tile(S0_), *S1 = c.tile(S1_), *S2 = c.tile(S2_), *S3 = c.tile(S3_), *S4 = c.tile(S4_),           \
          *DENC = c.tile(DENC_);                                                                                  \
    (void)cZ1; (void)cN0; (void)cQ; (void)cK; (void)cV; (void)cY; (void)cN1; (void)cZM; (void)cN2; (void)cENC;     \
    (void)cZH; (void)cNH; (void)aZ; (void)aN0; (void)aQ1; (void)aK1; (void)aV1; (void)aY1; (void)aN1; (void)aK2;   \
    (void)aV2; (void)aQ2; (void)aY2; (void)aN2; (void)aZM; (void)aN3; (void)aZH; (void)aNH; (void)S0; (void)S1;   \
    (void)S2; (void)S3; (void)S4; (void)DENC;                                                                     \
    float *sXA = c.sXA, *sXO = c.sXO, *sYO = c.sYO, *sP0 = c.sP0, *sP1 = c.sP1, *sP2 = c.sP2, *sS = c.sS,          \
          *sOutC = c.sOutC, *sOutA = c.sOutA;                                                                     \
    (void)sXA; (void)sXO; (void)sYO; (void)sP0; (void)sP1; (void)sP2; (void)sS; (void)sOutC; (void)sOutA;          \
    const int O = c.O, L = c.L, NA = c.NA, Ain = c.Ain, OS = c.OS, n_rows = c.n_rows;                              \
    (void)O; (void)L; (void)NA; (void)Ain; (void)OS; (void)n_rows;                                                \
    auto W = [&](int k) -> const float* { return c.W(k); };                                                       \
    auto rstd = [&](int k) -> float* { return c.rstd(k); };                                                       \
    (void)W; (void)rstd

// split-wgrad chain: the forward publishes every linear's input tile the moment it is settled (the backward then has
// no reason to rebuild it from the saved normalised activations); K16 and the slab form publish nothing
struct MatNoPub { __device__ __forceinline__ void operator()(int, const float*) const {} };
struct MatPanelPub {
    float* xpanel; long plane; int g, tid;
    __device__ __forceinline__ void operator()(int xid, const float* T) const { publish_tile(T, xpanel + (long)xid * plane, g, tid); }
};

// critic (encoder): observations in sYO -> rep_enc in cENC, value in sOutC[:, 0]; ends WITHOUT a barrier
// PF (K15): the fragments of linear `n` are requested one phase ahead -- MAT_NEXT(n) at the start of the phase before;
// else (K16: its registers are spoken for) at the point of use
#define MAT_LIN(r, Wk, Bk, A, out) do { if (!PF) pf_fwd(W(Wk), W(Bk), wave, lane, r); lin_fwd_r(r, A, out, wave, lane); } while (0)
#define MAT_NEXT(r, Wk, Bk) do { if (PF) pf_fwd(W(Wk), W(Bk), wave, lane, r); } while (0)
template <bool PF, typename Pub = MatNoPub>
__device__ __forceinline__ void mat_encoder_forward(const MatCtx& c, int tid, int wave, int lane, const Pub pub = Pub()) {
    MAT_TILES(c);
    MatFr rq, rk, rv, rp, rm1, rm2, rh1;
    MAT_NEXT(rq, C_Q_W, C_Q_B); MAT_NEXT(rk, C_K_W, C_K_B); MAT_NEXT(rv, C_V_W, C_V_B);       // two phases ahead of their use
    tile_ln_fwd(sYO, OS, O, W(C_OLN_G), W(C_OLN_B), sXO, rstd(0), sYO, tid);                 // obs_encoder.0
    MAT_SYNC();
    MAT_STAMP(16);
    narrow_fwd(W(C_ENC_W), O, W(C_ENC_B), sYO, OS, cZ1, wave, lane);                          // obs_encoder.1
    MAT_SYNC();
    MAT_STAMP(17);
    tile_gelu(cZ1, S0, tid);
    tile_ln_fwd(S0, kMHS, kMD, W(C_LN_G), W(C_LN_B), cN0, rstd(1), S1, tid);                  // critic.ln -> H0 in S1
    MAT_SYNC();
    MAT_STAMP(18);
    pub(X_H0, S1);
    MAT_NEXT(rp, C_P_W, C_P_B);
    MAT_LIN(rq, C_Q_W, C_Q_B, S1, cQ);
    MAT_LIN(rk, C_K_W, C_K_B, S1, cK);
    MAT_LIN(rv, C_V_W, C_V_B, S1, cV);
    MAT_SYNC();
    MAT_STAMP(19);
    att_fwd(cQ, cK, cV, cY, sP0, L, n_rows, false, wave, lane);
    MAT_SYNC();
    MAT_STAMP(20);
    pub(X_CY, cY);
    MAT_NEXT(rm1, C_M1_W, C_M1_B);
    MAT_LIN(rp, C_P_W, C_P_B, cY, S2);
    MAT_SYNC();
    MAT_STAMP(21);
    tile_add(S1, S2, S2, tid);
    tile_ln_fwd(S2, kMHS, kMD, W(C_LN1_G), W(C_LN1_B), cN1, rstd(2), S0, tid);                // H1 in S0
    MAT_SYNC();
    MAT_STAMP(22);
    pub(X_H1, S0);
    MAT_NEXT(rm2, C_M2_W, C_M2_B);
    MAT_LIN(rm1, C_M1_W, C_M1_B, S0, cZM);
    MAT_SYNC();
    MAT_STAMP(23);
    tile_gelu(cZM, S2, tid);
    MAT_SYNC();
    MAT_STAMP(24);
    pub(X_GCM, S2);
    MAT_NEXT(rh1, C_H1_W, C_H1_B);
    MAT_LIN(rm2, C_M2_W, C_M2_B, S2, S3);
    MAT_SYNC();
    MAT_STAMP(25);
    tile_add(S0, S3, S3, tid);
    tile_ln_fwd(S3, kMHS, kMD, W(C_LN2_G), W(C_LN2_B), cN2, rstd(3), cENC, tid);              // rep_enc
    MAT_SYNC();
    MAT_STAMP(26);
    pub(X_ENC, cENC);
    MAT_LIN(rh1, C_H1_W, C_H1_B, cENC, cZH);
    MAT_SYNC();
    MAT_STAMP(27);
    tile_gelu(cZH, S0, tid);
    tile_ln_fwd(S0, kMHS, kMD, W(C_HLN_G), W(C_HLN_B), cNH, rstd(4), S1, tid);
    MAT_SYNC();
    MAT_STAMP(28);
    head_out_fwd(W(C_H2_W), W(C_H2_B), 1, S1, sOutC, tid);

}

// actor (decoder): token block in sXA + rep_enc in cENC -> logits in sOutA; starts and ends with the tiles settled
template <bool PF, typename Pub = MatNoPub>
__device__ __forceinline__ void mat_decoder_forward(const MatCtx& c, int tid, int wave, int lane, const Pub pub = Pub()) {
    MAT_TILES(c);
    MatFr rk1, rq1, rv1, rp1, rk2, rv2, rq2, rp2, rm1, rm2, rh1;
    MAT_NEXT(rk1, A_K1_W, A_K1_B); MAT_NEXT(rq1, A_Q1_W, A_Q1_B); MAT_NEXT(rv1, A_V1_W, A_V1_B);
    narrow_fwd(W(A_ENC_W), Ain, nullptr, sXA, kMXS, aZ, wave, lane);                          // action_encoder.0 (no bias)
    MAT_SYNC();
    tile_gelu(aZ, S0, tid);
    tile_ln_fwd(S0, kMHS, kMD, W(A_LN_G), W(A_LN_B), aN0, rstd(5), S1, tid);                  // x0 in S1
    MAT_SYNC();
    pub(X_X0, S1);
    MAT_NEXT(rp1, A_P1_W, A_P1_B);
    MAT_LIN(rk1, A_K1_W, A_K1_B, S1, aK1);
    MAT_LIN(rq1, A_Q1_W, A_Q1_B, S1, aQ1);
    MAT_LIN(rv1, A_V1_W, A_V1_B, S1, aV1);
    MAT_SYNC();
    att_fwd(aQ1, aK1, aV1, aY1, sP1, L, n_rows, true, wave, lane);
    MAT_SYNC();
    pub(X_AY1, aY1);
    MAT_NEXT(rk2, A_K2_W, A_K2_B); MAT_NEXT(rv2, A_V2_W, A_V2_B); MAT_NEXT(rq2, A_Q2_W, A_Q2_B);
    MAT_LIN(rp1, A_P1_W, A_P1_B, aY1, S2);
    MAT_SYNC();
    tile_add(S1, S2, S2, tid);
    tile_ln_fwd(S2, kMHS, kMD, W(A_LN1_G), W(A_LN1_B), aN1, rstd(6), S0, tid);                // x1 in S0
    MAT_SYNC();
    pub(X_X1, S0);
    MAT_NEXT(rp2, A_P2_W, A_P2_B);
    MAT_LIN(rk2, A_K2_W, A_K2_B, S0, aK2);                                       // key = value = x1
    MAT_LIN(rv2, A_V2_W, A_V2_B, S0, aV2);
    MAT_LIN(rq2, A_Q2_W, A_Q2_B, cENC, aQ2);                                     // query = rep_enc
    MAT_SYNC();
    att_fwd(aQ2, aK2, aV2, aY2, sP2, L, n_rows, true, wave, lane);
    MAT_SYNC();
    pub(X_AY2, aY2);
    MAT_NEXT(rm1, A_M1_W, A_M1_B);
    MAT_LIN(rp2, A_P2_W, A_P2_B, aY2, S2);
    MAT_SYNC();
    tile_add(cENC, S2, S2, tid);
    tile_ln_fwd(S2, kMHS, kMD, W(A_LN2_G), W(A_LN2_B), aN2, rstd(7), S1, tid);                // x2 in S1
    MAT_SYNC();
    pub(X_X2, S1);
    MAT_NEXT(rm2, A_M2_W, A_M2_B);
    MAT_LIN(rm1, A_M1_W, A_M1_B, S1, aZM);
    MAT_SYNC();
    tile_gelu(aZM, S2, tid);
    MAT_SYNC();
    pub(X_GAM, S2);
    MAT_NEXT(rh1, A_H1_W, A_H1_B);
    MAT_LIN(rm2, A_M2_W, A_M2_B, S2, S3);
    MAT_SYNC();
    tile_add(S1, S3, S3, tid);
    tile_ln_fwd(S3, kMHS, kMD, W(A_LN3_G), W(A_LN3_B), aN3, rstd(8), S0, tid);                // x3 in S0
    MAT_SYNC();
    pub(X_X3, S0);
    MAT_LIN(rh1, A_H1_W, A_H1_B, S0, aZH);
    MAT_SYNC();
    tile_gelu(aZH, S2, tid);
    tile_ln_fwd(S2, kMHS, kMD, W(A_HLN_G), W(A_HLN_B), aNH, rstd(9), S1, tid);                // head LayerNorm output in S1
    MAT_SYNC();
    head_out_fwd(W(A_H2_W), W(A_H2_B), NA, S1, sOutA, tid);
    MAT_SYNC();

}


// SPLIT: the weight gradients of the 18 64x64 linears are NOT formed here (a third of the kernel's MFMA work, on the
// one CU that owns the tile: ablation 72 -> 59 us); the D and input tiles each of them needs go to the panels instead.
// (the inputs were published by the forward pass: MatPanelPub)
#define MAT_WGRAD(lin, D, In, Wk, Bk)                                                               \
    do {                                                                                            \
        if constexpr (SPLIT) publish_tile(D, u.dpanel + (long)(lin) * u.R * kMD, g, tid);           \
        else lin_wgrad(D, In, G(Wk), G(Bk), wave, lane, tid);                                       \
    } while (0)
// slab form only: an input tile rebuilt for a weight gradient (and the barrier that settles it)
#define MAT_SLAB_ONLY(stmt) do { if constexpr (!SPLIT) { stmt; } } while (0)
template <bool SPLIT>
__global__ __launch_bounds__(kMT) void mat_update_fwd_bwd_kernel(MatDev u) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x;
    const int NT0 = (u.O + 15) >> 4;
    const float* P = u.params;
    float* slab = u.slabs + (long)g * u.slab_stride;
    const long mb = u.cursor[0] + u.mb_offset;
    const long seq0 = (long)g * u.per_tile;
    const long rem = u.B - seq0;
    const int n_seq = (int)(rem < u.per_tile ? (rem < 0 ? 0 : rem) : u.per_tile);
    const float inv_n = 1.0f / (float)(u.B * u.L);

    MAT_STAMP(0);
    // ---- LDS carve
    MatCtx c;
    c.P = P; c.off = u.off; c.L = u.L; c.NA = u.NA; c.Ain = u.Ain; c.n_rows = n_seq * u.L;
    c.carve(reinterpret_cast<float*>(mat_smem), u.O);
    MAT_TILES(c);
    int* sRow = c.sRow; int* sAct = c.sAct; float* sMisc = c.sMisc; float* sRowF = c.sRowF;
    float* sDOutC = c.sDOutC; float* sDOutA = c.sDOutA;
    auto G = [&](int k) -> float* { return slab + u.goff[k]; };

    // ---- rows, per-token scalars, mini-batch statistics
    if (tid < kRows) {
        int row = -1, act = 0;
        float av = 0.f, lpo = 0.f, rt = 0.f;
        if (tid < n_rows) {
            const int s = tid / L, a = tid - s * L;
            const long pos = mb * u.batch_stride + seq0 + s;
            const long p = u.perm[pos];
            if (u.pregathered) {
                // per-epoch tables in shuffled order: entry `pos` belongs to perm[pos], so these loads depend on the cursor
                // only and go out together with the perm load (instead of behind perm -> row_map: two cold round trips)
                const long tok = pos * L + a;
                const int a_raw = (int)u.actions[tok];
                const float av_ = u.adv[tok], lpo_ = u.old_lp[tok], rt_ = u.rtg[tok];
                if (p >= 0 && p < u.n_rows) {
                    row = u.row_map ? u.row_map[p] : (int)p;
                    act = a_raw < 0 ? 0 : (a_raw >= NA ? NA - 1 : a_raw);
                    av = av_; lpo = lpo_; rt = rt_;
                }
            } else if (p >= 0 && p < u.n_rows) {
                row = u.row_map ? u.row_map[p] : (int)p;
                const long tok = (long)row * L + a;
                act = (int)u.actions[tok];
                act = act < 0 ? 0 : (act >= NA ? NA - 1 : act);
                av = u.adv[tok]; lpo = u.old_lp[tok]; rt = u.rtg[tok];
            }
        }
        sRow[tid] = row; sAct[tid] = act;
        sRowF[tid] = av; sRowF[16 + tid] = lpo; sRowF[32 + tid] = rt;
    }
    if (tid == 64) {
        float mean_f = 0.f, std_f = 1.f;
        if (u.normalize_adv) {
            const double* rec = u.adv_records + mb * 3;
            mean_f = (float)rec[1];
            std_f = (float)sqrt(rec[2] / (rec[0] - 1.0));
        }
        sMisc[0] = mean_f; sMisc[1] = std_f;
    }
    if (tid == 128) {
        // value normaliser: Chan merge of the rank records of this mini-batch + the reference's integrate
        // (utils/stats.py:73-94), as in K12
        const int slot = (int)(mb & 1);
        float m = u.vn_mean[slot], v = u.vn_var[slot];
        double cnt = u.vn_count[slot];
        if (u.normalize_values) {
            double n = 0.0, bm = 0.0, M2 = 0.0;
            for (int r = 0; r < u.n_ranks; ++r) {
                const double* rec = u.vn_records + (mb * u.n_ranks + r) * 3;
                const double nb = rec[0];
                if (nb <= 0.0) continue;
                const double d = rec[1] - bm, nn = n + nb;
                bm += d * (nb / nn);
                M2 += rec[2] + d * d * n * nb / nn;
                n = nn;
            }
            if (n > 0.0) {
                const float batch_mean = (float)bm, batch_var = (float)(M2 / n);
                const float delta = batch_mean - m;
                const double new_count = cnt + n;
                const float new_mean = (float)((double)m + (double)delta * (n / new_count));
                const double m_2 = (double)v * cnt + (double)batch_var * n + (double)(delta * delta) * cnt * n / (cnt + n);
                m = new_mean; v = (float)(m_2 / (cnt + n)); cnt = new_count;
            }
        }
        sMisc[2] = m; sMisc[3] = v;
        if (g == 0) { u.vn_mean[slot ^ 1] = m; u.vn_var[slot ^ 1] = v; u.vn_count[slot ^ 1] = cnt; }
    }
    for (int i = tid; i < kRows * OS; i += kMT) { sXO[i] = 0.f; sYO[i] = 0.f; }
    for (int i = tid; i < kRows * kMXS; i += kMT) sXA[i] = 0.f;
    MAT_SYNC();
    MAT_STAMP(1);
    // observations of the tile's tokens; the shifted one-hot token block (mat_policy.py:308-344,378-416)
    for (int idx = tid; idx < n_rows * O; idx += kMT) {
        const int s = idx / O, i = idx - s * O;
        const int row = sRow[s];
        const long orow = u.pregathered ? mb * u.batch_stride + seq0 + s / L : (long)row;
        if (row >= 0) sYO[s * OS + i] = u.obs[(orow * L + (s % L)) * O + i];
    }
    if (tid < n_rows && sRow[tid] >= 0) {
        const int a = tid % L;
        if (a == 0) sXA[tid * kMXS] = 1.0f;
        else sXA[tid * kMXS + 1 + sAct[tid - 1]] = 1.0f;
    }
    MAT_SYNC();
    MAT_STAMP(2);

    if constexpr (SPLIT) {
        const MatPanelPub pub{u.xpanel, (long)u.R * kMD, g, tid};
        mat_encoder_forward<true>(c, tid, wave, lane, pub);
        MAT_STAMP(3);
        mat_decoder_forward<true>(c, tid, wave, lane, pub);
    } else {
        mat_encoder_forward<true>(c, tid, wave, lane);
        MAT_STAMP(3);
        mat_decoder_forward<true>(c, tid, wave, lane);
    }
    MAT_STAMP(4);
    // input-gradient (dgrad) fragment sets of the backward: each requested at the start of the phase before its use
    MatFr dH1, dM2, dM1, dP2, dQ2, dK2, dV2, dP1, dQ1, dK1, dV1;
    pf_dgrad(W(A_H1_W), wave, lane, dH1);

    // =========================================== heads: distribution + losses (K6 + K3) =============================
    // wave 0: the categorical head of the 16 token rows, 4 lanes per row (lane (s4, q) owns classes q and q + 4: the
    // transcendental chain is 2 values long instead of 8, class reductions are quad permutes -- as K12's head);
    // wave 1: the value loss of the rows, at the same time.
    if (wave == 0) {
        const int s4 = lane >> 2, q = lane & 3;
        const bool live4 = s4 < n_rows && sRow[s4] >= 0;
        float av4 = 0.f, lpo4 = 0.f;
        if (live4) {
            av4 = sRowF[s4]; lpo4 = sRowF[16 + s4];
            if (u.normalize_adv) av4 = (av4 - sMisc[0]) / (sMisc[1] + 1e-8f);
        }
        const int k0 = q, k1 = q + 4;
        const bool v0 = k0 < NA, v1 = k1 < NA;
        const float z0 = v0 ? sOutA[s4 * 8 + k0] : -INFINITY, z1 = v1 ? sOutA[s4 * 8 + k1] : -INFINITY;
        const float m = group4_max(fmaxf(z0, z1));
        float p0 = v0 ? expf(z0 - m) : 0.f, p1 = v1 ? expf(z1 - m) : 0.f;
        const float inv = 1.0f / group4_sum(p0 + p1);
        p0 *= inv; p1 *= inv;
        const float s2 = group4_sum(p0 + p1);
        const int a = sAct[s4];
        const float n0 = p0 / s2, n1 = p1 / s2;                                   // Categorical's renormalisation
        const float l0 = v0 ? logf(clamp_prob_u(n0)) : 0.f, l1 = v1 ? logf(clamp_prob_u(n1)) : 0.f;
        const float ent4 = -group4_sum((v0 ? n0 * l0 : 0.f) + (v1 ? n1 * l1 : 0.f));
        const float logp4 = group4_sum((k0 == a ? l0 : 0.f) + (k1 == a ? l1 : 0.f));
        const float ratio = expf(logp4 - lpo4);
        const float bad4 = (isnan(ratio) || isinf(ratio)) ? 1.f : 0.f;
        const float lo = 1.0f - u.surr_clip, hi = 1.0f + u.surr_clip;
        const float surr1 = ratio * av4, surr2 = fminf(fmaxf(ratio, lo), hi) * av4;
        float glp;
        if (surr1 <= surr2) glp = -av4 * ratio;
        else glp = (ratio >= lo && ratio <= hi) ? -av4 * ratio : 0.f;
        glp *= inv_n;
        const float gH = (u.entropy_weight != 0.0f) ? -u.entropy_weight * inv_n : 0.f;
        auto gk_of = [&](bool valid, int k, float nk, float lg) {               // chain: z -softmax-> p -(/sum)-> n -clamp,log-> l
            if (!valid) return 0.f;
            const float ck = clamp_prob_u(nk);
            const float in_range = (nk >= FLT_EPSILON && nk <= 1.0f - FLT_EPSILON) ? 1.f : 0.f;
            float gk = gH * (-lg - nk * in_range / ck);
            if (k == a) gk += glp * in_range / ck;
            return gk;
        };
        float g0 = gk_of(v0, k0, n0, l0), g1 = gk_of(v1, k1, n1, l1);
        const float dot = group4_sum(g0 * n0 + g1 * n1);
        g0 = (g0 - dot) / s2; g1 = (g1 - dot) / s2;
        const float dot2 = group4_sum(g0 * p0 + g1 * p1);
        sDOutA[s4 * 8 + k0] = live4 ? p0 * (g0 - dot2) : 0.f;
        sDOutA[s4 * 8 + k1] = live4 ? p1 * (g1 - dot2) : 0.f;
        // row results -> lane s (= row s) for the partial sums of the tile
        const int src = (lane & 15) * 4;
        const float r_live = __shfl(live4 ? 1.f : 0.f, src, 64);
        const float r_surr = __shfl(-fminf(surr1, surr2), src, 64), r_ent = __shfl(ent4, src, 64);
        const float r_kl = __shfl(lpo4 - logp4, src, 64), r_bad = __shfl(bad4, src, 64);
        const bool row_live = lane < kRows && r_live != 0.f;
        const float t0 = group16_sum(row_live ? r_surr : 0.f), t3 = group16_sum(row_live ? r_ent : 0.f);
        const float t4 = group16_sum(row_live ? r_kl : 0.f), t7 = group16_sum(row_live ? r_bad : 0.f);
        if (lane == 0) {
            float* lp = u.loss_partials + (long)g * 8;
            lp[0] = t0; lp[1] = 0.f; lp[3] = t3; lp[4] = t4; lp[7] = t7;
            lp[5] = g == 0 ? sMisc[0] : 0.f; lp[6] = g == 0 ? sMisc[1] : 0.f;
        }
    } else if (wave == 1) {
        const int s = lane;
        const bool live = s < n_rows && sRow[s] >= 0;
        float l = 0.f;
        if (live) {
            const float v = sOutC[s * 8];
            float rt = sRowF[32 + s];
            if (u.normalize_values) rt = (rt - sMisc[2]) / sqrtf(sMisc[3] + 1e-8f);
            const float diff = v - rt;
            float dl;
            if (u.use_huber) {
                const float ad = fabsf(diff);
                if (ad < u.huber_delta) { l = 0.5f * diff * diff; dl = diff; }
                else { l = u.huber_delta * (ad - 0.5f * u.huber_delta); dl = diff > 0.f ? u.huber_delta : -u.huber_delta; }
            } else { l = diff * diff; dl = 2.0f * diff; }
            sDOutC[s * 8] = dl * inv_n;
            u.values[(long)sRow[s] * L + (s % L)] = v;                                          // ppo.py:2340
        } else if (s < kRows) {
#pragma unroll
            for (int k = 0; k < 8; ++k) sDOutC[s * 8 + k] = 0.f;
        }
        const float t2 = group16_sum((lane < kRows && live) ? l : 0.f);
        if (lane == 0) u.loss_partials[(long)g * 8 + 2] = t2;
    }
    MAT_SYNC();
    MAT_STAMP(5);
    // =========================================== actor backward ===========================================
    // S1 still holds the head LayerNorm output
    head_out_bwd(W(A_H2_W), NA, S1, sDOutA, S2, G(A_H2_W), G(A_H2_B), tid);                   // d head-LN out -> S2
    MAT_SYNC();
    tile_ln_bwd(S2, kMHS, kMD, aNH, rstd(9), W(A_HLN_G), S3, G(A_HLN_G), G(A_HLN_B), tid);     // -> d gelu out in S3
    tile_gelu_bwd(S3, aZH, tid);                                                              // d aZH
    MAT_SLAB_ONLY(tile_affine(aN3, W(A_LN3_G), W(A_LN3_B), S0, tid));                         // x3
    MAT_SYNC();
    MAT_WGRAD(L_A_H1, S3, S0, A_H1_W, A_H1_B);
    pf_dgrad(W(A_M2_W), wave, lane, dM2);
    lin_dgrad_r<false>(dH1, S3, S2, wave, lane);                                          // d x3 -> S2
    MAT_SYNC();
    tile_ln_bwd(S2, kMHS, kMD, aN3, rstd(8), W(A_LN3_G), S4, G(A_LN3_G), G(A_LN3_B), tid);     // d r3 -> S4 (= d x2 residual = d mlp out)
    MAT_SLAB_ONLY(tile_gelu(aZM, S0, tid));                                                   // mlp hidden activation
    MAT_SYNC();
    MAT_WGRAD(L_A_M2, S4, S0, A_M2_W, A_M2_B);
    pf_dgrad(W(A_M1_W), wave, lane, dM1);
    lin_dgrad_r<false>(dM2, S4, S3, wave, lane);
    MAT_SYNC();
    tile_gelu_bwd(S3, aZM, tid);                                                              // d aZM
    MAT_SLAB_ONLY(tile_affine(aN2, W(A_LN2_G), W(A_LN2_B), S0, tid));                         // x2
    MAT_SYNC();
    MAT_WGRAD(L_A_M1, S3, S0, A_M1_W, A_M1_B);
    pf_dgrad(W(A_P2_W), wave, lane, dP2);
    lin_dgrad_r<true>(dM1, S3, S4, wave, lane);                                           // d x2 total in S4
    MAT_SYNC();
    tile_ln_bwd(S4, kMHS, kMD, aN2, rstd(7), W(A_LN2_G), DENC, G(A_LN2_G), G(A_LN2_B), tid);   // d r2 -> DENC (rep_enc share) = d proj2 out
    MAT_SYNC();
    MAT_WGRAD(L_A_P2, DENC, aY2, A_P2_W, A_P2_B);
    pf_dgrad(W(A_Q2_W), wave, lane, dQ2); pf_dgrad(W(A_K2_W), wave, lane, dK2); pf_dgrad(W(A_V2_W), wave, lane, dV2);
    lin_dgrad_r<false>(dP2, DENC, S2, wave, lane);                                        // d Y2 -> S2
    MAT_SYNC();
    MAT_STAMP(6);
    att_bwd(aQ2, aK2, aV2, sP2, S2, S0, S3, S4, sS, wave, lane);                              // dQ2 -> S0, dK2 -> S3, dV2 -> S4
    MAT_SYNC();
    MAT_STAMP(7);
    MatFr eH1, eM2, eM1, eP, eQ, eK, eV;
    MAT_WGRAD(L_A_Q2, S0, cENC, A_Q2_W, A_Q2_B);
    lin_dgrad_r<true>(dQ2, S0, DENC, wave, lane);                                         // rep_enc gradient from the query path
    MAT_SLAB_ONLY(tile_affine(aN1, W(A_LN1_G), W(A_LN1_B), S1, tid); MAT_SYNC());             // x1
    MAT_WGRAD(L_A_K2, S3, S1, A_K2_W, A_K2_B);
    MAT_WGRAD(L_A_V2, S4, S1, A_V2_W, A_V2_B);
    pf_dgrad(W(A_P1_W), wave, lane, dP1);
    lin_dgrad_r<false>(dK2, S3, S2, wave, lane);
    MAT_SLAB_ONLY(MAT_SYNC());                       // (a thread accumulates into its own elements: no barrier needed)
    lin_dgrad_r<true>(dV2, S4, S2, wave, lane);                                           // d x1 -> S2
    MAT_SYNC();
    tile_ln_bwd(S2, kMHS, kMD, aN1, rstd(6), W(A_LN1_G), S0, G(A_LN1_G), G(A_LN1_B), tid);     // d r1 -> S0 (= d x0 residual = d proj1 out)
    MAT_SYNC();
    MAT_WGRAD(L_A_P1, S0, aY1, A_P1_W, A_P1_B);
    pf_dgrad(W(A_Q1_W), wave, lane, dQ1); pf_dgrad(W(A_K1_W), wave, lane, dK1); pf_dgrad(W(A_V1_W), wave, lane, dV1);
    lin_dgrad_r<false>(dP1, S0, S2, wave, lane);                                          // d Y1 -> S2
    MAT_SYNC();
    att_bwd(aQ1, aK1, aV1, sP1, S2, S1, S3, S4, sS, wave, lane);                              // dQ1 -> S1, dK1 -> S3, dV1 -> S4
    MAT_SYNC();                                                                               // dY (S2) fully consumed
    MAT_SLAB_ONLY(tile_affine(aN0, W(A_LN_G), W(A_LN_B), S2, tid); MAT_SYNC());               // x0
    MAT_WGRAD(L_A_Q1, S1, S2, A_Q1_W, A_Q1_B);
    MAT_WGRAD(L_A_K1, S3, S2, A_K1_W, A_K1_B);
    MAT_WGRAD(L_A_V1, S4, S2, A_V1_W, A_V1_B);
    pf_dgrad(W(C_H1_W), wave, lane, eH1);
    lin_dgrad_r<true>(dQ1, S1, S0, wave, lane);
    lin_dgrad_r<true>(dK1, S3, S0, wave, lane);
    lin_dgrad_r<true>(dV1, S4, S0, wave, lane);                                           // d x0 total in S0
    MAT_SYNC();
    tile_ln_bwd(S0, kMHS, kMD, aN0, rstd(5), W(A_LN_G), S1, G(A_LN_G), G(A_LN_B), tid);        // d gelu out -> S1
    tile_gelu_bwd(S1, aZ, tid);
    MAT_SYNC();
    layer_wgrad<4, kMNW>(S1, sXA, kMXS, 1, Ain, G(A_ENC_W), Ain, nullptr, wave, lane, tid);

    MAT_STAMP(8);
    // =========================================== critic backward ===========================================
    tile_affine(cNH, W(C_HLN_G), W(C_HLN_B), S0, tid);                                        // head LayerNorm output
    MAT_SYNC();
    head_out_bwd(W(C_H2_W), 1, S0, sDOutC, S2, G(C_H2_W), G(C_H2_B), tid);
    MAT_SYNC();
    tile_ln_bwd(S2, kMHS, kMD, cNH, rstd(4), W(C_HLN_G), S3, G(C_HLN_G), G(C_HLN_B), tid);
    tile_gelu_bwd(S3, cZH, tid);
    MAT_SYNC();
    MAT_WGRAD(L_C_H1, S3, cENC, C_H1_W, C_H1_B);
    pf_dgrad(W(C_M2_W), wave, lane, eM2);
    lin_dgrad_r<true>(eH1, S3, DENC, wave, lane);                                         // total d rep_enc
    MAT_SYNC();
    tile_ln_bwd(DENC, kMHS, kMD, cN2, rstd(3), W(C_LN2_G), S4, G(C_LN2_G), G(C_LN2_B), tid);   // d r2 -> S4
    MAT_SLAB_ONLY(tile_gelu(cZM, S0, tid));
    MAT_SYNC();
    MAT_WGRAD(L_C_M2, S4, S0, C_M2_W, C_M2_B);
    pf_dgrad(W(C_M1_W), wave, lane, eM1);
    lin_dgrad_r<false>(eM2, S4, S3, wave, lane);
    MAT_SYNC();
    tile_gelu_bwd(S3, cZM, tid);
    MAT_SLAB_ONLY(tile_affine(cN1, W(C_LN1_G), W(C_LN1_B), S0, tid));                         // H1
    MAT_SYNC();
    MAT_WGRAD(L_C_M1, S3, S0, C_M1_W, C_M1_B);
    pf_dgrad(W(C_P_W), wave, lane, eP);
    lin_dgrad_r<true>(eM1, S3, S4, wave, lane);                                           // d H1 total
    MAT_SYNC();
    tile_ln_bwd(S4, kMHS, kMD, cN1, rstd(2), W(C_LN1_G), S0, G(C_LN1_G), G(C_LN1_B), tid);     // d r1 -> S0
    MAT_SYNC();
    MAT_WGRAD(L_C_P, S0, cY, C_P_W, C_P_B);
    pf_dgrad(W(C_Q_W), wave, lane, eQ); pf_dgrad(W(C_K_W), wave, lane, eK); pf_dgrad(W(C_V_W), wave, lane, eV);
    lin_dgrad_r<false>(eP, S0, S2, wave, lane);                                           // d Y
    MAT_SYNC();
    att_bwd(cQ, cK, cV, sP0, S2, S1, S3, S4, sS, wave, lane);                                 // dQ -> S1, dK -> S3, dV -> S4
    MAT_SYNC();
    MAT_SLAB_ONLY(tile_affine(cN0, W(C_LN_G), W(C_LN_B), S2, tid); MAT_SYNC());               // H0
    MAT_WGRAD(L_C_Q, S1, S2, C_Q_W, C_Q_B);
    MAT_WGRAD(L_C_K, S3, S2, C_K_W, C_K_B);
    MAT_WGRAD(L_C_V, S4, S2, C_V_W, C_V_B);
    lin_dgrad_r<true>(eQ, S1, S0, wave, lane);
    lin_dgrad_r<true>(eK, S3, S0, wave, lane);
    lin_dgrad_r<true>(eV, S4, S0, wave, lane);                                            // d H0 total
    MAT_SYNC();
    tile_ln_bwd(S0, kMHS, kMD, cN0, rstd(1), W(C_LN_G), S1, G(C_LN_G), G(C_LN_B), tid);
    tile_gelu_bwd(S1, cZ1, tid);
    MAT_SYNC();
    layer_wgrad<4, kMNW>(S1, sYO, OS, NT0, O, G(C_ENC_W), O, G(C_ENC_B), wave, lane, tid);
    narrow_dgrad(W(C_ENC_W), O, S1, S2, kMHS, wave, lane);                                     // d (obs LayerNorm output) in S2[:, :O]
    MAT_SYNC();
    MAT_STAMP(9);
    // observation LayerNorm: only its affine parameters receive gradient
    if (tid < 64) {
        if (tid < O) {
            float a = 0.f;
#pragma unroll
            for (int s = 0; s < kRows; ++s) a = fmaf(S2[s * kMHS + tid], sXO[s * OS + tid], a);
            G(C_OLN_G)[tid] = a;
        }
    } else if (tid < 128) {
        const int c = tid - 64;
        if (c < O) {
            float a = 0.f;
#pragma unroll
            for (int s = 0; s < kRows; ++s) a += S2[s * kMHS + c];
            G(C_OLN_B)[c] = a;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K16: one rollout step of a MATPolicy for all envs of the rank (mat_policy.py:441-519,587-626,660-675):
// encoder forward (values), then L autoregressive decoder passes -- agent i's action is sampled
// (Philox4x32-10, counter = offset + env * L + i) from the logits of pass i and becomes agent i+1's
// token -- and the step's row of the rollout buffer is written in place.
// ------------------------------------------------------------------------------------------------
struct MatStepDev {
    long off[64];
    int O, L, NA, Ain, per_tile, Oa;
    const float* params;
    const float* obs; const float* actor_obs; long E;
    unsigned long long seed, offset;
    int normalize_values; const float* vn_mean; const float* vn_var;
    int64_t* action_out; int64_t* raw_action_out; float* logp_out; float* value_out;
    float* critic_obs_out; float* obs_out;
    const int64_t* forced_action;                  // NULL: sample; else [E, L] actions to log (replay)
};

__global__ __launch_bounds__(kMT) void mat_policy_step_kernel(MatStepDev u) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long seq0 = (long)blockIdx.x * u.per_tile;
    const long rem = u.E - seq0;
    const int n_seq = (int)(rem < u.per_tile ? (rem < 0 ? 0 : rem) : u.per_tile);
    MatCtx c;
    c.P = u.params; c.off = u.off; c.L = u.L; c.NA = u.NA; c.Ain = u.Ain; c.n_rows = n_seq * u.L;
    c.carve(reinterpret_cast<float*>(mat_smem), u.O);
    MAT_TILES(c);
    int* sAct = c.sAct;
    for (int i = tid; i < kRows * OS; i += kMT) { sXO[i] = 0.f; sYO[i] = 0.f; }
    for (int i = tid; i < kRows * kMXS; i += kMT) sXA[i] = 0.f;
    MAT_SYNC();
    const long tok0 = seq0 * L;                                  // first token (env-major, agents inside)
    for (int idx = tid; idx < n_rows * O; idx += kMT) {
        const int s = idx / O, i = idx - s * O;
        const float v = u.obs[(tok0 + s) * O + i];
        sYO[s * OS + i] = v;
        if (u.critic_obs_out) u.critic_obs_out[(tok0 + s) * O + i] = v;
    }
    if (u.obs_out) {
        const float* src = u.actor_obs ? u.actor_obs : u.obs;
        for (long idx = tid; idx < (long)n_rows * u.Oa; idx += kMT) u.obs_out[tok0 * u.Oa + idx] = src[tok0 * u.Oa + idx];
    }
    if (tid < n_rows && tid % L == 0) sXA[tid * kMXS] = 1.0f;     // start token of agent 0 (mat_policy.py:325-333)
    MAT_SYNC();
    mat_encoder_forward<false>(c, tid, wave, lane);
    MAT_SYNC();
    if (tid < n_rows) {
        float v = sOutC[tid * 8];
        if (u.normalize_values) v = u.vn_mean[0] + v * sqrtf(u.vn_var[0] + 1e-8f);     // misc.py:124-128
        u.value_out[tok0 + tid] = v;
    }
    for (int i = 0; i < L; ++i) {
        mat_decoder_forward<false>(c, tid, wave, lane);           // ends with a barrier
        if (tid < n_rows && tid % L == i) {
            float p[8];
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < 8; ++k) if (k < NA) m = fmaxf(m, sOutA[tid * 8 + k]);
            float ssum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) { p[k] = k < NA ? expf(sOutA[tid * 8 + k] - m) : 0.f; ssum += p[k]; }
            const float inv = 1.0f / ssum;
            float s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) { p[k] *= inv; s2 += p[k]; }
            int a = NA - 1;
            float cum = 0.f, pa = 0.f;
            if (u.forced_action) {
                const long fa = u.forced_action[tok0 + tid];
                a = fa < 0 ? 0 : (fa >= NA ? NA - 1 : (int)fa);
            } else {
                const Philox4 rnd = philox4x32_10(u.seed, u.offset + (unsigned long long)(tok0 + tid), 0u);
                const float uu = u32_to_unit(rnd.x) * s2;         // inverse CDF over the probability mass
                bool found = false;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k < NA && !found) {
                        cum += p[k];
                        if (uu < cum) { a = k; found = true; }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) if (k == a) pa = p[k];
            sAct[tid] = a;
            u.action_out[tok0 + tid] = a;
            if (u.raw_action_out) u.raw_action_out[tok0 + tid] = a;
            u.logp_out[tok0 + tid] = logf(clamp_prob_u(pa / s2));
            if (i + 1 < L) sXA[(tid + 1) * kMXS + 1 + a] = 1.0f;  // agent i+1 sees agent i's action
        }
        MAT_SYNC();
    }
}

// slabs -> gradient bucket in a fixed order (each thread owns a float4 column, 8 slab loads in flight at a
// time, no LDS staging); the last workgroup folds the loss partials and advances the cursor
constexpr int kMatRedThreads = 256;
__device__ __forceinline__ void mat_update_bookkeeping(const MatDev& u) {
    {
        if (threadIdx.x >= 64) return;
        const int lane = threadIdx.x;
        float p0 = 0.f, p2 = 0.f, p3 = 0.f, p4 = 0.f, p7 = 0.f;
        for (int g = lane; g < u.nT; g += 64) {
            const float* a = u.loss_partials + (long)g * 8;
            p0 += a[0]; p2 += a[2]; p3 += a[3]; p4 += a[4]; p7 += a[7];
        }
        p0 = wave_sum(p0); p2 = wave_sum(p2); p3 = wave_sum(p3); p4 = wave_sum(p4); p7 = wave_sum(p7);
        if (lane == 0) {
            const float n = (float)(u.B * u.L);
            const float surr = p0 / n, ent = p3 / n, kl = p4 / n, crit = p2 / n;
            float total = surr;
            if (u.entropy_weight != 0.0f) total -= u.entropy_weight * ent;
            if (u.kl_loss_weight > 0.0f) total += u.kl_loss_weight * kl;
            u.totals[0] += (double)surr; u.totals[1] += (double)total; u.totals[2] += (double)crit;
            u.totals[3] += (double)ent; u.totals[4] += (double)kl;
            u.totals[5] += (double)u.loss_partials[5]; u.totals[6] += (double)u.loss_partials[6];
            u.totals[7] += p7 > 0.f ? 1.0 : 0.0;
            u.totals[8] += 1.0;
            if (u.cursor_advance) u.cursor[0] += u.cursor_advance;
            if (u.fuse_norm) u.step_count[0] += 1;
        }
    }
}
__global__ __launch_bounds__(kMatRedThreads) void mat_update_reduce_kernel(MatDev u) {
    if (blockIdx.x == gridDim.x - 1) { mat_update_bookkeeping(u); return; }
    __shared__ double red[17];
    const long n4 = u.total >> 2;
    const long idx = (long)blockIdx.x * kMatRedThreads + threadIdx.x;
    const float4* sl = reinterpret_cast<const float4*>(u.slabs);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // 52 slabs at C5: 16 loads of a column in flight at a time (the slabs were written by other XCDs a moment ago: every
    // batch is one cold round trip, so the launch lasts ceil(slabs / 16) of them), added in slab order
    if (idx < n4)
    for (int g0 = 0; g0 < u.nT; g0 += 16) {
        float4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k)
            v[k] = (g0 + k < u.nT) ? sl[(long)(g0 + k) * n4 + idx] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
    }
    if (idx < n4) reinterpret_cast<float4*>(u.grads)[idx] = acc;
    if (u.fuse_norm) {                                        // uniform per launch: every thread reaches the barriers
        double q = (double)acc.x * acc.x + (double)acc.y * acc.y + (double)acc.z * acc.z + (double)acc.w * acc.w;
        q = block_sum(q, red);
        if (threadIdx.x == 0) u.norm_scratch[2 + blockIdx.x] = q;      // partials: ppoaf_adam_step_prenormed adds them in a fixed order
    }
}

// Split-wgrad chain: what ppoaf_mat_update_reduce launches when args->split_workspace is set.
//   workgroups [0, 144): 16 x 32 of dW_k = dz_k^T x_k over ALL R rows of the mini-batch (K = R on f32 MFMA, the four
//     waves take every fourth 16-row chunk and are folded in wave order) for the 18 linears; the jobs of input half 0
//     also form db_k = column sums of dz_k.  Jobs are dealt so that the ~2 linears an XCD works on stay in its L2:
//     workgroup b runs on XCD b % 8 (round-robin dispatch; placement only changes speed) and takes job (b % 8) * 18 + b / 8
//     of the linear-major job list.
//   then ceil(4 n_small4 / 256) workgroups: the compact slabs of the small tensors (LayerNorm gains, narrow layers, heads)
//     -> gradient bucket, in slab order (as mat_update_reduce_kernel);
//   last workgroup: loss partials -> totals, cursor, step counter.
// fuse_norm: one squared-norm partial per workgroup in norm_scratch[2 + b].
constexpr int kMatWgJobs = kMatLin * 8;
__global__ __launch_bounds__(256) void mat_update_wgrad_kernel(MatDev u, int n_small_blocks) {
    __shared__ double s_red[17];
    __shared__ __attribute__((aligned(16))) float s_fold[2 * 3 * 256 + 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: the chunk offsets below stay in scalar registers
    if (b == kMatWgJobs + n_small_blocks) { mat_update_bookkeeping(u); return; }          // uniform per workgroup
    double q = 0.0;
    if (b < kMatWgJobs) {
        // a job = 16 output rows x 32 input columns of one linear's dW: the X operand then uses whole 128-byte lines and
        // the dz operand is shared by the job's two MFMA tiles (counters of the 16 x 16 form: 66 MB of L2 requests per
        // launch for 7 MB of panels -- every line half used and fetched by four jobs -- at 7.5 TB/s: L2-bandwidth bound)
        constexpr int per_xcd = kMatWgJobs / 8;
        const int job = (b & 7) * per_xcd + (b >> 3);
        const int k = job >> 3, ot = (job >> 1) & 3, ih = job & 1;
        const int wk = kMatLinW[k];
        const long plane = (long)u.R * kMD;
        // buffer loads: resource = the panel, scalar offset = chunk + row quad, vector offset = the lane's constant byte
        // offset: no vector address arithmetic per load (the same launch with 64-bit addresses spent as many VALU
        // instructions on addresses as on everything else)
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(u.dpanel + (long)k * plane, 0, 0xFFFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(u.xpanel + (long)kMatLinX[k] * plane, 0, 0xFFFFFFFF, 0x00020000);
        const unsigned dl = 4u * (unsigned)((lane >> 4) * kMD + ot * 16 + (lane & 15));
        const unsigned xl = 4u * (unsigned)((lane >> 4) * kMD + ih * 32 + (lane & 15));
        const int nc = u.nT;                                   // 16-row chunks
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        float bsum = 0.f;
        // every operand of the job is requested before the first MFMA (the panels were written by other XCDs a moment
        // ago: each batch is one cold round trip, so there must be one batch at the BASELINE sizes: 52 chunks / 4 waves)
        constexpr int MAXC = 16;
        for (int c0 = wave; c0 < nc; c0 += 4 * MAXC) {         // wave-uniform trip count
            float a[MAXC][4], x0[MAXC][4], x1[MAXC][4];
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int ch = c0 + 4 * c;
                if (ch < nc) {                                 // wave-uniform
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned so = 4u * (unsigned)((16 * ch + 4 * j) * kMD);
                        a[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, dl, so, 0));
                        x0[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xl, so, 0));
                        x1[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xl + 64u, so, 0));
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                if (c0 + 4 * c < nc) {                         // wave-uniform
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], x0[c][j], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], x1[c][j], acc1, 0, 0, 0);
                        bsum += a[c][j];
                    }
                }
            }
        }
        // fold the four waves' partial tiles in wave order (waves 1..3 park theirs in LDS)
        if (wave > 0) {
            *reinterpret_cast<f32x4*>(s_fold + (((wave - 1) * 2 + 0) * 64 + lane) * 4) = acc0;
            *reinterpret_cast<f32x4*>(s_fold + (((wave - 1) * 2 + 1) * 64 + lane) * 4) = acc1;
        }
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (lane < 16) s_fold[1536 + wave * 16 + lane] = bsum;
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                acc0 += *reinterpret_cast<const f32x4*>(s_fold + ((w * 2 + 0) * 64 + lane) * 4);
                acc1 += *reinterpret_cast<const f32x4*>(s_fold + ((w * 2 + 1) * 64 + lane) * 4);
            }
            float* GW = u.grads + u.off[wk];
            const int i = ih * 32 + (lane & 15);               // C layout: column = lane & 15, rows 4 (lane >> 4) + r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = ot * 16 + 4 * (lane >> 4) + r;
                GW[o * kMD + i] = acc0[r];
                GW[o * kMD + i + 16] = acc1[r];
                q += (double)acc0[r] * acc0[r] + (double)acc1[r] * acc1[r];
            }
            if (ih == 0 && lane < 16) {
                const float bg = s_fold[1536 + lane] + s_fold[1536 + 16 + lane] + s_fold[1536 + 32 + lane] + s_fold[1536 + 48 + lane];
                u.grads[u.off[wk + 1] + ot * 16 + lane] = bg;
                q += (double)bg * bg;
            }
        }
    } else {
        // one float per thread, every slab's value requested before the first add (one cold round trip for nT <= 64;
        // float4 columns would need 256 registers for that and halve the job workgroups' occupancy)
        const int sidx = (b - kMatWgJobs) * 256 + tid;
        if (sidx < 4 * u.n_small4) {
            const long stride = u.slab_stride;
            float acc = 0.f;
            constexpr int SB = 64;
            for (int g0 = 0; g0 < u.nT; g0 += SB) {
                float v[SB];
#pragma unroll
                for (int kk = 0; kk < SB; ++kk) v[kk] = (g0 + kk < u.nT) ? u.slabs[(long)(g0 + kk) * stride + sidx] : 0.f;
#pragma unroll
                for (int kk = 0; kk < SB; ++kk) acc += v[kk];
            }
            int seg = 0;
            for (int j = 1; j < u.n_seg; ++j) if (sidx >= u.seg_start[j]) seg = j;
            u.grads[u.seg_dst[seg] + (sidx - u.seg_start[seg])] = acc;
            q = (double)acc * acc;
        }
    }
    if (u.fuse_norm) {                                          // uniform per launch
        q = block_sum(q, s_red);
        if (tid == 0) u.norm_scratch[2 + b] = q;
    }
}

// ---- fused tail of K15's split-wgrad chain (round 4; single rank): mat_update_wgrad_kernel's jobs carried through to the
// optimiser step -- every workgroup requests the optimiser state of its elements beside its operands, forms its piece of
// the gradient (same tiles, folds and orders), publishes its squared-norm partial as a tagged record (tail_sync.hpp),
// waits for all records, and applies clip + Adam (ppoaf_adam_step_prenormed's arithmetic: mat_policy.py:677-699, ONE
// optimiser over actor + critic) to exactly those elements.  mat_update_wgrad_kernel + clip_adam_kernel (7.6 + 5.1 us at
// C5) become one launch.  Parameters, moments and the gradient bucket end bitwise as after the two launches.
struct MatAdam { float* exp_avg; float* exp_avg_sq; const float* lr; float beta1, beta2, eps, grad_scale, max_norm; float* grad_norm_out; };
struct MatPmv { float p, m, v; };
__device__ __forceinline__ MatPmv mat_pmv_load(const MatDev& u, const MatAdam& ad, const long idx, const bool ok) {
    MatPmv r = {0.f, 0.f, 0.f};
    if (ok) { r.p = u.params[idx]; r.m = ad.exp_avg[idx]; r.v = ad.exp_avg_sq[idx]; }
    return r;
}
__device__ __forceinline__ void mat_adam1(const MatDev& u, const MatAdam& ad, const long idx, const float g, const MatPmv& s,
                                          const float gs, const float step_size, const float bc2_sqrt) {
    const float gi = g * gs;                                   // clip_adam_kernel, expression for expression
    const float mi = ad.beta1 * s.m + (1.0f - ad.beta1) * gi;
    const float vi = ad.beta2 * s.v + (1.0f - ad.beta2) * gi * gi;
    ad.exp_avg[idx] = mi;
    ad.exp_avg_sq[idx] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + ad.eps;
    const_cast<float*>(u.params)[idx] = s.p - step_size * (mi / denom);
}
// wave 0 of a workgroup: publish q, wait for everybody's, coefficients into s_coef (t = the step being taken)
__device__ __forceinline__ void mat_tail_sync_wave0(const MatAdam& ad, const TailDev& td, const unsigned tag, const int b, const long long t_next,
                                                    const long long pre_t, double bc1, double bc2s, const float lr, const double q, float* s_coef) {
    if (threadIdx.x == 0) tail_publish(td, tag, b, q);
    if (pre_t != t_next) {                                     // uniform: first launch / restored state
        bc1 = 1.0 - pow((double)ad.beta1, (double)t_next);
        bc2s = sqrt(1.0 - pow((double)ad.beta2, (double)t_next));
    }
    double sq, unused;
    tail_gather(td, tag, sq, unused);
    if (threadIdx.x == 0) {
        const float total_norm = (float)sqrt(sq);
        float coef = 1.0f;
        if (ad.max_norm > 0.f) coef = fminf(ad.max_norm / (total_norm + 1e-6f), 1.0f);
        s_coef[0] = ad.grad_scale * coef;
        s_coef[1] = (float)((double)lr / bc1);
        s_coef[2] = (float)bc2s;
        if (b == 0 && ad.grad_norm_out) ad.grad_norm_out[0] = total_norm;
    }
}

__global__ __launch_bounds__(256) void mat_update_wgrad_adam_kernel(MatDev u, int n_small_blocks, MatAdam ad, TailDev td) {
    __shared__ double s_red[17];
    __shared__ __attribute__((aligned(16))) float s_fold[2 * 3 * 256 + 64];
    __shared__ float s_tile[16 * 32 + 16];
    __shared__ float s_coef[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long seq = __hip_atomic_load(&td.ctl->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned tag = ((unsigned)seq & 0x7fffffffu) + 1u;
    // read BEFORE this workgroup publishes: the bookkeeping workgroup rewrites them only after everybody has
    const long long t_next = (long long)u.step_count[0] + 1;
    const float lr = ad.lr[0];
    const long long pre_t = td.ctl->bc_t[0];
    const double pre_bc1 = td.ctl->bc[0], pre_bc2s = td.ctl->bc[1];
    if (b == td.nblk) {
        // bookkeeping: totals and cursor need nobody; the step counter and the launch tag move after all have published
        if (tid >= 64) return;
        const int keep_fuse = u.fuse_norm;
        MatDev ub = u;
        ub.fuse_norm = 0;                                      // (the step counter is advanced below, after the wait)
        mat_update_bookkeeping(ub);
        (void)keep_fuse;
        double c_next[2] = {0.0, 0.0};
        if (tid == 0) {
            c_next[0] = 1.0 - pow((double)ad.beta1, (double)(t_next + 1));
            c_next[1] = sqrt(1.0 - pow((double)ad.beta2, (double)(t_next + 1)));
        }
        double sq, unused;
        tail_gather(td, tag, sq, unused);
        if (tid == 0) {
            u.step_count[0] = t_next;
            u.norm_scratch[0] = sq;
            td.ctl->bc_t[0] = t_next + 1;
            td.ctl->bc[0] = c_next[0];
            td.ctl->bc[1] = c_next[1];
            __hip_atomic_store(&td.ctl->seq, seq + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    double q = 0.0;
    if (b < kMatWgJobs) {
        constexpr int per_xcd = kMatWgJobs / 8;
        const int job = (b & 7) * per_xcd + (b >> 3);
        const int k = job >> 3, ot = (job >> 1) & 3, ih = job & 1;
        const int wk = kMatLinW[k];
        const long plane = (long)u.R * kMD;
        // this thread's two elements of the 16 x 32 piece (row e / 32, column e % 32) and, threads 64..79 of the jobs of
        // input half 0, one bias: optimiser state first
        long eidx[2];
        MatPmv se[2], sb;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int e = tid + 256 * kk, row = e >> 5, col = e & 31;
            eidx[kk] = u.off[wk] + (long)(ot * 16 + row) * kMD + ih * 32 + col;
            se[kk] = mat_pmv_load(u, ad, eidx[kk], true);
        }
        const bool has_b = ih == 0 && tid >= 64 && tid < 80;
        const long bidx = u.off[wk + 1] + ot * 16 + (tid - 64);
        sb = mat_pmv_load(u, ad, bidx, has_b);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(u.dpanel + (long)k * plane, 0, 0xFFFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(u.xpanel + (long)kMatLinX[k] * plane, 0, 0xFFFFFFFF, 0x00020000);
        const unsigned dl = 4u * (unsigned)((lane >> 4) * kMD + ot * 16 + (lane & 15));
        const unsigned xl = 4u * (unsigned)((lane >> 4) * kMD + ih * 32 + (lane & 15));
        const int nc = u.nT;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        float bsum = 0.f;
        constexpr int MAXC = 16;
        for (int c0 = wave; c0 < nc; c0 += 4 * MAXC) {         // wave-uniform trip count
            float a[MAXC][4], x0[MAXC][4], x1[MAXC][4];
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int ch = c0 + 4 * c;
                if (ch < nc) {                                 // wave-uniform
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned so = 4u * (unsigned)((16 * ch + 4 * j) * kMD);
                        a[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, dl, so, 0));
                        x0[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xl, so, 0));
                        x1[c][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, xl + 64u, so, 0));
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                if (c0 + 4 * c < nc) {                         // wave-uniform
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], x0[c][j], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][j], x1[c][j], acc1, 0, 0, 0);
                        bsum += a[c][j];
                    }
                }
            }
        }
        if (wave > 0) {
            *reinterpret_cast<f32x4*>(s_fold + (((wave - 1) * 2 + 0) * 64 + lane) * 4) = acc0;
            *reinterpret_cast<f32x4*>(s_fold + (((wave - 1) * 2 + 1) * 64 + lane) * 4) = acc1;
        }
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (lane < 16) s_fold[1536 + wave * 16 + lane] = bsum;
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                acc0 += *reinterpret_cast<const f32x4*>(s_fold + ((w * 2 + 0) * 64 + lane) * 4);
                acc1 += *reinterpret_cast<const f32x4*>(s_fold + ((w * 2 + 1) * 64 + lane) * 4);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (lane >> 4) + r;
                s_tile[row * 32 + (lane & 15)] = acc0[r];
                s_tile[row * 32 + 16 + (lane & 15)] = acc1[r];
                q += (double)acc0[r] * acc0[r] + (double)acc1[r] * acc1[r];
            }
            if (ih == 0 && lane < 16) {
                const float bg = s_fold[1536 + lane] + s_fold[1536 + 16 + lane] + s_fold[1536 + 32 + lane] + s_fold[1536 + 48 + lane];
                s_tile[512 + lane] = bg;
                q += (double)bg * bg;
            }
            q = tail_wave_sum(q);                              // block_sum of the wgrad launch: this wave's sum + three exact zeros
            mat_tail_sync_wave0(ad, td, tag, b, t_next, pre_t, pre_bc1, pre_bc2s, lr, q, s_coef);
        }
        __syncthreads();
        const float gs = s_coef[0], step_size = s_coef[1], bc2_sqrt = s_coef[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) mat_adam1(u, ad, eidx[kk], s_tile[tid + 256 * kk], se[kk], gs, step_size, bc2_sqrt);
        if (has_b) mat_adam1(u, ad, bidx, s_tile[512 + tid - 64], sb, gs, step_size, bc2_sqrt);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) u.grads[eidx[kk]] = s_tile[tid + 256 * kk];
        if (has_b) u.grads[bidx] = s_tile[512 + tid - 64];
    } else {
        // compact slabs of the small tensors: one float per thread (mat_update_wgrad_kernel's fold), then its Adam step
        const int sidx = (b - kMatWgJobs) * 256 + tid;
        const bool live = sidx < 4 * u.n_small4;
        int seg = 0;
        for (int j = 1; j < u.n_seg; ++j) if (sidx >= u.seg_start[j]) seg = j;
        const long gidx = live ? (long)u.seg_dst[seg] + (sidx - u.seg_start[seg]) : 0;
        const MatPmv st = mat_pmv_load(u, ad, gidx, live);
        float acc = 0.f;
        if (live) {
            const long stride = u.slab_stride;
            constexpr int SB = 64;
            for (int g0 = 0; g0 < u.nT; g0 += SB) {
                float v[SB];
#pragma unroll
                for (int kk = 0; kk < SB; ++kk) v[kk] = (g0 + kk < u.nT) ? u.slabs[(long)(g0 + kk) * stride + sidx] : 0.f;
#pragma unroll
                for (int kk = 0; kk < SB; ++kk) acc += v[kk];
            }
            u.grads[gidx] = acc;
            q = (double)acc * acc;
        }
        // block_sum(q) in its own association: wave sums, then the wave sum of the four of them
        q = tail_wave_sum(q);
        if (lane == 0) s_red[wave] = q;
        __syncthreads();
        if (wave == 0) {
            q = tail_wave_sum(lane < 4 ? s_red[lane] : 0.0);
            mat_tail_sync_wave0(ad, td, tag, b, t_next, pre_t, pre_bc1, pre_bc2s, lr, q, s_coef);
        }
        __syncthreads();
        if (live) mat_adam1(u, ad, gidx, acc, st, s_coef[0], s_coef[1], s_coef[2]);
    }
}

static size_t mat_lds_bytes(int O) {
    const size_t OS = 16 * ((O + 15) / 16) + 4;
    return (256 + 4 * 128 + 4 * 272 + kRows * kMXS + 2 * kRows * OS + 34 * (size_t)kMTile) * 4;
}

static int make_mat(const ppoaf_mat_update_args_t* a, MatDev& u) {
    PPOAF_REQUIRE(a, "mat_update: null args");
    PPOAF_REQUIRE(a->embedding == kMD, "mat_update: embedding=%d (the fused kernel is built for 64)", a->embedding);
    PPOAF_REQUIRE(a->num_agents >= 1 && a->num_agents <= 16, "mat_update: num_agents=%d out of [1,16]", a->num_agents);
    PPOAF_REQUIRE(a->obs_dim >= 1 && a->obs_dim <= 64, "mat_update: obs_dim=%d out of [1,64]", a->obs_dim);
    PPOAF_REQUIRE(a->num_actions >= 1 && a->num_actions <= 8, "mat_update: num_actions=%d out of [1,8]", a->num_actions);
    PPOAF_REQUIRE(a->B >= 1 && a->batch_stride >= a->B, "mat_update: B=%ld stride=%ld", (long)a->B, (long)a->batch_stride);
    PPOAF_REQUIRE(a->params && a->grads && a->slabs && a->critic_obs && a->raw_actions && a->advantages &&
                      a->old_log_probs && a->rewards_to_go && a->values && a->perm && a->cursor && a->vn_mean &&
                      a->vn_var && a->vn_count && a->loss_partials && a->totals,
                  "mat_update: null pointer");
    PPOAF_REQUIRE(!a->normalize_values || (a->vn_records && a->n_ranks >= 1), "mat_update: vn_records missing");
    PPOAF_REQUIRE(!a->normalize_adv || a->adv_records, "mat_update: adv_records missing");
    PPOAF_REQUIRE(a->bucket_total % 4 == 0 && ((uintptr_t)a->params & 15) == 0 && ((uintptr_t)a->grads & 15) == 0 &&
                      ((uintptr_t)a->slabs & 15) == 0, "mat_update: buckets must be 16-byte aligned");
    // the offset table must describe MATActorCritic's module order with every tensor padded to 4 floats
    const long D = kMD, O = a->obs_dim, NA = a->num_actions, Ain = NA + 1;
    auto pad4 = [](long x) { return (x + 3) / 4 * 4; };
    long want[MAT_NUM_PARAMS];
    {
        long sizes[MAT_NUM_PARAMS];
        int k = 0;
        sizes[k++] = D * Ain;
        for (int i = 0; i < 8; ++i) sizes[k++] = D;                       // a.ln, ln1, ln2, ln3 (g, b)
        for (int i = 0; i < 8; ++i) { sizes[k++] = D * D; sizes[k++] = D; }   // attn1 k,q,v,proj; attn2 k,q,v,proj
        for (int i = 0; i < 2; ++i) { sizes[k++] = D * D; sizes[k++] = D; }   // mlp.0, mlp.2
        sizes[k++] = D * D; sizes[k++] = D; sizes[k++] = D; sizes[k++] = D; sizes[k++] = NA * D; sizes[k++] = NA;   // head
        sizes[k++] = O; sizes[k++] = O; sizes[k++] = D * O; sizes[k++] = D;   // obs_encoder
        for (int i = 0; i < 6; ++i) sizes[k++] = D;                       // c.ln, ln1, ln2
        for (int i = 0; i < 4; ++i) { sizes[k++] = D * D; sizes[k++] = D; }   // attn k,q,v,proj
        for (int i = 0; i < 2; ++i) { sizes[k++] = D * D; sizes[k++] = D; }   // mlp
        sizes[k++] = D * D; sizes[k++] = D; sizes[k++] = D; sizes[k++] = D; sizes[k++] = D; sizes[k++] = 1;         // head
        PPOAF_REQUIRE(k == MAT_NUM_PARAMS, "mat_update: internal size table (%d)", k);
        long off = 0;
        for (int i = 0; i < MAT_NUM_PARAMS; ++i) { want[i] = off; off += pad4(sizes[i]); }
        PPOAF_REQUIRE(off == a->bucket_total, "mat_update: bucket_total=%ld, the topology needs %ld", (long)a->bucket_total, off);
    }
    for (int i = 0; i < MAT_NUM_PARAMS; ++i) {
        PPOAF_REQUIRE(a->offsets[i] == want[i], "mat_update: parameter %d sits at %ld, expected %ld", i,
                      (long)a->offsets[i], want[i]);
        u.off[i] = a->offsets[i];
    }
    u.off[63] = 0;
    u.O = a->obs_dim; u.L = a->num_agents; u.NA = a->num_actions; u.Ain = (int)Ain;
    u.per_tile = 16 / u.L;
    u.nT = (int)((a->B + u.per_tile - 1) / u.per_tile);
    u.total = a->bucket_total;
    u.params = a->params; u.grads = a->grads; u.slabs = a->slabs;
    u.obs = a->critic_obs; u.actions = a->raw_actions; u.adv = a->advantages; u.old_lp = a->old_log_probs;
    u.rtg = a->rewards_to_go; u.values = a->values; u.perm = a->perm; u.row_map = a->row_map; u.n_rows = a->n_rows;
    u.cursor = a->cursor; u.B = a->B; u.batch_stride = a->batch_stride;
    PPOAF_REQUIRE(a->mb_offset >= 0 && a->cursor_advance >= 0, "mat_update: mb_offset=%ld cursor_advance=%ld", (long)a->mb_offset, (long)a->cursor_advance);
    u.mb_offset = a->mb_offset; u.cursor_advance = a->cursor_advance;
    u.normalize_values = a->normalize_values; u.n_ranks = a->n_ranks; u.normalize_adv = a->normalize_adv;
    u.use_huber = a->use_huber; u.vn_mean = a->vn_mean; u.vn_var = a->vn_var; u.vn_count = a->vn_count;
    u.vn_records = a->vn_records; u.adv_records = a->adv_records;
    u.surr_clip = a->surr_clip; u.entropy_weight = a->entropy_weight; u.kl_loss_weight = a->kl_loss_weight;
    u.huber_delta = a->huber_delta; u.loss_partials = a->loss_partials; u.totals = a->totals;
    PPOAF_REQUIRE(!a->fuse_norm || (a->norm_scratch && a->step_count), "mat_update: fuse_norm needs norm_scratch and step_count");
    u.norm_scratch = a->norm_scratch; u.step_count = a->step_count; u.fuse_norm = a->fuse_norm != 0;
    u.pregathered = a->inputs_in_batch_order != 0;
    // gradient-partial layout of a workgroup's slab; split-wgrad chain: panels + the compact layout of the small tensors
    u.split = 0; u.R = 16 * u.nT; u.xpanel = nullptr; u.dpanel = nullptr;
    u.n_small4 = 0; u.n_seg = 0;
    for (int i = 0; i < 4; ++i) { u.seg_start[i] = 0; u.seg_dst[i] = 0; }
    u.slab_stride = u.total;
    for (int i = 0; i < 64; ++i) u.goff[i] = i < MAT_NUM_PARAMS ? (int)u.off[i] : 0;
    if (a->split_workspace) {
        PPOAF_REQUIRE((((uintptr_t)a->split_workspace) & 255) == 0, "mat_update: split_workspace must be 256-byte aligned");
        const size_t plane = (size_t)u.R * kMD * 4;
        const size_t need = (size_t)(MAT_NUM_XPANELS + kMatLin) * plane;
        PPOAF_REQUIRE((size_t)a->split_workspace_bytes >= need, "mat_update: split_workspace of %ld B, %zu needed",
                      (long)a->split_workspace_bytes, need);
        u.split = 1;
        u.xpanel = reinterpret_cast<float*>(a->split_workspace);
        u.dpanel = u.xpanel + (size_t)MAT_NUM_XPANELS * u.R * kMD;
        bool big[MAT_NUM_PARAMS] = {};
        for (int k = 0; k < kMatLin; ++k) { big[kMatLinW_host[k]] = true; big[kMatLinW_host[k] + 1] = true; }
        long compact = 0;
        bool in_run = false;
        for (int i = 0; i < MAT_NUM_PARAMS; ++i) {
            if (big[i]) { u.goff[i] = -1; in_run = false; continue; }
            if (!in_run) {
                PPOAF_REQUIRE(u.n_seg < 4, "mat_update: internal segment table");
                u.seg_start[u.n_seg] = (int)compact; u.seg_dst[u.n_seg] = (int)u.off[i]; ++u.n_seg;
                in_run = true;
            }
            u.goff[i] = (int)compact;
            compact += (i + 1 < MAT_NUM_PARAMS ? u.off[i + 1] : u.total) - u.off[i];
        }
        u.slab_stride = compact;
        u.n_small4 = (int)(compact >> 2);
    }
    return PPOAF_OK;
}

}  // namespace ppoaf

using namespace ppoaf;

extern "C" int ppoaf_mat_update_fwd_bwd(const ppoaf_mat_update_args_t* args, ppoaf_stream_t stream) {
    return ppoaf_mat_update_fwd_bwd_timed(args, nullptr, nullptr, stream);
}

extern "C" int ppoaf_mat_update_fwd_bwd_timed(const ppoaf_mat_update_args_t* args, void* start_event, void* stop_event,
                                              ppoaf_stream_t stream) {
    hipEvent_t e0 = (hipEvent_t)start_event, e1 = (hipEvent_t)stop_event;
    MatDev u;
    const int rc = make_mat(args, u);
    if (rc) return rc;
    const size_t lds = mat_lds_bytes(u.O);
    PPOAF_REQUIRE(lds <= 160 * 1024, "mat_update: needs %zu B of LDS (> 160 KiB)", lds);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mat_update_fwd_bwd_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(mat_update_fwd_bwd_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PPOAF_E_LAUNCH; }
        attr_set = true;
    }
    const dim3 grid((unsigned)u.nT), block(kMT);
    hipStream_t st = (hipStream_t)stream;
    if (u.split) {
        if (e0 || e1) hipExtLaunchKernelGGL(mat_update_fwd_bwd_kernel<true>, grid, block, lds, st, e0, e1, 0, u);
        else hipLaunchKernelGGL(mat_update_fwd_bwd_kernel<true>, grid, block, lds, st, u);
    } else {
        if (e0 || e1) hipExtLaunchKernelGGL(mat_update_fwd_bwd_kernel<false>, grid, block, lds, st, e0, e1, 0, u);
        else hipLaunchKernelGGL(mat_update_fwd_bwd_kernel<false>, grid, block, lds, st, u);
    }
    return check_launch("mat_update_fwd_bwd");
}

extern "C" int ppoaf_mat_update_reduce(const ppoaf_mat_update_args_t* args, ppoaf_stream_t stream) {
    MatDev u;
    const int rc = make_mat(args, u);
    if (rc) return rc;
    if (u.split) {
        const int nsb = (4 * u.n_small4 + 255) / 256;
        hipLaunchKernelGGL(mat_update_wgrad_kernel, dim3((unsigned)(kMatWgJobs + nsb + 1)), dim3(256), 0, (hipStream_t)stream, u, nsb);
        return check_launch("mat_update_reduce (wgrad)");
    }
    const long n4 = u.total >> 2;
    hipLaunchKernelGGL(mat_update_reduce_kernel, dim3((unsigned)((n4 + kMatRedThreads - 1) / kMatRedThreads) + 1u),
                       dim3(kMatRedThreads), 0, (hipStream_t)stream, u);
    return check_launch("mat_update_reduce");
}

extern "C" int ppoaf_mat_update_tail_ctl_bytes(const ppoaf_mat_update_args_t* args, int64_t* bytes_out) {
    PPOAF_REQUIRE(args && bytes_out, "mat_update_tail_ctl_bytes: null argument");
    MatDev u;
    const int rc = make_mat(args, u);
    if (rc) return rc;
    PPOAF_REQUIRE(u.split, "mat_update_tail_ctl_bytes: args->split_workspace is not set");
    *bytes_out = (int64_t)kTailRecOff + 16 * (int64_t)(kMatWgJobs + (4 * u.n_small4 + 255) / 256);
    return PPOAF_OK;
}

extern "C" int ppoaf_mat_update_wgrad_adam(const ppoaf_mat_update_args_t* args, void* ctl, float* exp_avg, float* exp_avg_sq,
                                           const float* lr, float beta1, float beta2, float eps, float grad_scale, float max_norm,
                                           float* grad_norm_out, double wait_seconds, ppoaf_stream_t stream) {
    MatDev u;
    const int rc = make_mat(args, u);
    if (rc) return rc;
    PPOAF_REQUIRE(u.split, "mat_update_wgrad_adam: args->split_workspace is not set (the tail of the split-wgrad chain)");
    PPOAF_REQUIRE(u.fuse_norm, "mat_update_wgrad_adam: args->fuse_norm must be set (the launch owns the step counter)");
    PPOAF_REQUIRE(ctl && (((uintptr_t)ctl) & 63) == 0, "mat_update_wgrad_adam: control block missing or not 64-byte aligned");
    PPOAF_REQUIRE(exp_avg && exp_avg_sq && lr, "mat_update_wgrad_adam: null optimiser state");
    PPOAF_REQUIRE(wait_seconds > 0.0 && wait_seconds <= 600.0, "mat_update_wgrad_adam: wait_seconds=%g", wait_seconds);
    const int nsb = (4 * u.n_small4 + 255) / 256;
    TailDev td;
    td.ctl = reinterpret_cast<TailCtl*>(ctl);
    td.budget = (long long)(wait_seconds * 1.0e8);
    td.nblk = kMatWgJobs + nsb;
    td.jobs_a = 1 << 30; td.jobs_c = 0; td.per_xcd = 1;       // one optimiser: every record belongs to the one norm
    PPOAF_REQUIRE(td.nblk <= 64 * kTailMaxRounds, "mat_update_wgrad_adam: %d workgroups, a polling wave holds %d records", td.nblk,
                  64 * kTailMaxRounds);
    static int per_cu = 0, cus = 0;            // all workgroups wait for each other: they must fit on the device together
    if (per_cu == 0) {
        int n = 0, dev = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(mat_update_wgrad_adam_kernel), 256, 0);
        if (e != hipSuccess) { set_error("mat_update_wgrad_adam: occupancy query: %s", hipGetErrorString(e)); return PPOAF_E_LAUNCH; }
        per_cu = n > 0 ? n : -1;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    }
    PPOAF_REQUIRE(per_cu > 0 && (long)(td.nblk + 1) <= (long)per_cu * cus,
                  "mat_update_wgrad_adam: %d workgroups cannot be resident together (%d per CU x %d CUs)", td.nblk + 1, per_cu, cus);
    MatAdam ad{exp_avg, exp_avg_sq, lr, beta1, beta2, eps, grad_scale, max_norm, grad_norm_out};
    hipLaunchKernelGGL(mat_update_wgrad_adam_kernel, dim3((unsigned)(td.nblk + 1)), dim3(256), 0, (hipStream_t)stream, u, nsb, ad, td);
    return check_launch("mat_update_wgrad_adam");
}

extern "C" int ppoaf_mat_update_split_workspace_bytes(const ppoaf_mat_update_args_t* args, int64_t* bytes_out) {
    PPOAF_REQUIRE(args && bytes_out, "mat_update_split_workspace_bytes: null argument");
    ppoaf_mat_update_args_t a = *args;
    a.split_workspace = nullptr; a.split_workspace_bytes = 0;
    MatDev u;
    const int rc = make_mat(&a, u);
    if (rc) return rc;
    *bytes_out = (int64_t)((size_t)(MAT_NUM_XPANELS + kMatLin) * u.R * kMD * 4);
    return PPOAF_OK;
}

extern "C" int ppoaf_mat_update_norm_partials(const ppoaf_mat_update_args_t* args) {
    MatDev u;
    if (make_mat(args, u)) return -1;
    if (u.split) return kMatWgJobs + (4 * u.n_small4 + 255) / 256;
    return (int)(((u.total >> 2) + kMatRedThreads - 1) / kMatRedThreads);
}

extern "C" int ppoaf_mat_policy_step(const ppoaf_mat_step_args_t* a, ppoaf_stream_t stream) {
    PPOAF_REQUIRE(a, "mat_policy_step: null args");
    PPOAF_REQUIRE(a->embedding == kMD, "mat_policy_step: embedding=%d (built for 64)", a->embedding);
    PPOAF_REQUIRE(a->num_agents >= 1 && a->num_agents <= 16 && a->obs_dim >= 1 && a->obs_dim <= 64 &&
                      a->num_actions >= 1 && a->num_actions <= 8 && a->actor_obs_dim >= 1,
                  "mat_policy_step: sizes (agents %d, obs %d, actions %d)", a->num_agents, a->obs_dim, a->num_actions);
    PPOAF_REQUIRE(a->E >= 1, "mat_policy_step: E=%ld", (long)a->E);
    PPOAF_REQUIRE(a->params && a->critic_obs && a->action_out && a->logp_out && a->value_out, "mat_policy_step: null pointer");
    PPOAF_REQUIRE(!a->normalize_values || (a->vn_mean && a->vn_var), "mat_policy_step: value normaliser statistics missing");
    PPOAF_REQUIRE(((uintptr_t)a->params & 15) == 0, "mat_policy_step: params must be 16-byte aligned");
    MatStepDev u;
    for (int i = 0; i < 64; ++i) u.off[i] = a->offsets[i];
    u.O = a->obs_dim; u.L = a->num_agents; u.NA = a->num_actions; u.Ain = a->num_actions + 1;
    u.per_tile = 16 / u.L; u.Oa = a->actor_obs_dim;
    u.params = a->params; u.obs = a->critic_obs; u.actor_obs = a->actor_obs; u.E = a->E;
    u.seed = a->seed; u.offset = a->offset;
    u.normalize_values = a->normalize_values; u.vn_mean = a->vn_mean; u.vn_var = a->vn_var;
    u.action_out = a->action_out; u.raw_action_out = a->raw_action_out; u.logp_out = a->logp_out;
    u.value_out = a->value_out; u.critic_obs_out = a->critic_obs_copy_out; u.obs_out = a->obs_copy_out;
    u.forced_action = a->forced_action;
    const size_t lds = mat_lds_bytes(u.O);
    PPOAF_REQUIRE(lds <= 160 * 1024, "mat_policy_step: needs %zu B of LDS (> 160 KiB)", lds);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mat_policy_step_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PPOAF_E_LAUNCH; }
        attr_set = true;
    }
    const long n_wg = (a->E + u.per_tile - 1) / u.per_tile;
    PPOAF_REQUIRE(n_wg <= 0x7fffffffL, "mat_policy_step: too many envs");
    hipLaunchKernelGGL(mat_policy_step_kernel, dim3((unsigned)n_wg), dim3(kMT), lds, (hipStream_t)stream, u);
    return check_launch("mat_policy_step");
}

// mpi_avg_gradients at the MAT update's per-mini-batch call site (mat_policy.py:692) when the K17 peer exchange is not
// available: fwd_bwd -> reduce -> RCCL sum all-reduce of the shared bucket -> K11 clip + Adam (mat_policy.py:694-699),
// for n consecutive mini-batches issued from this one call.
extern "C" int ppoaf_mat_update_chain_allreduce(const ppoaf_mat_update_args_t* args, ppoaf_comm_t* comm, int64_t n_minibatches,
                                                float* exp_avg, float* exp_avg_sq, const float* lr, float beta1, float beta2,
                                                float eps, float grad_scale, float max_norm, float* grad_norm_out,
                                                ppoaf_stream_t stream) {
    PPOAF_REQUIRE(args && comm && exp_avg && exp_avg_sq && lr, "mat_update_chain_allreduce: null argument");
    PPOAF_REQUIRE(n_minibatches >= 1 && n_minibatches <= (1 << 20), "mat_update_chain_allreduce: n_minibatches=%ld", (long)n_minibatches);
    PPOAF_REQUIRE(args->fuse_norm == 0 && args->norm_scratch && args->step_count,
                  "mat_update_chain_allreduce: fuse_norm must be 0 (the norm is that of the all-reduced gradient)");
    for (int64_t j = 0; j < n_minibatches; ++j) {
        int rc = ppoaf_mat_update_fwd_bwd(args, stream);
        if (rc == PPOAF_OK) rc = ppoaf_mat_update_reduce(args, stream);
        if (rc == PPOAF_OK) rc = ppoaf_allreduce_sum_f32(comm, args->grads, args->bucket_total, stream);
        if (rc == PPOAF_OK)
            rc = ppoaf_clip_adam_step(const_cast<float*>(args->params), args->grads, exp_avg, exp_avg_sq, args->bucket_total,
                                      args->step_count, lr, beta1, beta2, eps, grad_scale, max_norm, args->norm_scratch,
                                      grad_norm_out, stream);
        if (rc != PPOAF_OK) return rc;
    }
    return PPOAF_OK;
}

#ifdef PPOAF_MAT_STAMPS
extern "C" int ppoaf_debug_read_mat_stamps(unsigned long long* out /* host [64] */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ppoaf::g_mat_stamps), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -2;
}
#endif
