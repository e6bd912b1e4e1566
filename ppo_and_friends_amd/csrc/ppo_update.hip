// K12: fused PPO mini-batch update for MLP actor / critic networks.
// One iteration of PPO._ppo_batch_train (ppo.py:2292-2469) =
//   ppo_update_fwd_bwd_kernel  (gather + normalisers + forward + head/loss + backward -> slabs)
//   ppo_update_reduce_kernel   (slabs -> gradient bucket, loss partials -> totals [, norms])
//   ppo_update_adam_kernel     (clip + Adam for both networks, advance the cursor)
//
// Work decomposition (MI355X): a mini-batch is B = 256 rows of a 4->128->128->128->k MLP pair,
// ~100 MFLOP: one CU would need ~170 us at the f32 MFMA rate, so the rows are split 16 per
// workgroup and the two networks run in separate workgroups: grid (B/16, 2).  Each workgroup
// streams the weights from L2 (271 KB, shared by all), keeps its 16 rows' activations in LDS,
// runs every HxH layer on v_mfma_f32_16x16x4_f32 (exact fmaf chains: float32 parity), and
// writes its weight-gradient partial to a private slab with plain coalesced stores; the slabs
// are summed in a fixed order by the reduce kernel, so results are bitwise reproducible (float
// atomics would not be) and nothing needs zeroing.
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

struct UpdateDev {
    NetDev net[2];
    const float* params; float* grads; float* exp_avg; float* exp_avg_sq; float* slabs;
    long bucket_total;
    int64_t* step_counts; const float* lr; double* norm_scratch;
    float beta1, beta2, adam_eps, grad_scale, max_norm; int head_kind;
    const float* obs; const float* critic_obs; const void* raw_actions;
    const float* adv; const float* old_lp; const float* rtg; float* values;
    const int64_t* perm; const int32_t* row_map; long n_rows;
    int64_t* cursor; long B, batch_stride;
    int normalize_values, n_ranks;
    float* vn_mean; float* vn_var; double* vn_count; const double* vn_records;
    int normalize_adv, use_huber;
    float surr_clip, entropy_weight, kl_loss_weight, huber_delta, min_std;
    float* loss_partials; double* totals;
    int n_wg;
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

__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
    return v;
}

__device__ __forceinline__ float clamp_prob_u(float n) {
    return fminf(fmaxf(n, FLT_EPSILON), 1.0f - FLT_EPSILON);
}
__device__ __forceinline__ float softplus_u(float x) { return x > 20.f ? x : log1pf(expf(x)); }

extern __shared__ __attribute__((aligned(16))) unsigned char ppo_update_smem[];

__global__ __launch_bounds__(kThreads) void ppo_update_fwd_bwd_kernel(UpdateDev u) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, which = blockIdx.y;          // which: 0 actor, 1 critic
    const NetDev& nd = u.net[which];
    const int H = nd.H, HS = H + 4, in_dim = nd.in_dim, INP = ((in_dim + 3) & ~3) + 4;
    const int depth = nd.depth, out_dim = nd.out_dim, act = nd.act;
    const float* P = u.params + nd.offset;
    float* slab = u.slabs + (long)g * u.bucket_total + nd.offset;
    const long B = u.B;
    const long mb = u.cursor[0];
    const long base = mb * u.batch_stride;

    // ---- LDS carve (all offsets multiples of 16 B)
    float* smem = reinterpret_cast<float*>(ppo_update_smem);
    int* sRow = reinterpret_cast<int*>(smem);                 // [16]
    float* sMisc = smem + 16;                                 // [16]: adv mean, adv std, vn mean, vn var
    double* sRed = reinterpret_cast<double*>(smem + 32);      // [17] doubles -> 34 floats, round to 48
    float* sX = smem + 80;                                    // [16, INP]
    float* sH = sX + kRows * INP;                             // depth x [16, HS]
    float* sD0 = sH + (long)depth * kRows * HS;               // [16, HS]
    float* sD1 = sD0 + kRows * HS;                            // [16, HS]
    float* sOut = sD1 + kRows * HS;                           // [16, 16]
    float* sDOut = sOut + kRows * kMaxOut;                    // [16, 16]
    float* sAdv = sDOut + kRows * kMaxOut;                    // [min(B, kMaxAdvLds)] (actor only)

    if (g == 0 && which == 0 && tid == 0) { u.norm_scratch[0] = 0.0; u.norm_scratch[1] = 0.0; }

    // ---- P0: rows of this workgroup, mini-batch statistics
    if (tid < kRows) {
        const long s = (long)g * kRows + tid;
        int row = -1;
        if (s < B) {
            long p = u.perm[base + s];
            if (p >= 0 && p < u.n_rows) row = u.row_map ? u.row_map[p] : (int)p;
        }
        sRow[tid] = row;
    }
    if (which == 0) {
        float mean_f = 0.f, std_f = 1.f;
        if (u.normalize_adv) {
            double s = 0.0;
            for (long i = tid; i < B; i += kThreads) {
                const long p = u.perm[base + i];
                const float a = u.adv[u.row_map ? u.row_map[p] : p];
                if (i < kMaxAdvLds) sAdv[i] = a;
                s += (double)a;
            }
            const double mean = block_sum(s, sRed) / (double)B;
            double q = 0.0;
            for (long i = tid; i < B; i += kThreads) {
                float a;
                if (i < kMaxAdvLds) a = sAdv[i];
                else { const long p = u.perm[base + i]; a = u.adv[u.row_map ? u.row_map[p] : p]; }
                const double d = (double)a - mean;
                q += d * d;
            }
            q = block_sum(q, sRed);
            mean_f = (float)mean;
            std_f = (float)sqrt(q / (double)(B - 1));
        }
        if (tid == 0) { sMisc[0] = mean_f; sMisc[1] = std_f; }
    } else {
        if (tid == 0) {
            const int slot = (int)(mb & 1);
            float m = u.vn_mean[slot], v = u.vn_var[slot];
            double cnt = u.vn_count[slot];
            if (u.normalize_values) {
                // Chan merge of the R per-rank records of this mini-batch, then the reference's
                // integrate (utils/stats.py:73-94) -- same arithmetic as running_moments_integrate_kernel.
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
                    const double m_2 = (double)v * cnt + (double)batch_var * n +
                                       (double)(delta * delta) * cnt * n / (cnt + n);
                    m = new_mean; v = (float)(m_2 / (cnt + n)); cnt = new_count;
                }
            }
            sMisc[2] = m; sMisc[3] = v;
            if (g == 0) { u.vn_mean[slot ^ 1] = m; u.vn_var[slot ^ 1] = v; u.vn_count[slot ^ 1] = cnt; }
        }
    }
    __syncthreads();

    // ---- P1: gather the input rows (K4)
    {
        const float* src = which == 0 ? u.obs : u.critic_obs;
        for (int idx = tid; idx < kRows * in_dim; idx += kThreads) {
            const int s = idx / in_dim, i = idx - s * in_dim;
            const int row = sRow[s];
            sX[s * INP + i] = row >= 0 ? src[(long)row * in_dim + i] : 0.f;
        }
    }
    __syncthreads();

    // ---- P2: first layer (in_dim is small: VALU)
    {
        const float* W = P + nd.offW[0];
        const float* bb = P + nd.offB[0];
        const int s = tid >> 4, og = tid & 15;
        for (int o = og; o < H; o += 16) {
            float acc = bb[o];
            const float* w = W + (long)o * in_dim;
            for (int i = 0; i < in_dim; ++i) acc = fmaf(sX[s * INP + i], w[i], acc);
            sH[s * HS + o] = act_fwd(acc, act);
        }
    }
    __syncthreads();

    // ---- P3: hidden layers on f32 MFMA: Z[16,H] = Hprev[16,H] . W^T + b
    for (int l = 1; l < depth; ++l) {
        const float* W = P + nd.offW[l];
        const float* bb = P + nd.offB[l];
        const float* Hp = sH + (long)(l - 1) * kRows * HS;
        float* Hc = sH + (long)l * kRows * HS;
        for (int nt = wave; nt < H / 16; nt += kThreads / 64) {
            const int o = nt * 16 + (lane & 15);
            const float bv = bb[o];
            f32x4 acc = {bv, bv, bv, bv};
            const float* arow = Hp + (lane & 15) * HS + 4 * (lane >> 4);
            const float* wrow = W + (long)o * H + 4 * (lane >> 4);
#pragma unroll 4
            for (int c = 0; c < H / 16; ++c) {
                const float4 a4 = *reinterpret_cast<const float4*>(arow + 16 * c);
                const float4 b4 = *reinterpret_cast<const float4*>(wrow + 16 * c);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) Hc[(4 * (lane >> 4) + r) * HS + o] = act_fwd(acc[r], act);
        }
        __syncthreads();
    }
    const float* Hlast = sH + (long)(depth - 1) * kRows * HS;

    // ---- P4: output layer (out_dim <= 16: VALU + 16-lane reductions)
    {
        const float* W = P + nd.offW[depth];
        const float* bb = P + nd.offB[depth];
        const int s = tid >> 4, part = tid & 15;
        for (int k = 0; k < out_dim; ++k) {
            float acc = 0.f;
            for (int i = part; i < H; i += 16) acc = fmaf(Hlast[s * HS + i], W[(long)k * H + i], acc);
            acc = group16_sum(acc);
            if (part == 0) sOut[s * kMaxOut + k] = acc + bb[k];
        }
    }
    __syncthreads();

    // ---- P5: distribution head + loss terms for this workgroup's rows (K6 + K3)
    if (wave == 0) {
        const int s = lane;                       // lanes 0..15 hold one row each
        const int row = s < kRows ? sRow[s] : -1;
        const bool live = row >= 0;
        const float inv_B = 1.0f / (float)B;
        float part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (which == 0) {
            float logp = 0.f, ent = 0.f;
            if (live) {
                if (u.head_kind == PPOAF_HEAD_CATEGORICAL) {
                    float p[kMaxOut];
                    float m = -INFINITY;
                    for (int k = 0; k < out_dim; ++k) m = fmaxf(m, sOut[s * kMaxOut + k]);
                    float ssum = 0.f;
                    for (int k = 0; k < out_dim; ++k) { p[k] = expf(sOut[s * kMaxOut + k] - m); ssum += p[k]; }
                    const float inv = 1.0f / ssum;
                    float s2 = 0.f;
                    for (int k = 0; k < out_dim; ++k) { p[k] *= inv; s2 += p[k]; }
                    long a = reinterpret_cast<const int64_t*>(u.raw_actions)[row];
                    a = a < 0 ? 0 : (a >= out_dim ? out_dim - 1 : a);
                    for (int k = 0; k < out_dim; ++k) { const float nk = p[k] / s2; ent -= nk * logf(clamp_prob_u(nk)); }
                    logp = logf(clamp_prob_u(p[a] / s2));
                    // loss terms
                    float av = u.adv[row];
                    if (u.normalize_adv) av = (av - sMisc[0]) / (sMisc[1] + 1e-8f);
                    const float lpo = u.old_lp[row];
                    const float ratio = expf(logp - lpo);
                    if (isnan(ratio) || isinf(ratio)) part[7] = 1.f;
                    const float lo = 1.0f - u.surr_clip, hi = 1.0f + u.surr_clip;
                    const float surr1 = ratio * av, surr2 = fminf(fmaxf(ratio, lo), hi) * av;
                    part[0] = -fminf(surr1, surr2);
                    part[3] = ent;
                    part[4] = lpo - logp;
                    float glp;
                    if (surr1 <= surr2) glp = -av * ratio;
                    else glp = (ratio >= lo && ratio <= hi) ? -av * ratio : 0.f;
                    glp *= inv_B;
                    const float gH = (u.entropy_weight != 0.0f) ? -u.entropy_weight * inv_B : 0.f;
                    // chain: z -softmax-> p -(/sum)-> n -clamp,log-> l
                    float gn[kMaxOut];
                    float dot = 0.f;
                    for (int k = 0; k < out_dim; ++k) {
                        const float nk = p[k] / s2, ck = clamp_prob_u(nk);
                        const float in_range = (nk >= FLT_EPSILON && nk <= 1.0f - FLT_EPSILON) ? 1.f : 0.f;
                        float gk = gH * (-logf(ck) - nk * in_range / ck);
                        if (k == a) gk += glp * in_range / ck;
                        gn[k] = gk; dot += gk * nk;
                    }
                    float dot2 = 0.f;
                    for (int k = 0; k < out_dim; ++k) { gn[k] = (gn[k] - dot) / s2; dot2 += gn[k] * p[k]; }
                    for (int k = 0; k < out_dim; ++k) sDOut[s * kMaxOut + k] = p[k] * (gn[k] - dot2);
                } else {
                    // tanh-Gaussian (distributions.py:441-694)
                    const float* log_std = P + nd.log_std_off;
                    const float* x = reinterpret_cast<const float*>(u.raw_actions) + (long)row * out_dim;
                    float lp = 0.f, slog = 0.f;
                    for (int d = 0; d < out_dim; ++d) {
                        const float sd = fmaxf(softplus_u(log_std[d]), u.min_std);
                        const float zz = x[d] - sOut[s * kMaxOut + d];
                        float l = -(zz * zz) / (2.0f * sd * sd) - logf(sd) - 0.91893853320467274178f;
                        l = fminf(fmaxf(l, -100.f), 100.f);
                        lp += l;
                        const float th = tanhf(x[d]);
                        slog += logf(fmaxf(1.0f - th * th, 1e-6f));
                    }
                    logp = lp - slog; ent = -logp;
                    float av = u.adv[row];
                    if (u.normalize_adv) av = (av - sMisc[0]) / (sMisc[1] + 1e-8f);
                    const float lpo = u.old_lp[row];
                    const float ratio = expf(logp - lpo);
                    if (isnan(ratio) || isinf(ratio)) part[7] = 1.f;
                    const float lo = 1.0f - u.surr_clip, hi = 1.0f + u.surr_clip;
                    const float surr1 = ratio * av, surr2 = fminf(fmaxf(ratio, lo), hi) * av;
                    part[0] = -fminf(surr1, surr2);
                    part[3] = ent;
                    part[4] = lpo - logp;
                    float glp;
                    if (surr1 <= surr2) glp = -av * ratio;
                    else glp = (ratio >= lo && ratio <= hi) ? -av * ratio : 0.f;
                    glp *= inv_B;
                    const float gH = (u.entropy_weight != 0.0f) ? -u.entropy_weight * inv_B : 0.f;
                    const float gg = glp - gH;                         // entropy = -logp
                    for (int d = 0; d < out_dim; ++d) {
                        const float ls = log_std[d];
                        const float sp = softplus_u(ls), sd = fmaxf(sp, u.min_std);
                        const float zz = x[d] - sOut[s * kMaxOut + d];
                        const float l = -(zz * zz) / (2.0f * sd * sd) - logf(sd) - 0.91893853320467274178f;
                        const float pass = (l >= -100.f && l <= 100.f) ? 1.f : 0.f;
                        sDOut[s * kMaxOut + d] = gg * pass * zz / (sd * sd);
                        const float dmax = sp > u.min_std ? 1.f : (sp == u.min_std ? 0.5f : 0.f);
                        const float dsp = ls > 20.f ? 1.f : 1.0f / (1.0f + expf(-ls));
                        // per-row d logp / d log_std, parked in sOut's upper half for the reduction below
                        sOut[s * kMaxOut + 8 + d] = gg * pass * (zz * zz / (sd * sd * sd) - 1.0f / sd) * dmax * dsp;
                    }
                }
            } else if (s < kRows) {
                for (int k = 0; k < kMaxOut; ++k) sDOut[s * kMaxOut + k] = 0.f;
                if (u.head_kind == PPOAF_HEAD_GAUSSIAN)
                    for (int d = 0; d < 8; ++d) sOut[s * kMaxOut + 8 + d] = 0.f;
            }
        } else {
            if (live) {
                const float v = sOut[s * kMaxOut];
                float rt = u.rtg[row];
                if (u.normalize_values) rt = (rt - sMisc[2]) / sqrtf(sMisc[3] + 1e-8f);
                const float diff = v - rt;
                float l, dl;
                if (u.use_huber) {
                    const float ad = fabsf(diff);
                    if (ad < u.huber_delta) { l = 0.5f * diff * diff; dl = diff; }
                    else { l = u.huber_delta * (ad - 0.5f * u.huber_delta); dl = diff > 0.f ? u.huber_delta : -u.huber_delta; }
                } else { l = diff * diff; dl = 2.0f * diff; }
                part[2] = l;
                sDOut[s * kMaxOut] = dl * inv_B;
                u.values[row] = v;                                   // ppo.py:2340
            } else if (s < kRows) {
                sDOut[s * kMaxOut] = 0.f;
            }
        }
        // per-workgroup partial sums (lanes >= 16 contribute zeros)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = (lane < kRows) ? part[k] : 0.f;
            v = group16_sum(v);
            part[k] = v;
        }
        if (lane == 0) {
            if (which == 0 && g == 0) { part[5] = sMisc[0]; part[6] = sMisc[1]; }
            float* lp = u.loss_partials + ((long)which * u.n_wg + g) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) lp[k] = part[k];
        }
    }
    __syncthreads();

    // ---- P6: output layer backward
    {
        const float* W = P + nd.offW[depth];
        // dW_out[k][i], db_out[k]
        for (int i = tid; i < H; i += kThreads) {
            for (int k = 0; k < out_dim; ++k) {
                float acc = 0.f;
#pragma unroll
                for (int s = 0; s < kRows; ++s) acc = fmaf(sDOut[s * kMaxOut + k], Hlast[s * HS + i], acc);
                slab[nd.offW[depth] + (long)k * H + i] = acc;
            }
        }
        if (tid < out_dim) {
            float acc = 0.f;
            for (int s = 0; s < kRows; ++s) acc += sDOut[s * kMaxOut + tid];
            slab[nd.offB[depth] + tid] = acc;
        }
        if (which == 0 && u.head_kind == PPOAF_HEAD_GAUSSIAN && tid >= 64 && tid < 64 + out_dim) {
            const int d = tid - 64;
            float acc = 0.f;
            for (int s = 0; s < kRows; ++s) acc += sOut[s * kMaxOut + 8 + d];
            slab[nd.log_std_off + d] = acc;
        }
        // dz_last = (dOut . W_out) * act'(Hlast)
        const int s = tid >> 4, ig = tid & 15;
        for (int i = ig; i < H; i += 16) {
            float acc = 0.f;
            for (int k = 0; k < out_dim; ++k) acc = fmaf(sDOut[s * kMaxOut + k], W[(long)k * H + i], acc);
            sD0[s * HS + i] = acc * act_bwd(Hlast[s * HS + i], act);
        }
    }
    __syncthreads();

    // ---- P7: hidden layers backward (wgrad + dgrad on MFMA)
    float* Dc = sD0;
    float* Dn = sD1;
    for (int l = depth - 1; l >= 1; --l) {
        const float* W = P + nd.offW[l];
        const float* Hin = sH + (long)(l - 1) * kRows * HS;
        // wgrad: dW[o][i] = sum_s dz[s][o] * Hin[s][i]   (M = o, N = i, K = s = 16)
        for (int mt = wave; mt < H / 16; mt += kThreads / 64) {
            const int m0 = mt * 16;
            float a[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = Dc[(4 * (lane >> 4) + j) * HS + m0 + (lane & 15)];
            for (int nt = 0; nt < H / 16; ++nt) {
                const int n0 = nt * 16;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float bj = Hin[(4 * (lane >> 4) + j) * HS + n0 + (lane & 15)];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bj, acc, 0, 0, 0);
                }
                float* dst = slab + nd.offW[l] + (long)(m0 + 4 * (lane >> 4)) * H + n0 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(long)r * H] = acc[r];
            }
        }
        for (int o = tid; o < H; o += kThreads) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < kRows; ++s) acc += Dc[s * HS + o];
            slab[nd.offB[l] + o] = acc;
        }
        // dgrad: dh[s][i] = sum_o dz[s][o] * W[o][i]     (M = s, N = i, K = o)
        for (int nt = wave; nt < H / 16; nt += kThreads / 64) {
            const int n0 = nt * 16;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* arow = Dc + (lane & 15) * HS + 4 * (lane >> 4);
            const float* wcol = W + (long)(4 * (lane >> 4)) * H + n0 + (lane & 15);
#pragma unroll 2
            for (int c = 0; c < H / 16; ++c) {
                const float4 a4 = *reinterpret_cast<const float4*>(arow + 16 * c);
                const float* wp = wcol + (long)(16 * c) * H;
                const float b0 = wp[0], b1 = wp[H], b2 = wp[2 * (long)H], b3 = wp[3 * (long)H];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b3, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = 4 * (lane >> 4) + r, i = n0 + (lane & 15);
                Dn[s * HS + i] = acc[r] * act_bwd(Hin[s * HS + i], act);
            }
        }
        __syncthreads();
        float* t = Dc; Dc = Dn; Dn = t;
    }

    // ---- P8: first layer backward (VALU)
    {
        for (int idx = tid; idx < H * in_dim; idx += kThreads) {
            const int o = idx / in_dim, i = idx - o * in_dim;
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < kRows; ++s) acc = fmaf(Dc[s * HS + o], sX[s * INP + i], acc);
            slab[nd.offW[0] + idx] = acc;
        }
        for (int o = tid; o < H; o += kThreads) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < kRows; ++s) acc += Dc[s * HS + o];
            slab[nd.offB[0] + o] = acc;
        }
    }
}

// slabs -> gradient bucket in a fixed order; block 0 folds the loss partials into the totals and
// advances the Adam step counters.
__global__ __launch_bounds__(256) void ppo_update_reduce_kernel(UpdateDev u, int compute_norms) {
    __shared__ double red[17];
    const long n4 = u.bucket_total >> 2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double q0 = 0.0, q1 = 0.0;
    if (idx < n4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* sl = reinterpret_cast<const float4*>(u.slabs) + idx;
        for (int g = 0; g < u.n_wg; ++g) {
            const float4 v = sl[(long)g * n4];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        reinterpret_cast<float4*>(u.grads)[idx] = acc;
        if (compute_norms) {
            const float sc = u.grad_scale;
            const double q = (double)(acc.x * sc) * (acc.x * sc) + (double)(acc.y * sc) * (acc.y * sc) +
                             (double)(acc.z * sc) * (acc.z * sc) + (double)(acc.w * sc) * (acc.w * sc);
            if (idx * 4 < u.net[0].size) q0 = q; else q1 = q;
        }
    }
    if (compute_norms) {
        q0 = block_sum(q0, red);
        q1 = block_sum(q1, red);
        if (threadIdx.x == 0) {
            if (q0 != 0.0) atomicAdd(&u.norm_scratch[0], q0);
            if (q1 != 0.0) atomicAdd(&u.norm_scratch[1], q1);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int g = 0; g < u.n_wg; ++g) {
            const float* a = u.loss_partials + (long)g * 8;
            const float* c = u.loss_partials + ((long)u.n_wg + g) * 8;
            p[0] += a[0]; p[3] += a[3]; p[4] += a[4]; p[7] += a[7]; p[2] += c[2];
        }
        const float n = (float)u.B;
        const float surr = p[0] / n, ent = p[3] / n, kl = p[4] / n, crit = p[2] / n;
        float total = surr;
        if (u.entropy_weight != 0.0f) total -= u.entropy_weight * ent;
        if (u.kl_loss_weight > 0.0f) total += u.kl_loss_weight * kl;
        u.totals[0] += (double)surr; u.totals[1] += (double)total; u.totals[2] += (double)crit;
        u.totals[3] += (double)ent; u.totals[4] += (double)kl;
        u.totals[5] += (double)u.loss_partials[5]; u.totals[6] += (double)u.loss_partials[6];
        u.totals[7] += p[7] > 0.f ? 1.0 : 0.0;
        u.totals[8] += 1.0;
        u.step_counts[0] += 1; u.step_counts[1] += 1;
    }
}

__global__ __launch_bounds__(256) void ppo_update_sqnorm_kernel(UpdateDev u) {
    __shared__ double red[17];
    const long n4 = u.bucket_total >> 2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    double q0 = 0.0, q1 = 0.0;
    if (idx < n4) {
        const float4 v = reinterpret_cast<const float4*>(u.grads)[idx];
        const float sc = u.grad_scale;
        const double q = (double)(v.x * sc) * (v.x * sc) + (double)(v.y * sc) * (v.y * sc) +
                         (double)(v.z * sc) * (v.z * sc) + (double)(v.w * sc) * (v.w * sc);
        if (idx * 4 < u.net[0].size) q0 = q; else q1 = q;
    }
    q0 = block_sum(q0, red);
    q1 = block_sum(q1, red);
    if (threadIdx.x == 0) {
        if (q0 != 0.0) atomicAdd(&u.norm_scratch[0], q0);
        if (q1 != 0.0) atomicAdd(&u.norm_scratch[1], q1);
    }
}

__global__ __launch_bounds__(256) void ppo_update_adam_kernel(UpdateDev u) {
    const long n4 = u.bucket_total >> 2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n4) {
        const int which = (idx * 4 < u.net[0].size) ? 0 : 1;
        const float total_norm = (float)sqrt(u.norm_scratch[which]);
        float coef = 1.0f;
        if (u.max_norm > 0.f) coef = fminf(u.max_norm / (total_norm + 1e-6f), 1.0f);
        const float gs = u.grad_scale * coef;
        const double t = (double)u.step_counts[which];
        const float lr = u.lr[0];
        const double bc1 = 1.0 - pow((double)u.beta1, t);
        const double bc2 = 1.0 - pow((double)u.beta2, t);
        const float step_size = (float)((double)lr / bc1);
        const float bc2_sqrt = (float)sqrt(bc2);
        float4 p = reinterpret_cast<float4*>(const_cast<float*>(u.params))[idx];
        const float4 g = reinterpret_cast<const float4*>(u.grads)[idx];
        float4 m = reinterpret_cast<float4*>(u.exp_avg)[idx];
        float4 v = reinterpret_cast<float4*>(u.exp_avg_sq)[idx];
#define PPOAF_ADAM1(c)                                                   \
        {                                                                \
            const float gi = g.c * gs;                                   \
            m.c = u.beta1 * m.c + (1.0f - u.beta1) * gi;                 \
            v.c = u.beta2 * v.c + (1.0f - u.beta2) * gi * gi;            \
            p.c = p.c - step_size * (m.c / (sqrtf(v.c) / bc2_sqrt + u.adam_eps)); \
        }
        PPOAF_ADAM1(x) PPOAF_ADAM1(y) PPOAF_ADAM1(z) PPOAF_ADAM1(w)
#undef PPOAF_ADAM1
        reinterpret_cast<float4*>(const_cast<float*>(u.params))[idx] = p;
        reinterpret_cast<float4*>(u.exp_avg)[idx] = m;
        reinterpret_cast<float4*>(u.exp_avg_sq)[idx] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) u.cursor[0] += 1;
}

// (n, mean, M2) of every mini-batch's rewards-to-go: one workgroup per mini-batch
__global__ __launch_bounds__(256) void minibatch_moments_kernel(const float* __restrict__ data,
                                                                const int64_t* __restrict__ perm,
                                                                const int32_t* __restrict__ row_map,
                                                                long n_perm, long B,
                                                                double* __restrict__ records) {
    __shared__ double red[17];
    const long k = blockIdx.x;
    const long base = k * B;
    const long n = (n_perm - base) < B ? (n_perm - base) : B;
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const long p = perm[base + i];
        s += (double)data[row_map ? row_map[p] : p];
    }
    const double mean = block_sum(s, red) / (double)n;
    double q = 0.0;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
        const long p = perm[base + i];
        const double d = (double)data[row_map ? row_map[p] : p] - mean;
        q += d * d;
    }
    q = block_sum(q, red);
    if (threadIdx.x == 0) { records[k * 3] = (double)n; records[k * 3 + 1] = mean; records[k * 3 + 2] = q; }
}

static int fill_net(const ppoaf_mlp_desc_t& d, NetDev& n, const char* what) {
    PPOAF_REQUIRE(d.hidden >= 16 && d.hidden <= 256 && d.hidden % 16 == 0,
                  "%s: hidden=%d must be a multiple of 16 in [16,256]", what, d.hidden);
    PPOAF_REQUIRE(d.depth >= 1 && d.depth + 1 <= kMaxLayers, "%s: depth=%d out of [1,%d]", what, d.depth, kMaxLayers - 1);
    PPOAF_REQUIRE(d.in_dim >= 1 && d.in_dim <= 1024, "%s: in_dim=%d", what, d.in_dim);
    PPOAF_REQUIRE(d.out_dim >= 1 && d.out_dim <= 8, "%s: out_dim=%d out of [1,8]", what, d.out_dim);
    PPOAF_REQUIRE(d.activation >= 0 && d.activation <= 2, "%s: activation=%d", what, d.activation);
    PPOAF_REQUIRE(d.offset % 4 == 0, "%s: bucket offset must be 16-byte aligned", what);
    n.in_dim = d.in_dim; n.H = d.hidden; n.depth = d.depth; n.out_dim = d.out_dim; n.act = d.activation;
    n.offset = d.offset; n.size = d.size; n.log_std_off = d.log_std_offset;
    auto pad4 = [](long x) { return (x + 3) / 4 * 4; };
    long off = 0;
    for (int l = 0; l <= d.depth; ++l) {
        const long in = (l == 0) ? d.in_dim : d.hidden;
        const long out = (l == d.depth) ? d.out_dim : d.hidden;
        n.offW[l] = off; off += pad4(in * out);
        n.offB[l] = off; off += pad4(out);
    }
    for (int l = d.depth + 1; l < kMaxLayers; ++l) { n.offW[l] = 0; n.offB[l] = 0; }
    if (d.log_std_offset >= 0) {
        PPOAF_REQUIRE(d.log_std_offset == off, "%s: log_std expected at offset %ld, got %ld", what, off,
                      (long)d.log_std_offset);
        off += pad4(d.out_dim);
    }
    PPOAF_REQUIRE(off == d.size, "%s: bucket size %ld does not match the layer table (%ld)", what,
                  (long)d.size, off);
    return PPOAF_OK;
}

static int make_dev(const ppoaf_ppo_update_args_t* a, UpdateDev& u) {
    PPOAF_REQUIRE(a, "ppo_update: null args");
    int rc = fill_net(a->actor, u.net[0], "actor");
    if (rc) return rc;
    rc = fill_net(a->critic, u.net[1], "critic");
    if (rc) return rc;
    PPOAF_REQUIRE(a->actor.offset == 0 && a->critic.offset == a->actor.size &&
                      a->bucket_total == a->actor.size + a->critic.size,
                  "ppo_update: actor and critic must be adjacent in one bucket");
    PPOAF_REQUIRE(a->critic.out_dim == 1 && a->critic.log_std_offset < 0, "ppo_update: critic out_dim must be 1");
    PPOAF_REQUIRE(a->head_kind == PPOAF_HEAD_CATEGORICAL || a->head_kind == PPOAF_HEAD_GAUSSIAN,
                  "ppo_update: head_kind=%d", a->head_kind);
    PPOAF_REQUIRE((a->head_kind == PPOAF_HEAD_GAUSSIAN) == (a->actor.log_std_offset >= 0),
                  "ppo_update: log_std offset must be given exactly for the Gaussian head");
    PPOAF_REQUIRE(a->B >= 2 && a->batch_stride >= a->B, "ppo_update: B=%ld stride=%ld", (long)a->B,
                  (long)a->batch_stride);
    PPOAF_REQUIRE(a->params && a->grads && a->exp_avg && a->exp_avg_sq && a->slabs && a->step_counts &&
                      a->lr && a->norm_scratch && a->obs && a->critic_obs && a->raw_actions &&
                      a->advantages && a->old_log_probs && a->rewards_to_go && a->values && a->perm &&
                      a->cursor && a->vn_mean && a->vn_var && a->vn_count && a->loss_partials && a->totals,
                  "ppo_update: null pointer");
    PPOAF_REQUIRE(!a->normalize_values || (a->vn_records && a->n_ranks >= 1), "ppo_update: vn_records missing");
    PPOAF_REQUIRE(((uintptr_t)a->params & 15) == 0 && ((uintptr_t)a->grads & 15) == 0 &&
                      ((uintptr_t)a->slabs & 15) == 0 && ((uintptr_t)a->exp_avg & 15) == 0 &&
                      ((uintptr_t)a->exp_avg_sq & 15) == 0,
                  "ppo_update: buckets must be 16-byte aligned");
    u.params = a->params; u.grads = a->grads; u.exp_avg = a->exp_avg; u.exp_avg_sq = a->exp_avg_sq;
    u.slabs = a->slabs; u.bucket_total = a->bucket_total; u.step_counts = a->step_counts; u.lr = a->lr;
    u.norm_scratch = a->norm_scratch; u.beta1 = a->beta1; u.beta2 = a->beta2; u.adam_eps = a->adam_eps;
    u.grad_scale = a->grad_scale; u.max_norm = a->max_norm; u.head_kind = a->head_kind;
    u.obs = a->obs; u.critic_obs = a->critic_obs; u.raw_actions = a->raw_actions; u.adv = a->advantages;
    u.old_lp = a->old_log_probs; u.rtg = a->rewards_to_go; u.values = a->values; u.perm = a->perm;
    u.row_map = a->row_map; u.n_rows = a->n_rows; u.cursor = a->cursor; u.B = a->B;
    u.batch_stride = a->batch_stride; u.normalize_values = a->normalize_values; u.n_ranks = a->n_ranks;
    u.vn_mean = a->vn_mean; u.vn_var = a->vn_var; u.vn_count = a->vn_count; u.vn_records = a->vn_records;
    u.normalize_adv = a->normalize_adv; u.use_huber = a->use_huber; u.surr_clip = a->surr_clip;
    u.entropy_weight = a->entropy_weight; u.kl_loss_weight = a->kl_loss_weight;
    u.huber_delta = a->huber_delta; u.min_std = a->min_std; u.loss_partials = a->loss_partials;
    u.totals = a->totals;
    u.n_wg = (int)((a->B + kRows - 1) / kRows);
    return PPOAF_OK;
}

static size_t fwd_bwd_lds_bytes(const UpdateDev& u) {
    size_t worst = 0;
    for (int w = 0; w < 2; ++w) {
        const NetDev& n = u.net[w];
        const size_t HS = n.H + 4, INP = ((n.in_dim + 3) & ~3) + 4;
        size_t f = 80 + kRows * INP + (size_t)n.depth * kRows * HS + 2 * kRows * HS + 2 * kRows * kMaxOut;
        if (w == 0) f += (size_t)(u.B < kMaxAdvLds ? u.B : kMaxAdvLds);
        f = (f + 3) / 4 * 4;
        if (f * 4 > worst) worst = f * 4;
    }
    return worst;
}

}  // namespace ppoaf

using namespace ppoaf;

extern "C" int ppoaf_ppo_update_fwd_bwd(const ppoaf_ppo_update_args_t* args, ppoaf_stream_t stream) {
    UpdateDev u;
    int rc = make_dev(args, u);
    if (rc) return rc;
    const size_t lds = fwd_bwd_lds_bytes(u);
    PPOAF_REQUIRE(lds <= 160 * 1024, "ppo_update_fwd_bwd: needs %zu B of LDS (> 160 KiB)", lds);
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_update_fwd_bwd_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PPOAF_E_LAUNCH; }
        attr_set = true;
    }
    hipLaunchKernelGGL(ppo_update_fwd_bwd_kernel, dim3(u.n_wg, 2), dim3(kThreads), lds,
                       (hipStream_t)stream, u);
    return check_launch("ppo_update_fwd_bwd");
}

extern "C" int ppoaf_ppo_update_reduce(const ppoaf_ppo_update_args_t* args, int compute_norms,
                                       ppoaf_stream_t stream) {
    UpdateDev u;
    int rc = make_dev(args, u);
    if (rc) return rc;
    const long n4 = u.bucket_total >> 2;
    hipLaunchKernelGGL(ppo_update_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, u, compute_norms);
    return check_launch("ppo_update_reduce");
}

extern "C" int ppoaf_ppo_update_adam(const ppoaf_ppo_update_args_t* args, int compute_norms,
                                     ppoaf_stream_t stream) {
    UpdateDev u;
    int rc = make_dev(args, u);
    if (rc) return rc;
    const long n4 = u.bucket_total >> 2;
    const unsigned grid = (unsigned)((n4 + 255) / 256);
    if (compute_norms) {
        hipLaunchKernelGGL(ppo_update_sqnorm_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, u);
        rc = check_launch("ppo_update_adam/sqnorm");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(ppo_update_adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, u);
    return check_launch("ppo_update_adam");
}

extern "C" int ppoaf_minibatch_moments(const float* data, const int64_t* perm, const int32_t* row_map,
                                       int64_t n_perm, int64_t B, double* records,
                                       ppoaf_stream_t stream) {
    PPOAF_REQUIRE(data && perm && records, "minibatch_moments: null pointer");
    PPOAF_REQUIRE(n_perm >= 1 && B >= 1, "minibatch_moments: n_perm=%ld B=%ld", (long)n_perm, (long)B);
    const long nb = (n_perm + B - 1) / B;
    PPOAF_REQUIRE(nb <= 0x7fffffffL, "minibatch_moments: too many mini-batches");
    hipLaunchKernelGGL(minibatch_moments_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream,
                       data, perm, row_map, (long)n_perm, (long)B, records);
    return check_launch("minibatch_moments");
}
