// Shared by the fused PPO update kernels (ppo_update.hip: row-tiled three-launch chain and its single-XCD
// persistent form; ppo_update_ws.hip: weight-stationary persistent form): the device view of the C-ABI
// arguments and the distribution head + loss terms of one 16-row block.
#pragma once
#include "mlp_device.hpp"

namespace ppoaf {

// Per-mini-batch panels shared by the layered mode of the two-XCD persistent kernel and by the split-wgrad chain
// (fwd_bwd publishes them, the wgrad launch consumes them): workspace memory, 256-byte aligned pieces.
struct WsDev {
    float* hbuf[2];                       // [depth][Bp][H]   hidden activations of the mini-batch
    float* dbuf[2];                       // [depth][Bp][H]   dLoss / dz
    float* outpart[2];                    // [ceil(B/16)][seg] output-layer (+ log_std) gradient partials per row block
    float* xbuf[2];                       // [Bp][64]         the mini-batch's gathered input rows, zero padded (layer-0 wgrad)
    int W;                                // workers per network (persistent kernel)
    int xcc[2];                           // XCD of the actor / critic workers (persistent kernel)
    int Bp;                               // B rounded up to 64
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
    int64_t* cursor; long B, batch_stride, mb_offset, cursor_advance;
    int normalize_values, n_ranks;
    float* vn_mean; float* vn_var; double* vn_count; const double* vn_records;
    const double* adv_records;
    int normalize_adv, use_huber, pregathered;
    float surr_clip, entropy_weight, kl_loss_weight, huber_delta, min_std;
    float* loss_partials; double* totals;
    int n_wg;
    int confine;         // args->xcd_half: 0 every XCD; 1 / 2: fwd_bwd's workgroups on XCDs 0-3 / 4-7 only (actor on two of them, critic on two)
    int split;           // 1: split-wgrad chain -- fwd_bwd publishes activation / dz panels (sp) instead of weight-gradient slabs
    WsDev sp;
};

// offset of the output layer's segment (W_out, b_out, log_std) inside a network's bucket, and its length
inline long ws_seg_off(const NetDev& n) {
    const long szW0 = ((long)n.H * n.in_dim + 3) & ~3L;
    return n.depth == 0 ? 0 : szW0 + n.H + (long)(n.depth - 1) * ((long)n.H * n.H + n.H);
}
inline long ws_seg_len(const NetDev& n) { return n.size - ws_seg_off(n); }

// split-wgrad launch: one 4-wave workgroup per 16 x 32 piece of every layer's weight gradient (two MFMA tiles that share
// the dz operand and read whole 128-byte lines of the input panel; layer 0: ceil(in_dim / 32) pieces per 16 output rows)
// + one per network for the output layer's segment.  Jobs are dealt to workgroups in XCD-sized runs (workgroup b runs on
// XCD b % 8 and takes job (b % 8) * per_xcd + b / 8), so that a layer's panels are fetched by one or two XCDs instead of
// all eight; the bookkeeping workgroup comes last.
inline int split_wgrad_jobs(const NetDev& n) {
    const int t = n.H / 16, t2 = (t + 1) / 2, p0 = ((n.in_dim + 15) / 16 + 1) / 2;
    return (n.depth - 1) * t * t2 + t * p0 + 1;
}
inline int split_wgrad_per_xcd(const UpdateDev& u) { return (split_wgrad_jobs(u.net[0]) + split_wgrad_jobs(u.net[1]) + 7) / 8; }
inline int split_wgrad_blocks(const UpdateDev& u) { return 8 * split_wgrad_per_xcd(u); }      // some run no job (partials 0)

// workspace layout: per network hbuf, dbuf ([depth][Bp][H] each), outpart ([ceil(B/16)][seg]) and xbuf ([Bp][64])
inline size_t ws_layout(const UpdateDev& u, WsDev* ws, char* base) {
    const long Bp = (u.B + 63) & ~63L;
    size_t off = 0;
    auto take = [&](size_t floats) { const size_t o = off; off += (floats * 4 + 255) & ~(size_t)255; return o; };
    for (int w = 0; w < 2; ++w) {
        const NetDev& n = u.net[w];
        const size_t plane = (size_t)n.depth * Bp * n.H;
        const size_t oh = take(plane), od = take(plane), oo = take((size_t)((u.B + 15) / 16) * ws_seg_len(n));
        const size_t ox = take((size_t)Bp * 64);
        if (ws) {
            ws->hbuf[w] = reinterpret_cast<float*>(base + oh);
            ws->dbuf[w] = reinterpret_cast<float*>(base + od);
            ws->outpart[w] = reinterpret_cast<float*>(base + oo);
            ws->xbuf[w] = reinterpret_cast<float*>(base + ox);
        }
    }
    if (ws) ws->Bp = (int)Bp;
    return off;
}

// ---- row tiles of 256-wide networks on workgroup pairs (args->row_pairs; ppo_update_rowpair.hpp): the record region
constexpr int kPairHeaderBytes = 256;                 // word 0: error (a wait ran out of time)
constexpr int kPairRecBytes = 16 * 1024;              // a half panel (16 rows x 128 floats) as 1024 records of 16 bytes
constexpr int kPairMaxTiles = 32;                     // B <= 512 (split-wgrad chain)
constexpr long long kPairWaitTicks = 200000000LL;     // 2 s of wall_clock64 (100 MHz)

struct PairDev {
    unsigned char* base;                              // header, then per network [phase][tile][half] record blocks
    long net_off[2];                                  // byte offset of a network's blocks; 0: that network does not run in pairs
};

inline bool pair_eligible(const NetDev& n) { return n.H == 256 && n.depth >= 2 && n.depth <= 4; }
inline int pair_phases(const NetDev& n) { return 2 * n.depth - 3; }
// the region's place does not move with the mini-batch size (a tail mini-batch's panels are laid out differently, and
// nothing but records may ever be stored where records are polled): behind the panels of the FULL batch size
inline size_t pair_region_offset(const UpdateDev& u) {
    UpdateDev f = u;
    if (u.batch_stride > f.B) f.B = u.batch_stride;
    return ws_layout(f, nullptr, nullptr);
}
inline size_t pair_region_layout(const UpdateDev& u, PairDev* p, char* region) {
    size_t off = kPairHeaderBytes;
    for (int w = 0; w < 2; ++w) {
        const bool on = pair_eligible(u.net[w]);
        if (p) p->net_off[w] = on ? (long)off : 0;
        if (on) off += (size_t)pair_phases(u.net[w]) * kPairMaxTiles * 2 * kPairRecBytes;
    }
    if (p) p->base = reinterpret_cast<unsigned char*>(region);
    return off;
}

constexpr int kWgradThreads = 256;

// per-mini-batch bookkeeping of the split-wgrad chain, by ONE wave (threads 0..63 of a workgroup): loss partials -> totals
__device__ __forceinline__ void ppo_update_bookkeeping_totals(const UpdateDev& u) {
    const int lane = threadIdx.x;
    float p0 = 0.f, p2 = 0.f, p3 = 0.f, p4 = 0.f, p7 = 0.f;
    for (int g = lane; g < u.n_wg; g += 64) {
        const float* a = u.loss_partials + (long)g * 8;
        const float* cc = u.loss_partials + ((long)u.n_wg + g) * 8;
        p0 += a[0]; p3 += a[3]; p4 += a[4]; p7 += a[7]; p2 += cc[2];
    }
    p0 = wave_sum(p0); p2 = wave_sum(p2); p3 = wave_sum(p3); p4 = wave_sum(p4); p7 = wave_sum(p7);
    if (lane == 0) {
        const float n = (float)u.B;
        const float surr = p0 / n, ent = p3 / n, kl = p4 / n, crit = p2 / n;
        float total = surr;
        if (u.entropy_weight != 0.0f) total -= u.entropy_weight * ent;
        if (u.kl_loss_weight > 0.0f) total += u.kl_loss_weight * kl;
        u.totals[0] += (double)surr; u.totals[1] += (double)total; u.totals[2] += (double)crit;
        u.totals[3] += (double)ent; u.totals[4] += (double)kl;
        u.totals[5] += (double)u.loss_partials[5]; u.totals[6] += (double)u.loss_partials[6];
        u.totals[7] += p7 > 0.f ? 1.0 : 0.0;
        u.totals[8] += 1.0;
    }
}
// ... and the step counters + Adam bias corrections of the step being taken
__device__ __forceinline__ void ppo_update_bookkeeping_steps(const UpdateDev& u) {
    const int lane = threadIdx.x;
    if (lane < 2) {                                       // one lane per network: step counter + bias corrections
        const int w = lane;
        const int64_t t = u.step_counts[w] + 1;
        u.step_counts[w] = t;
        u.norm_scratch[2 + 2 * w] = 1.0 - pow((double)u.beta1, (double)t);
        u.norm_scratch[3 + 2 * w] = sqrt(1.0 - pow((double)u.beta2, (double)t));
    }
}

__device__ __forceinline__ void ppo_update_bookkeeping_split(const UpdateDev& u) {
    if (threadIdx.x >= 64) return;
    ppo_update_bookkeeping_totals(u);
    ppo_update_bookkeeping_steps(u);
}

// host: validate ppoaf_ppo_update_args_t and fill the device view (ppo_update.hip)
int make_update_dev(const ppoaf_ppo_update_args_t* a, UpdateDev& u);

__device__ __forceinline__ unsigned hw_xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xfu;
}

// K6 + K3 for the 16 rows of block g of network `which` (0 actor, 1 critic), run by ONE wave (lanes 0..15 hold a
// row each; the categorical head uses 4 lanes per row): distribution head on the output-layer values sOut[16][16],
// PPO / value loss terms (ppo.py:2325-2438), d loss / d out -> sDOut[16][16] (and, Gaussian head, per-row
// d / d log_std parked in sOut[.][8..]), the block's loss partials -> u.loss_partials, critic values written back.
//   sRow16[16] dataset row of each block row (-1: dead), sRowF[3][16] adv / old log-prob / rewards-to-go,
//   sMisc[4] adv mean / std, value-normaliser mean / var, sActF[16][8] raw actions, log_std_p the actor's log_std.
template <bool NT, typename U>
__device__ __forceinline__ void ppo_head_loss(const U& u, const int which, const int g, const int out_dim,
                                              const float* __restrict__ log_std_p, const int* sRow,
                                              const float* sRowF, const float* sMisc, float* sActF, float* sOut,
                                              float* sDOut, const int lane, const long B) {
    const int s = lane;                       // lanes 0..15 hold one row each
    const int row = s < kRows ? sRow[s] : -1;
    const bool live = row >= 0;
    const float inv_B = 1.0f / (float)B;
    float part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (which == 0) {
        float logp = 0.f, ent = 0.f;
        float av = 0.f, lpo = 0.f;
        if (live) {
            av = sRowF[s]; lpo = sRowF[16 + s];
            if (u.normalize_adv) av = (av - sMisc[0]) / (sMisc[1] + 1e-8f);
        }
        if (u.head_kind == PPOAF_HEAD_CATEGORICAL) {
            // lane-parallel: 4 lanes per row, lane (s4, q) owns classes q and q + 4; the class
            // reductions are two xor-shuffles inside the 4-lane group, so the transcendental chain
            // is 2 values long instead of 8.  Row results are handed to lane s4 at the end.
            const int s4 = lane >> 2, q = lane & 3;
            const bool live4 = sRow[s4] >= 0;
            float av4 = 0.f, lpo4 = 0.f;
            if (live4) {
                av4 = sRowF[s4]; lpo4 = sRowF[16 + s4];
                if (u.normalize_adv) av4 = (av4 - sMisc[0]) / (sMisc[1] + 1e-8f);
            }
            auto rsum = [](float v) { return group4_sum(v); };       // xor 1, xor 2 as DPP quad permutes (mlp_device.hpp)
            auto rmax = [](float v) { return group4_max(v); };
            const int k0 = q, k1 = q + 4;
            const bool v0 = k0 < out_dim, v1 = k1 < out_dim;
            const float z0 = v0 ? sOut[s4 * kMaxOut + k0] : -INFINITY, z1 = v1 ? sOut[s4 * kMaxOut + k1] : -INFINITY;
            const float m = rmax(fmaxf(z0, z1));
            float p0 = v0 ? expf(z0 - m) : 0.f, p1 = v1 ? expf(z1 - m) : 0.f;
            const float inv = 1.0f / rsum(p0 + p1);
            p0 *= inv; p1 *= inv;
            const float s2 = rsum(p0 + p1);
            int a = reinterpret_cast<const int*>(sActF)[s4 * 8];
            a = a < 0 ? 0 : (a >= out_dim ? out_dim - 1 : a);
            const float n0 = p0 / s2, n1 = p1 / s2;                          // Categorical's renormalisation
            const float l0 = v0 ? logf(clamp_prob_u(n0)) : 0.f, l1 = v1 ? logf(clamp_prob_u(n1)) : 0.f;
            const float ent4 = -rsum((v0 ? n0 * l0 : 0.f) + (v1 ? n1 * l1 : 0.f));
            const float logp4 = rsum((k0 == a ? l0 : 0.f) + (k1 == a ? l1 : 0.f));
            const float ratio = expf(logp4 - lpo4);
            const float bad4 = (isnan(ratio) || isinf(ratio)) ? 1.f : 0.f;
            const float lo = 1.0f - u.surr_clip, hi = 1.0f + u.surr_clip;
            const float surr1 = ratio * av4, surr2 = fminf(fmaxf(ratio, lo), hi) * av4;
            float glp;
            if (surr1 <= surr2) glp = -av4 * ratio;
            else glp = (ratio >= lo && ratio <= hi) ? -av4 * ratio : 0.f;
            glp *= inv_B;
            const float gH = (u.entropy_weight != 0.0f) ? -u.entropy_weight * inv_B : 0.f;
            // chain: z -softmax-> p -(/sum)-> n -clamp,log-> l
            auto gk_of = [&](bool valid, int k, float nk, float lg) {
                if (!valid) return 0.f;
                const float ck = clamp_prob_u(nk);
                const float in_range = (nk >= FLT_EPSILON && nk <= 1.0f - FLT_EPSILON) ? 1.f : 0.f;
                float gk = gH * (-lg - nk * in_range / ck);
                if (k == a) gk += glp * in_range / ck;
                return gk;
            };
            float g0 = gk_of(v0, k0, n0, l0), g1 = gk_of(v1, k1, n1, l1);
            const float dot = rsum(g0 * n0 + g1 * n1);
            g0 = (g0 - dot) / s2; g1 = (g1 - dot) / s2;
            const float dot2 = rsum(g0 * p0 + g1 * p1);
            if (live4) {
                if (k0 < 8) sDOut[s4 * kMaxOut + k0] = p0 * (g0 - dot2);
                sDOut[s4 * kMaxOut + k1] = p1 * (g1 - dot2);
            } else {
                sDOut[s4 * kMaxOut + k0] = 0.f; sDOut[s4 * kMaxOut + k1] = 0.f;
                if (q == 0) for (int k2 = 8; k2 < kMaxOut; ++k2) sDOut[s4 * kMaxOut + k2] = 0.f;
            }
            // row results -> lane s (= row s) for the partial sums below
            const int src = (lane & 15) * 4;
            const float r_surr = __shfl(-fminf(surr1, surr2), src, 64), r_ent = __shfl(ent4, src, 64);
            const float r_kl = __shfl(lpo4 - logp4, src, 64), r_bad = __shfl(bad4, src, 64);
            if (live) { part[0] = r_surr; part[3] = r_ent; part[4] = r_kl; part[7] = r_bad; }
        } else if (live) {
            // tanh-Gaussian (distributions.py:441-694)
            const float* log_std = log_std_p;
            const float* x = sActF + s * 8;
            float lp = 0.f, slog = 0.f;
            ent = 0.f;
            for (int d = 0; d < out_dim; ++d) {
                const float sd = fmaxf(softplus_u(ld1<NT>(log_std + d)), u.min_std);
                const float mu = sOut[s * kMaxOut + d];
                const float zz = x[d] - mu;
                const float l0 = -logf(sd) - 0.91893853320467274178f;
                float l = -(zz * zz) / (2.0f * sd * sd) + l0;
                l = fminf(fmaxf(l, -100.f), 100.f);
                lp += l;
                const float th = tanhf(x[d]);
                slog += logf(fmaxf(1.0f - th * th, 1e-6f));
                // entropy := -log_prob of the distribution's MEAN (ppo_policy.py:950, distributions.py:672-694)
                const float thm = tanhf(mu);
                ent += logf(fmaxf(1.0f - thm * thm, 1e-6f)) - fminf(fmaxf(l0, -100.f), 100.f);
            }
            logp = lp - slog;
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
            for (int d = 0; d < out_dim; ++d) {
                const float ls = ld1<NT>(log_std + d);
                const float sp = softplus_u(ls), sd = fmaxf(sp, u.min_std);
                const float mu = sOut[s * kMaxOut + d];
                const float zz = x[d] - mu;
                const float l0 = -logf(sd) - 0.91893853320467274178f;
                const float l = -(zz * zz) / (2.0f * sd * sd) + l0;
                const float pass = (l >= -100.f && l <= 100.f) ? 1.f : 0.f;
                const float pass0 = (l0 >= -100.f && l0 <= 100.f) ? 1.f : 0.f;
                const float thm = tanhf(mu);
                const float pass_t = (1.0f - thm * thm >= 1e-6f) ? 1.f : 0.f;
                sDOut[s * kMaxOut + d] = glp * pass * zz / (sd * sd) - gH * pass_t * 2.0f * thm;
                if (d == 0) for (int k2 = out_dim; k2 < 8; ++k2) sDOut[s * kMaxOut + k2] = 0.f;
                const float dmax = sp > u.min_std ? 1.f : (sp == u.min_std ? 0.5f : 0.f);
                const float dsp = ls > 20.f ? 1.f : 1.0f / (1.0f + expf(-ls));
                // per-row d logp / d log_std, parked in sOut's upper half for the reduction below
                sOut[s * kMaxOut + 8 + d] = (glp * pass * (zz * zz / (sd * sd * sd) - 1.0f / sd) + gH * pass0 / sd) * dmax * dsp;
            }
        } else if (s < kRows) {
            for (int k = 0; k < kMaxOut; ++k) sDOut[s * kMaxOut + k] = 0.f;
            if (u.head_kind == PPOAF_HEAD_GAUSSIAN)
                for (int d = 0; d < 8; ++d) sOut[s * kMaxOut + 8 + d] = 0.f;
        }
    } else {
        if (live) {
            const float v = sOut[s * kMaxOut];
            float rt = sRowF[32 + s];
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
            for (int k2 = 1; k2 < 8; ++k2) sDOut[s * kMaxOut + k2] = 0.f;
            u.values[row] = v;                                   // ppo.py:2340
        } else if (s < kRows) {
            for (int k2 = 0; k2 < 8; ++k2) sDOut[s * kMaxOut + k2] = 0.f;
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

}  // namespace ppoaf
