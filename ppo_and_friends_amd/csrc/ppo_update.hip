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
#include "ppo_update_rowpair.hpp"
#include <hip/hip_ext.h>
#include "peer_exchange_device.hpp"
#include <cstdlib>

namespace ppoaf {


// 1-D grid of 2 * n_wg workgroups.  Every workgroup of a network streams that network's whole
// weight set, so workgroups of one network are placed on the same XCDs (blocks b and b + 8 share
// an XCD under the observed round-robin dispatch): XCDs 0-3 take the actor, 4-7 the critic, and
// each weight line is then fetched into an XCD's L2 once for its 4 consumers instead of once per
// pair.  Placement only changes speed; nothing depends on it.
template <int HTA, int HTC, bool SPLIT>
__global__ __launch_bounds__(kThreadsU) void ppo_update_fwd_bwd_kernel(UpdateDev u) {
    const int b = blockIdx.x;
    int which = (b >> 2) & 1;                              // b % 8 in {0..3} -> actor, {4..7} -> critic
    int g = ((b >> 3) << 2) | (b & 3);
    if (u.confine) {                                // one half of the XCDs left to another chain (args->xcd_half)
        const int x = b & 7;
        if ((x >> 2) != u.confine - 1) return;
        which = (x & 3) >> 1;                              // the half's first two XCDs: actor, the other two: critic
        g = ((b >> 3) << 1) | (x & 1);
    }
    if (g >= u.n_wg) return;                               // uniform per workgroup, before any barrier
    if (which == 0) ppo_update_fwd_bwd_body<HTA, SPLIT>(u, 0, g);
    else ppo_update_fwd_bwd_body<HTC, SPLIT>(u, 1, g);
}

// args->row_pairs (split-wgrad chain): a 256-wide network's row tiles on pairs of workgroups (ppo_update_rowpair.hpp); the
// other network's tiles as above.  Workgroup b runs on XCD b % 8; slot j = b / 8 of an XCD holds tile pair member j & 1.
template <int HTA, int HTC>
__global__ __launch_bounds__(kThreadsU) void ppo_update_fwd_bwd_pair_kernel(UpdateDev u, PairDev pd) {
    const int b = blockIdx.x, x = b & 7, j = b >> 3;
    int which, g, hf = 0;
    if (u.confine) {
        if ((x >> 2) != u.confine - 1) return;
        which = (x & 3) >> 1;
        if (which == 0 ? HTA == 16 : HTC == 16) { hf = j & 1; g = ((j >> 1) << 1) | (x & 1); }
        else g = (j << 1) | (x & 1);
    } else {
        which = x >> 2;
        if (which == 0 ? HTA == 16 : HTC == 16) { hf = j & 1; g = ((j >> 1) << 2) | (x & 3); }
        else g = (j << 2) | (x & 3);
    }
    if (g >= u.n_wg) return;                               // uniform per workgroup, before any barrier
    if (which == 0) {
        if constexpr (HTA == 16) ppo_update_fwd_bwd_pair_body<16>(u, 0, g, hf, pd);
        else ppo_update_fwd_bwd_body<HTA, true>(u, 0, g);
    } else {
        if constexpr (HTC == 16) ppo_update_fwd_bwd_pair_body<16>(u, 1, g, hf, pd);
        else ppo_update_fwd_bwd_body<HTC, true>(u, 1, g);
    }
}

// slabs -> gradient bucket in a fixed order.  The slabs were just written by other CUs, so every
// read is a cold miss: each thread owns one float4 column and puts the loads of 8 slabs in flight
// at a time, then adds them in slab order (bitwise reproducible, no LDS round trip, no barrier
// before the result is stored).  Block 0 also folds the loss partials into the totals, advances the
// Adam step counters and publishes the bias corrections (computed once, in double, instead of per
// thread in the Adam kernel).
constexpr int kRedThreads = 256;

__device__ __forceinline__ void ppo_update_bookkeeping(const UpdateDev& u);

__global__ __launch_bounds__(kRedThreads) void ppo_update_reduce_kernel(UpdateDev u, int compute_norms) {
    __shared__ double red[17];
    if (blockIdx.x == gridDim.x - 1) { ppo_update_bookkeeping(u); return; }      // uniform per workgroup
    const long n4 = u.bucket_total >> 2;
    const long idx = (long)blockIdx.x * kRedThreads + threadIdx.x;
    const float4* sl = reinterpret_cast<const float4*>(u.slabs);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < n4) {
        for (int g0 = 0; g0 < u.n_wg; g0 += 8) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                v[k] = (g0 + k < u.n_wg) ? sl[(long)(g0 + k) * n4 + idx] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
        }
    }
    double q0 = 0.0, q1 = 0.0;
    if (idx < n4) {
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
        // per-workgroup partials (no atomics): the Adam launch adds them in a fixed association
        if (threadIdx.x == 0) { u.norm_scratch[6 + 2 * blockIdx.x] = q0; u.norm_scratch[7 + 2 * blockIdx.x] = q1; }
    }
}

// N > 1 with the K17 peer exchange: the slab reduce and the exchange in ONE launch.  The workgroup's float4
// column sums go straight from registers into this rank's exchange slot (no copy through the gradient
// bucket), the ranks' sums are added in rank order, and the bucket receives the cross-rank gradient together
// with both clip norms (fixed-order, identical on every rank) -- the Adam launch follows without a norm pass.
// Grid as the plain reduce: one column per thread, all workgroups resident (<= kXchgMaxGrid), bookkeeping last.
__global__ __launch_bounds__(kRedThreads) void ppo_update_reduce_exchange_kernel(UpdateDev u, XchgDev x,
                                                                                long long wait_ticks) {
    __shared__ double red[17];
    if (blockIdx.x == gridDim.x - 1) { ppo_update_bookkeeping(u); return; }      // uniform per workgroup
    const long n4 = u.bucket_total >> 2;
    const long idx = (long)blockIdx.x * kRedThreads + threadIdx.x;
    const float4* sl = reinterpret_cast<const float4*>(u.slabs);
    const long long seq = xchg_sequence(x, blockIdx.x);
    const long slot = (long)(seq & 1) * x.n4;
    float4 own = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < n4) {
        for (int g0 = 0; g0 < u.n_wg; g0 += 8) {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                v[k] = (g0 + k < u.n_wg) ? sl[(long)(g0 + k) * n4 + idx] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 8; ++k) { own.x += v[k].x; own.y += v[k].y; own.z += v[k].z; own.w += v[k].w; }
        }
        x.my_slots[slot + idx] = own;
    }
    xchg_publish(x, seq, blockIdx.x);
    xchg_wait(x, seq, blockIdx.x, wait_ticks);
    double q0 = 0.0, q1 = 0.0;
    if (idx < n4) {
        const float4 acc = xchg_sum(x, slot, idx, own);
        reinterpret_cast<float4*>(u.grads)[idx] = acc;
        const double q = xchg_sq(acc, u.grad_scale);
        if (idx * 4 < u.net[0].size) q0 = q; else q1 = q;
    }
    // partials only: the Adam launch that follows adds them in workgroup order itself (ppo_update_adam_kernel)
    q0 = block_sum(q0, red);
    q1 = block_sum(q1, red);
    if (threadIdx.x == 0) { x.norm_partials[2 * blockIdx.x] = q0; x.norm_partials[2 * blockIdx.x + 1] = q1; }
    xchg_advance(x, seq, blockIdx.x);
}

// The per-mini-batch bookkeeping runs in its own (last) workgroup so that it overlaps the slab
// reads instead of extending workgroup 0: loss partials summed by one wave (lane = workgroup of the
// fwd_bwd kernel), totals, Adam step counters and bias corrections.
__device__ __forceinline__ void ppo_update_bookkeeping(const UpdateDev& u) {
    if (threadIdx.x >= 64) return;
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
    if (lane < 2) {                                       // one lane per network: step counter + bias corrections
        const int w = lane;
        const int64_t t = u.step_counts[w] + 1;
        u.step_counts[w] = t;
        u.norm_scratch[2 + 2 * w] = 1.0 - pow((double)u.beta1, (double)t);       // bias correction 1
        u.norm_scratch[3 + 2 * w] = sqrt(1.0 - pow((double)u.beta2, (double)t)); // sqrt(bias correction 2)
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
    if (threadIdx.x == 0) { u.norm_scratch[6 + 2 * blockIdx.x] = q0; u.norm_scratch[7 + 2 * blockIdx.x] = q1; }
}

// norm_partials != nullptr: the squared norms arrive as per-workgroup partials of the fused reduce + exchange
// launch (n_norm_groups pairs); every wave adds them in workgroup order -- identical on all ranks.
__global__ __launch_bounds__(256) void ppo_update_adam_kernel(UpdateDev u, const double* norm_partials,
                                                              unsigned n_norm_groups) {
    const long n4 = u.bucket_total >> 2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = idx < n4;
    // the column's loads go out first; the norm partials (another cold round trip, then an ordered sum) are
    // fetched while they are in flight
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f), g = p, m = p, v = p;
    if (live) {
        p = reinterpret_cast<float4*>(const_cast<float*>(u.params))[idx];
        g = reinterpret_cast<const float4*>(u.grads)[idx];
        m = reinterpret_cast<float4*>(u.exp_avg)[idx];
        v = reinterpret_cast<float4*>(u.exp_avg_sq)[idx];
    }
    double sq0 = 0.0, sq1 = 0.0;
    if (norm_partials) xchg_ordered_norms(norm_partials, n_norm_groups, sq0, sq1);   // uniform branch, whole waves
    if (live) {
        const int which = (idx * 4 < u.net[0].size) ? 0 : 1;
        const float total_norm = (float)sqrt(norm_partials ? (which ? sq1 : sq0) : u.norm_scratch[which]);
        float coef = 1.0f;
        if (u.max_norm > 0.f) coef = fminf(u.max_norm / (total_norm + 1e-6f), 1.0f);
        const float gs = u.grad_scale * coef;
        const float lr = u.lr[0];
        const float step_size = (float)((double)lr / u.norm_scratch[2 + 2 * which]);
        const float bc2_sqrt = (float)u.norm_scratch[3 + 2 * which];
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
    if (blockIdx.x == 0 && threadIdx.x == 0 && u.cursor_advance) u.cursor[0] += u.cursor_advance;
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

int fill_net(const ppoaf_mlp_desc_t& d, NetDev& n, const char* what) {
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

int make_update_dev(const ppoaf_ppo_update_args_t* a, UpdateDev& u) {
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
    PPOAF_REQUIRE(!a->normalize_adv || a->adv_records, "ppo_update: adv_records missing");
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
    u.batch_stride = a->batch_stride; u.mb_offset = a->mb_offset; u.cursor_advance = a->cursor_advance;
    u.normalize_values = a->normalize_values; u.n_ranks = a->n_ranks;
    u.vn_mean = a->vn_mean; u.vn_var = a->vn_var; u.vn_count = a->vn_count; u.vn_records = a->vn_records;
    u.adv_records = a->adv_records;
    u.normalize_adv = a->normalize_adv; u.use_huber = a->use_huber; u.surr_clip = a->surr_clip;
    u.entropy_weight = a->entropy_weight; u.kl_loss_weight = a->kl_loss_weight;
    u.huber_delta = a->huber_delta; u.min_std = a->min_std; u.pregathered = a->inputs_in_batch_order != 0; u.loss_partials = a->loss_partials;
    u.totals = a->totals;
    u.n_wg = (int)((a->B + kRows - 1) / kRows);
    // split-wgrad chain: the caller's workspace holds this mini-batch's activation / dz panels
    u.split = 0;
    PPOAF_REQUIRE(a->xcd_half >= 0 && a->xcd_half <= 2, "ppo_update: xcd_half=%d (0, 1 or 2)", a->xcd_half);
    u.confine = a->xcd_half;
    u.sp = WsDev();
    if (a->split_workspace) {
        PPOAF_REQUIRE((((uintptr_t)a->split_workspace) & 255) == 0, "ppo_update: split_workspace must be 256-byte aligned");
        PPOAF_REQUIRE(u.net[0].in_dim <= 64 && u.net[1].in_dim <= 64 && u.B <= 512,
                      "ppo_update: the split-wgrad chain covers in_dim <= 64 and B <= 512 (got %d / %d, %ld)", u.net[0].in_dim,
                      u.net[1].in_dim, u.B);
        size_t need = ws_layout(u, &u.sp, reinterpret_cast<char*>(a->split_workspace));
        PPOAF_REQUIRE(a->row_pairs == 0 || a->row_pairs == 1, "ppo_update: row_pairs=%d (0 or 1)", a->row_pairs);
        if (a->row_pairs) need = pair_region_offset(u) + pair_region_layout(u, nullptr, nullptr);      // the record region comes last
        PPOAF_REQUIRE((size_t)a->split_workspace_bytes >= need, "ppo_update: split_workspace of %ld B, %zu needed",
                      (long)a->split_workspace_bytes, need);
        u.split = 1;
    }
    return PPOAF_OK;
}

static size_t fwd_bwd_lds_bytes(const UpdateDev& u) {
    const size_t a = rowtile_lds_floats(u.net[0]), c = rowtile_lds_floats(u.net[1]);
    return ((a > c ? a : c) * 4 + 15) / 16 * 16 + kRowtileLineFloats * 4;
}

template <int HTA, int HTC, bool SPLIT>
static int launch_fwd_bwd_as(const UpdateDev& u, size_t lds, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_update_fwd_bwd_kernel<HTA, HTC, SPLIT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PPOAF_E_LAUNCH; }
        attr_set = true;
    }
    const unsigned grid = u.confine ? 8u * (unsigned)((u.n_wg + 1) / 2) : 8u * (unsigned)((u.n_wg + 3) / 4);     // groups of 4 actor + 4 critic blocks
    if (e0 || e1)        // the kernel's own begin / end stamped into the events (bench.py: roofline_update)
        hipExtLaunchKernelGGL((ppo_update_fwd_bwd_kernel<HTA, HTC, SPLIT>), dim3(grid), dim3(kThreadsU), lds, s, e0, e1, 0, u);
    else
        hipLaunchKernelGGL((ppo_update_fwd_bwd_kernel<HTA, HTC, SPLIT>), dim3(grid), dim3(kThreadsU), lds, s, u);
    return check_launch("ppo_update_fwd_bwd");
}

// split-wgrad chain with args->row_pairs: the 256-wide networks' tiles on workgroup pairs
template <int HTA, int HTC>
static int launch_fwd_bwd_pairs(const UpdateDev& u, const PairDev& pd, size_t lds, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ppo_update_fwd_bwd_pair_kernel<HTA, HTC>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return PPOAF_E_LAUNCH; }
        attr_set = true;
    }
    // slots per XCD: two per tile of a paired network (both networks share the grid: the wider need decides)
    const unsigned per_slot = u.confine ? 2u : 4u;
    const unsigned grid = 8u * 2u * (unsigned)((u.n_wg + per_slot - 1) / per_slot);
    if (e0 || e1)
        hipExtLaunchKernelGGL((ppo_update_fwd_bwd_pair_kernel<HTA, HTC>), dim3(grid), dim3(kThreadsU), lds, s, e0, e1, 0, u, pd);
    else
        hipLaunchKernelGGL((ppo_update_fwd_bwd_pair_kernel<HTA, HTC>), dim3(grid), dim3(kThreadsU), lds, s, u, pd);
    return check_launch("ppo_update_fwd_bwd(row pairs)");
}

template <int HTA, int HTC>
static int launch_fwd_bwd(const UpdateDev& u, size_t lds, hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
    return u.split ? launch_fwd_bwd_as<HTA, HTC, true>(u, lds, s, e0, e1) : launch_fwd_bwd_as<HTA, HTC, false>(u, lds, s, e0, e1);
}

}  // namespace ppoaf

using namespace ppoaf;

extern "C" int ppoaf_ppo_update_fwd_bwd(const ppoaf_ppo_update_args_t* args, ppoaf_stream_t stream) {
    return ppoaf_ppo_update_fwd_bwd_timed(args, nullptr, nullptr, stream);
}

extern "C" int ppoaf_ppo_update_fwd_bwd_timed(const ppoaf_ppo_update_args_t* args, void* start_event, void* stop_event,
                                              ppoaf_stream_t stream) {
    hipEvent_t e0 = (hipEvent_t)start_event, e1 = (hipEvent_t)stop_event;
    UpdateDev u;
    int rc = make_update_dev(args, u);
    if (rc) return rc;
    const size_t lds = fwd_bwd_lds_bytes(u);
    PPOAF_REQUIRE(lds <= 160 * 1024, "ppo_update_fwd_bwd: needs %zu B of LDS (> 160 KiB)", lds);
    hipStream_t s = (hipStream_t)stream;
    // instantiated (actor width, critic width) pairs; the host falls back to the torch path otherwise
    const int ha = u.net[0].H, hc = u.net[1].H;
    if (u.split && args->row_pairs && (pair_eligible(u.net[0]) || pair_eligible(u.net[1]))) {
        PairDev pd;
        const size_t lds_pairs = lds;
        pair_region_layout(u, &pd, reinterpret_cast<char*>(args->split_workspace) + pair_region_offset(u));
        if (ha == 128 && hc == 256 && pd.net_off[1]) return launch_fwd_bwd_pairs<8, 16>(u, pd, lds_pairs, s, e0, e1);
        if (ha == 256 && hc == 256 && pd.net_off[0] && pd.net_off[1]) return launch_fwd_bwd_pairs<16, 16>(u, pd, lds_pairs, s, e0, e1);
        // other shapes (a 256-wide network of depth 1 or > 4, a 256-wide actor beside a narrower critic): one workgroup per tile
    }
    if (ha == 32 && hc == 32) return launch_fwd_bwd<2, 2>(u, lds, s, e0, e1);
    if (ha == 64 && hc == 64) return launch_fwd_bwd<4, 4>(u, lds, s, e0, e1);
    if (ha == 128 && hc == 128) return launch_fwd_bwd<8, 8>(u, lds, s, e0, e1);
    if (ha == 256 && hc == 256) return launch_fwd_bwd<16, 16>(u, lds, s, e0, e1);
    if (ha == 128 && hc == 256) return launch_fwd_bwd<8, 16>(u, lds, s, e0, e1);
    if (ha == 64 && hc == 128) return launch_fwd_bwd<4, 8>(u, lds, s, e0, e1);
    set_error("ppo_update_fwd_bwd: hidden widths (actor %d, critic %d) not instantiated", ha, hc);
    return PPOAF_E_INVALID;
}

extern "C" int ppoaf_ppo_update_reduce(const ppoaf_ppo_update_args_t* args, int compute_norms,
                                       ppoaf_stream_t stream) {
    UpdateDev u;
    int rc = make_update_dev(args, u);
    if (rc) return rc;
    const long n4 = u.bucket_total >> 2;
    hipLaunchKernelGGL(ppo_update_reduce_kernel, dim3((unsigned)((n4 + kRedThreads - 1) / kRedThreads) + 1u),
                       dim3(kRedThreads), 0, (hipStream_t)stream, u, compute_norms);
    return check_launch("ppo_update_reduce");
}

extern "C" int ppoaf_ppo_update_reduce_exchange(const ppoaf_ppo_update_args_t* args, ppoaf_peer_exchange_t* xchg,
                                                double wait_seconds, ppoaf_stream_t stream) {
    UpdateDev u;
    int rc = make_update_dev(args, u);
    if (rc) return rc;
    PPOAF_REQUIRE(xchg && xchg->connected, "ppo_update_reduce_exchange: exchange missing or not connected");
    const long n4 = u.bucket_total >> 2;
    PPOAF_REQUIRE(xchg->dev.n4 == n4, "ppo_update_reduce_exchange: exchange made for %ld float4, bucket has %ld",
                  xchg->dev.n4, n4);
    PPOAF_REQUIRE(xchg->dev.n_ranks == u.n_ranks || !u.normalize_values, "ppo_update_reduce_exchange: %d ranks in the exchange, %d in args",
                  xchg->dev.n_ranks, u.n_ranks);
    const long groups = (n4 + kRedThreads - 1) / kRedThreads;
    PPOAF_REQUIRE(groups <= kXchgMaxGrid, "ppo_update_reduce_exchange: bucket of %ld floats needs %ld workgroups (at most %d: "
                  "use ppoaf_ppo_update_reduce + ppoaf_peer_exchange_allreduce)", (long)u.bucket_total, groups, kXchgMaxGrid);
    PPOAF_REQUIRE(wait_seconds > 0.0 && wait_seconds <= 600.0, "ppo_update_reduce_exchange: wait_seconds=%g", wait_seconds);
    hipLaunchKernelGGL(ppo_update_reduce_exchange_kernel, dim3((unsigned)groups + 1u), dim3(kRedThreads), 0,
                       (hipStream_t)stream, u, xchg->dev, (long long)(wait_seconds * 1.0e8));
    return check_launch("ppo_update_reduce_exchange");
}

extern "C" int ppoaf_ppo_update_adam(const ppoaf_ppo_update_args_t* args, int compute_norms,
                                     ppoaf_stream_t stream) {
    UpdateDev u;
    int rc = make_update_dev(args, u);
    if (rc) return rc;
    const long n4 = u.bucket_total >> 2;
    const unsigned grid = (unsigned)((n4 + 255) / 256);
    PPOAF_REQUIRE(compute_norms >= 0 && compute_norms <= 3, "ppo_update_adam: compute_norms=%d (0 .. 3)", compute_norms);
    PPOAF_REQUIRE(compute_norms != 3 || u.split, "ppo_update_adam: compute_norms = 3 follows ppoaf_ppo_update_wgrad (split_workspace set)");
    static_assert(kRedThreads == 256, "the reduce launch and the norm pass leave one partial pair per 256 float4 columns");
    if (compute_norms == 1) {
        hipLaunchKernelGGL(ppo_update_sqnorm_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, u);
        rc = check_launch("ppo_update_adam/sqnorm");
        if (rc) return rc;
    }
    // 0 / 1: squared norms = the per-workgroup partials at norm_scratch + 6, added in a fixed association;
    // 2: norm_scratch[0..1] hold them already (ppoaf_peer_exchange_allreduce's norm_out)
    // 3: the partials of ppoaf_ppo_update_wgrad, one pair per workgroup of that launch
    hipLaunchKernelGGL(ppo_update_adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, u,
                       compute_norms == 2 ? (const double*)nullptr : (const double*)(u.norm_scratch + 6),
                       compute_norms == 3 ? (unsigned)split_wgrad_blocks(u) : grid);
    return check_launch("ppo_update_adam");
}

// The N > 1 fallback chain in one call: fwd_bwd -> reduce -> RCCL sum all-reduce of the gradient bucket -> norms +
// Adam, for n consecutive mini-batches.  What `fused_update._eager_multi_rank` does from Python (4 launches + one
// torch.distributed call per mini-batch: ~42 us of host time, more than the kernels take) issued from C: the host cost
// per mini-batch drops below the GPU's.
extern "C" int ppoaf_ppo_update_chain_allreduce(const ppoaf_ppo_update_args_t* args, ppoaf_comm_t* comm, int64_t n_minibatches,
                                                ppoaf_stream_t stream) {
    PPOAF_REQUIRE(args && comm, "ppo_update_chain_allreduce: null argument");
    PPOAF_REQUIRE(n_minibatches >= 1 && n_minibatches <= (1 << 20), "ppo_update_chain_allreduce: n_minibatches=%ld", (long)n_minibatches);
    PPOAF_REQUIRE(args->mb_offset == 0, "ppo_update_chain_allreduce: mb_offset must be 0");
    ppoaf_ppo_update_args_t a = *args;
    for (int64_t j = 0; j < n_minibatches; ++j) {
        a.mb_offset = j;
        a.cursor_advance = (j == n_minibatches - 1) ? n_minibatches : 0;
        int rc = ppoaf_ppo_update_fwd_bwd(&a, stream);
        if (rc == PPOAF_OK) rc = ppoaf_ppo_update_reduce(&a, 0, stream);
        if (rc == PPOAF_OK) rc = ppoaf_allreduce_sum_f32(comm, a.grads, a.bucket_total, stream);
        if (rc == PPOAF_OK) rc = ppoaf_ppo_update_adam(&a, 1, stream);
        if (rc != PPOAF_OK) return rc;
    }
    return PPOAF_OK;
}

extern "C" int ppoaf_ppo_update_adam_exchanged(const ppoaf_ppo_update_args_t* args, ppoaf_peer_exchange_t* xchg,
                                               ppoaf_stream_t stream) {
    UpdateDev u;
    int rc = make_update_dev(args, u);
    if (rc) return rc;
    PPOAF_REQUIRE(xchg && xchg->connected, "ppo_update_adam_exchanged: exchange missing or not connected");
    const long n4 = u.bucket_total >> 2;
    PPOAF_REQUIRE(xchg->dev.n4 == n4, "ppo_update_adam_exchanged: exchange made for %ld float4, bucket has %ld",
                  xchg->dev.n4, n4);
    // one pair of norm partials per exchange group: the workgroups of ppoaf_ppo_update_reduce_exchange, or (split-wgrad
    // chain) of ppoaf_ppo_update_wgrad_exchange
    const unsigned groups = u.split ? (unsigned)split_wgrad_blocks(u) : (unsigned)((n4 + kRedThreads - 1) / kRedThreads);
    PPOAF_REQUIRE(groups <= (unsigned)kXchgMaxGrid, "ppo_update_adam_exchanged: bucket too large for the fused exchange");
    hipLaunchKernelGGL(ppo_update_adam_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u,
                       (const double*)xchg->dev.norm_partials, groups);
    return check_launch("ppo_update_adam_exchanged");
}

#ifdef PPOAF_STAMPS
extern "C" int ppoaf_debug_read_stamps(unsigned long long* out /* host [32] */) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ppo_update_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -2;
}
#endif

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
