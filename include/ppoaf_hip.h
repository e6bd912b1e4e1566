/*
 * ppoaf_hip.h -- C ABI of libppoaf_hip.so, the MI355X (gfx950) hot path that
 * stands in for LLNL/ppo_and_friends' rollout-buffer + PPO-update arithmetic.
 *
 * The reference is pure Python (no FFI of its own); every entry point below
 * replaces a chain of NumPy / PyTorch ops or a Python loop, cited per function
 * as /root/reference-relative file:line.  The Python host (ppo_and_friends_amd/)
 * binds these with ctypes; INTEGRATION.md shows the stub a reference maintainer
 * would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed from the caller (torch tensor
 *     .data_ptr()); the caller owns the memory; nothing is allocated here except
 *     the small per-process scratch noted on individual calls;
 *   - the last argument is the hipStream_t to enqueue on (NULL = default
 *     stream); calls are asynchronous on that stream and never synchronise, so
 *     they are hipGraph-capturable;
 *   - return 0 on success, <0 on error (PPOAF_E_*); ppoaf_last_error() returns
 *     a thread-local message;
 *   - shapes are checked on the host before any launch; no kernel is launched
 *     with a shape it does not handle;
 *   - single Python thread per rank issues the calls; the library keeps no
 *     host threads.
 */
#ifndef PPOAF_HIP_H
#define PPOAF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ppoaf_stream_t;            /* hipStream_t */

#define PPOAF_OK            0
#define PPOAF_E_INVALID    -1            /* bad argument / unsupported shape   */
#define PPOAF_E_LAUNCH     -2            /* hipLaunch / runtime error          */

#define PPOAF_ABI_VERSION 7

int         ppoaf_abi_version(void);
const char* ppoaf_last_error(void);
/* Number of CUs of the current device (used by the host to size grids in tests). */
int         ppoaf_device_cu_count(void);

/* ------------------------------------------------------------------------ *
 * K1  GAE advantages + rewards-to-go
 * replaces EpisodeInfo.end_episode / compute_discounted_sums /
 *          _compute_gae_advantages   utils/episode_info.py:223-293,419-465
 * Numerics: delta = r + f64(fl32(gamma*V[t+1])) - V[t]; both scans accumulate
 * in float64 (the reference's accumulator under its pinned numpy<1.24), results
 * rounded once to float32 (PPODataset.build's float32 cast, :868-887).
 * ending_reward is clipped to [clip_lo, clip_hi] when has_clip (:450-454) and
 * rounded to float32 (:456-457); ending_value is NOT clipped (:410-411).
 * use_gae == 0  ->  advantages = rtg - V (:295-301).
 * ------------------------------------------------------------------------ */

/* Dense time-major rollout buffer [T,E] (E contiguous): the build's native
 * layout.  end_kind[t,e]: 0 = episode continues into t+1, 1 = terminal end
 * (ending value/reward 0, ppo.py:1818-1819), 2 = bootstrapped end (ending
 * value = boot_value[t,e], ending reward = boot_reward[t,e]; ppo.py:1932-1938).
 * Row T-1 must be non-zero everywhere (ppo.py:1870-1871) -- when end_kind is
 * NULL the trajectories are fixed-length: only row T-1 ends, bootstrapped, and
 * boot_value / boot_reward are [E] arrays.  Otherwise they are [T,E] and read
 * only where end_kind == 2. */
int ppoaf_gae_rtg_tmajor(const float* rewards, const float* values,
                         const float* boot_value, const float* boot_reward,
                         const int8_t* end_kind,
                         int32_t T, int64_t E,
                         double gamma, double lambd,
                         int has_clip, double clip_lo, double clip_hi,
                         int use_gae,
                         float* adv_out, float* rtg_out,
                         ppoaf_stream_t stream);

/* Same launch with the kernel's own begin / end stamped into two events (hipExtLaunchKernelGGL):
 * what bench.py uses for the roofline figure.  Events come from ppoaf_event_create(). */
int ppoaf_gae_rtg_tmajor_timed(const float* rewards, const float* values,
                               const float* boot_value, const float* boot_reward,
                               const int8_t* end_kind,
                               int32_t T, int64_t E,
                               double gamma, double lambd,
                               int has_clip, double clip_lo, double clip_hi,
                               int use_gae,
                               float* adv_out, float* rtg_out,
                               void* start_event, void* stop_event,
                               ppoaf_stream_t stream);
void* ppoaf_event_create(void);
int   ppoaf_event_destroy(void* event);
/* waits for `stop`, then *ms = time between the two stamps */
int   ppoaf_event_elapsed_ms(void* start_event, void* stop_event, float* ms);

/* Ragged trajectory list over a flat episode-major [N] layout (the layout
 * PPODataset.build produces, utils/episode_info.py:745-914): trajectory i
 * covers [traj_start[i], traj_start[i] + traj_len[i]).  Used for the
 * EpisodeInfo-level drop-in and PPODataset.recalculate_advantages (:721-743;
 * pass rtg_out = NULL there: rtg is not recomputed by the reference). */
int ppoaf_gae_rtg_traj(const float* rewards, const float* values,
                       const float* ending_value, const float* ending_reward,
                       const int64_t* traj_start, const int32_t* traj_len,
                       int64_t n_traj,
                       double gamma, double lambd,
                       int has_clip, double clip_lo, double clip_hi,
                       int use_gae,
                       float* adv_out, float* rtg_out,
                       ppoaf_stream_t stream);

/* ------------------------------------------------------------------------ *
 * K2+K3  mini-batch advantage normalisation + PPO loss, forward and backward
 * replaces PPO._ppo_batch_train  ppo.py:2325-2333 (normalise), 2352-2438 (loss)
 *
 * Inputs are length-B float32 streams.  adv is the RAW advantage; when
 * normalize_adv it is replaced by (adv-mean)/(unbiased_std+1e-8) (ppo.py:2326-2333).
 * Outputs:
 *   scalars[0] actor surrogate loss  mean(-min(surr1,surr2))        ppo.py:2392
 *   scalars[1] total actor loss = [0] - w_ent*mean(H) + kl_w*kl     ppo.py:2395-2405
 *   scalars[2] critic loss (MSE, or Huber(delta) if use_huber)      ppo.py:2416-2419
 *   scalars[3] mean entropy                                         ppo.py:2396
 *   scalars[4] kl = mean(old_logp - cur_logp)                       ppo.py:2358
 *   scalars[5] advantage mean, scalars[6] advantage unbiased std
 *   scalars[7] 1.0 if any ratio is NaN/Inf (ppo.py:2361), else 0.0
 *   d_logp[B], d_entropy[B]  = d scalars[1] / d cur_logp, / d entropy
 *   d_values[B]              = d scalars[2] / d values
 * The kl_loss_weight term carries no gradient (ppo.py:2403-2405 adds a Python
 * float).  The entropy term is skipped when entropy_weight == 0 (ppo.py:2395).
 * The vf_clip branch of the reference raises AttributeError (ppo.py:2432) and
 * is not reproduced.  Any gradient pointer may be NULL (forward only).
 * ------------------------------------------------------------------------ */
int ppoaf_ppo_loss_fwd_bwd(const float* cur_logp, const float* old_logp,
                           const float* adv, const float* entropy,
                           const float* values, const float* rtg,
                           int64_t B,
                           int normalize_adv, float surr_clip,
                           float entropy_weight, float kl_loss_weight,
                           int use_huber, float huber_delta,
                           float* scalars /* [8] */,
                           float* d_logp, float* d_entropy, float* d_values,
                           ppoaf_stream_t stream);

/* ------------------------------------------------------------------------ *
 * K4  mini-batch gather
 * replaces DataLoader(shuffle=True) + PPODataset.__getitem__ + default collate
 *          ppo.py:2181-2184,2292-2295; utils/episode_info.py:922-952
 * For b in [0,B): row = perm[b]; if row_map: row = row_map[row];
 *                 dst_f[b, :] = src_f[row, :]  for every field f.
 * row_map maps a position in the reference's flattened episode-major order to
 * the row of the time-major [T*E] buffer (NULL = identity).  Up to
 * PPOAF_MAX_GATHER_FIELDS fields per launch; row_bytes must be a multiple of 4.
 * ------------------------------------------------------------------------ */
#define PPOAF_MAX_GATHER_FIELDS 8
typedef struct {
    const void* src;        /* [n_rows, row_bytes] */
    void*       dst;        /* [B, row_bytes]      */
    int32_t     row_bytes;
    int32_t     _pad;
} ppoaf_gather_field_t;

int ppoaf_minibatch_gather(const ppoaf_gather_field_t* fields /* host array */,
                           int32_t n_fields,
                           const int64_t* perm, const int32_t* row_map,
                           int64_t n_rows, int64_t B,
                           ppoaf_stream_t stream);

/* Scatter of the freshly evaluated critic values back into the dataset
 * (ppo.py:2340  dataset.values[batch_idxs] = values.detach()). */
int ppoaf_scatter_rows_f32(const float* src /* [B] */, const int64_t* perm,
                           const int32_t* row_map, int64_t n_rows, int64_t B,
                           float* dst /* [n_rows] */, ppoaf_stream_t stream);

/* ------------------------------------------------------------------------ *
 * K5  running mean / variance (value, observation and reward normalisers)
 * replaces RunningMeanStd.update/_integrate_batch_data  utils/stats.py:29-94
 *          RunningStatNormalizer.normalize/denormalize  utils/misc.py:84-128
 * State lives on the device: mean[W], var[W] float32, count[1] float64
 * (initialise mean 0, var 1, count 1e-4: stats.py:25-27).
 * ------------------------------------------------------------------------ */

/* moments_out[0] = n, [1..W] = batch mean, [1+W..2W] = batch M2 = sum (x-mean)^2,
 * all float64; data is [n, W] float32 row-major. */
int ppoaf_batch_moments(const float* data, int64_t n, int32_t W,
                        double* moments_out /* [1+2W] */, ppoaf_stream_t stream);

/* Merge R per-rank moment records (concatenated, [R, 1+2W]; R = 1 on a single
 * rank; more after an all-gather -- equal to the reference's allgather of the
 * raw data, stats.py:47-50) into one batch (Chan), then integrate that batch
 * into the running state exactly as stats.py:73-94 does. */
int ppoaf_running_moments_integrate(const double* moments, int32_t R, int32_t W,
                                    float* mean, float* var, double* count,
                                    ppoaf_stream_t stream);

/* out = (x - mean) / sqrt(var + eps)   misc.py:106-111 ; x is [n, W] */
int ppoaf_normalize(const float* x, int64_t n, int32_t W,
                    const float* mean, const float* var, float eps,
                    float clip_lo, float clip_hi, int has_clip,
                    float* out, ppoaf_stream_t stream);
/* out = mean + x * sqrt(var + eps)     misc.py:124-128 */
int ppoaf_denormalize(const float* x, int64_t n, int32_t W,
                      const float* mean, const float* var, float eps,
                      float* out, ppoaf_stream_t stream);

/* ------------------------------------------------------------------------ *
 * K6  action distributions
 * replaces CategoricalDistribution / GaussianDistribution
 *          networks/distributions.py:199-269, 441-694 and
 *          PPOPolicy.get_rollout_actions / evaluate  policies/ppo_policy.py:729-794,891-952
 * Categorical(probs) semantics of torch.distributions: probs renormalised by
 * their sum, logits = log(clamp(probs, eps, 1-eps)) with eps = FLT_EPSILON.
 * ------------------------------------------------------------------------ */

/* Rollout: logits [n,K] -> softmax -> sample (Philox4x32-10 keyed by seed,
 * counter = offset + row) -> log-prob of the sample.  probs_out may be NULL. */
int ppoaf_categorical_sample(const float* logits, int64_t n, int32_t K,
                             uint64_t seed, uint64_t offset,
                             int64_t* action_out, float* logp_out,
                             float* probs_out, ppoaf_stream_t stream);

/* Update: logits [n,K], actions [n] -> logp[n], entropy[n] (+ probs for bwd). */
int ppoaf_categorical_eval_fwd(const float* logits, const int64_t* actions,
                               int64_t n, int32_t K,
                               float* logp_out, float* entropy_out,
                               float* probs_out, ppoaf_stream_t stream);
/* d_logits[n,K] from d_logp[n], d_entropy[n] (either may be NULL = zeros). */
int ppoaf_categorical_eval_bwd(const float* probs, const int64_t* actions,
                               const float* d_logp, const float* d_entropy,
                               int64_t n, int32_t K,
                               float* d_logits, ppoaf_stream_t stream);

/* Gaussian with tanh squashing (distributions.py:441-694):
 *   std = max(softplus(log_std), min_std)                         :514-516
 *   logp = sum_d clamp(N(mean,std).log_prob(x), -100, 100)
 *          - sum_d log(clamp(1 - tanh(x)^2, 1e-6, inf))            :551-558
 *   entropy = -logp(x := mean): PPOPolicy.evaluate asks for the entropy at the
 *             distribution's own mean                  ppo_policy.py:950, :672-694
 * mean, x: [n,D]; log_std: [D]. */
int ppoaf_gaussian_tanh_eval_fwd(const float* mean, const float* log_std,
                                 const float* x, int64_t n, int32_t D,
                                 float min_std,
                                 float* logp_out, float* entropy_out,
                                 ppoaf_stream_t stream);
/* d_mean[n,D], d_log_std[D] (the sum over all rows, in a fixed order: one launch, no memset, no atomics -- safe
 * inside a captured hipGraph) from d_logp[n] and d_entropy[n]. */
int ppoaf_gaussian_tanh_eval_bwd(const float* mean, const float* log_std,
                                 const float* x, const float* d_logp,
                                 const float* d_entropy, int64_t n, int32_t D,
                                 float min_std,
                                 float* d_mean, float* d_log_std,
                                 ppoaf_stream_t stream);
/* Rollout: raw = mean + std * N(0,1) (Philox), action = tanh(raw) rescaled to
 * [act_lo[d], act_hi[d]] per action dimension (device float32[D]; both NULL: the
 * unit box, no rescale) (:476-483, :580-609, :645-672), logp as above. */
int ppoaf_gaussian_tanh_sample(const float* mean, const float* log_std,
                               int64_t n, int32_t D, float min_std,
                               const float* act_lo, const float* act_hi,
                               uint64_t seed, uint64_t offset,
                               float* raw_out, float* action_out,
                               float* logp_out, ppoaf_stream_t stream);

/* ------------------------------------------------------------------------ *
 * K11  flat-bucket gradient clip + Adam
 * replaces PPOPolicy.update_weights' clip_grad_norm_ + Adam.step
 *          policies/ppo_policy.py:1037-1042,1050-1055 (Adam eps 1e-5 :336-339)
 *          and the 1/num_procs of mpi_avg  utils/mpi_utils.py:86
 * params/grads/exp_avg/exp_avg_sq are flat float32 buckets of n elements.
 * grad_scale multiplies the gradient first (1/world_size after a SUM
 * all-reduce).  max_norm <= 0 disables clipping; otherwise
 * coef = min(1, max_norm / (||grad_scale*g||_2 + 1e-6)) as torch does.
 * step_count is the 1-based Adam step kept on the device (int64[1],
 * incremented by the call) so the launch is graph-replayable.
 * norm_scratch: float64[PPOAF_NORM_SCRATCH_DOUBLES] device scratch owned by the caller: [0] receives the squared
 * norm, [2..] per-workgroup partials.  No atomics anywhere: the partials are added in a fixed association, so the
 * clip coefficient is bitwise the same in every run and on every rank of a DD-PPO job.
 * ------------------------------------------------------------------------ */
#define PPOAF_NORM_SCRATCH_DOUBLES 66
int ppoaf_clip_adam_step(float* params, const float* grads,
                         float* exp_avg, float* exp_avg_sq, int64_t n,
                         int64_t* step_count, const float* lr /* device [1] */,
                         float beta1, float beta2, float eps,
                         float grad_scale, float max_norm,
                         double* norm_scratch, float* grad_norm_out /* NULL ok */,
                         ppoaf_stream_t stream);
/* The Adam half alone: step_count has been advanced for this step and ||grad_scale * grads||^2 is known -- as
 * norm_scratch[0] (n_norm_partials = 0: e.g. ppoaf_peer_exchange_allreduce's norm_out), or as n_norm_partials
 * per-workgroup partials norm_scratch[2 ..] left by a producer kernel (K15's reduce with fuse_norm), which are added
 * in a fixed association here (and the sum stored to norm_scratch[0]). */
int ppoaf_adam_step_prenormed(float* params, const float* grads,
                              float* exp_avg, float* exp_avg_sq, int64_t n,
                              const int64_t* step_count, const float* lr /* device [1] */,
                              float beta1, float beta2, float eps,
                              float grad_scale, float max_norm,
                              double* norm_scratch, int32_t n_norm_partials, float* grad_norm_out /* NULL ok */,
                              ppoaf_stream_t stream);


/* ------------------------------------------------------------------------ *
 * K12  fused PPO mini-batch update for MLP actor/critic (discrete or
 *      tanh-Gaussian head): the whole body of one iteration of
 *      PPO._ppo_batch_train  ppo.py:2292-2469  in three launches
 *
 *   ppoaf_ppo_update_fwd_bwd : grid (ceil(B/16), 2 networks).  Each workgroup
 *        gathers its 16 rows through perm / row_map (K4), the critic side
 *        integrates this mini-batch's value-normaliser record and normalises
 *        rewards-to-go (K5, ppo.py:2299-2303), the actor side normalises the
 *        advantages over the whole mini-batch (K2, :2325-2333); forward through
 *        the MLP (hidden layers on f32 MFMA 16x16x4, exact fmaf chains),
 *        distribution log-prob / entropy (K6), loss terms (K3), dataset.values
 *        write-back (:2340), backward through the MLP; weight-gradient partials
 *        go to this workgroup's slab with plain stores (no atomics: the sum
 *        order is fixed, results are bitwise reproducible).
 *   ppoaf_ppo_update_reduce  : slabs -> flat gradient bucket (fixed order),
 *        per-workgroup loss partials -> running totals; optionally the squared
 *        gradient norms (single rank).  An all-reduce of the bucket goes
 *        between this and the next call on multi-rank runs.
 *   ppoaf_ppo_update_adam    : [norms if not done] clip + Adam for both
 *        networks (ppo_policy.py:1037-1055), advances the mini-batch cursor.
 *        compute_norms 0: the reduce launch (compute_norms = 1 there) left the
 *        per-workgroup squared-norm partials in norm_scratch[6 ..]; 1: compute
 *        those partials now (after an all-reduce of the bucket); 2:
 *        norm_scratch[0..1] already hold the two squared norms
 *        (ppoaf_peer_exchange_allreduce's norm_out).  Partials are added in a
 *        fixed association -- no atomics: the clip coefficients are bitwise the
 *        same in every run and on every rank.
 *
 * All three read the mini-batch index from a device cursor, so one captured
 * hipGraph of N repetitions walks N consecutive mini-batches of the epoch.
 * Network parameters are the flat bucket of nn.Linear tensors in module order
 * (weight [out,in] row-major then bias, each padded to 4 floats), hidden
 * layers of equal width H per network; instantiated (actor, critic) widths:
 * (32,32) (64,64) (128,128) (256,256) (128,256) (64,128); depth >= 1.
 * ------------------------------------------------------------------------ */
#define PPOAF_ACT_RELU        0
#define PPOAF_ACT_LEAKY_RELU  1      /* negative slope 0.01 (nn.LeakyReLU default) */
#define PPOAF_ACT_TANH        2
#define PPOAF_HEAD_CATEGORICAL 0
#define PPOAF_HEAD_GAUSSIAN    1
#define PPOAF_UPDATE_ROWS_PER_WG 16

typedef struct {
    int32_t in_dim, hidden, depth, out_dim;   /* depth = number of hidden layers */
    int32_t activation, _pad;
    int64_t offset;        /* first float of this network inside the policy bucket */
    int64_t size;          /* floats of this network's bucket (incl. padding + log_std) */
    int64_t log_std_offset;/* actor, Gaussian head: offset of log_std [out_dim] inside the bucket, else -1 */
} ppoaf_mlp_desc_t;

typedef struct {
    ppoaf_mlp_desc_t actor, critic;
    /* parameters / optimiser state: one bucket, actor then critic */
    float* params; float* grads; float* exp_avg; float* exp_avg_sq;
    float* slabs;                 /* [n_wg, bucket_total] gradient partials               */
    int64_t bucket_total;
    int64_t* step_counts;         /* [2] Adam steps (actor, critic), incremented per call  */
    const float* lr;              /* [1] device                                            */
    double* norm_scratch;         /* [6 + 2 * ceil(bucket_total / 1024)]: squared norms (2) + Adam bias corrections (4)
                                   * + per-workgroup squared-norm partials (actor, critic) of the reduce / norm pass */
    float beta1, beta2, adam_eps, grad_scale, max_norm; int32_t head_kind;
    /* rollout buffer (time-major rows) */
    const float* obs; const float* critic_obs; const void* raw_actions;  /* int64 [n,1] or f32 [n,D] */
    const float* advantages; const float* old_log_probs; const float* rewards_to_go;
    float* values;                /* write-back target                                     */
    const int64_t* perm; const int32_t* row_map; int64_t n_rows;
    int64_t* cursor;              /* [1] mini-batch index within the epoch                 */
    int64_t B;                    /* rows of this mini-batch (<= batch capacity)           */
    int64_t batch_stride;         /* perm offset of mini-batch k = k * batch_stride        */
    /* value normaliser: double-buffered state (slot = cursor & 1), records [n_batches, R, 3] */
    int32_t normalize_values, n_ranks;
    float* vn_mean; float* vn_var; double* vn_count;    /* [2] each */
    const double* vn_records;
    /* advantage normalisation (ppo.py:2326-2333): (n, mean, M2) of the raw advantages of every
     * mini-batch of the epoch, [n_batches, 3] from ppoaf_minibatch_moments (rank-local) */
    const double* adv_records;
    /* loss */
    int32_t normalize_adv, use_huber;
    float surr_clip, entropy_weight, kl_loss_weight, huber_delta, min_std;
    /* 1: obs / critic_obs / raw_actions / advantages / old_log_probs / rewards_to_go are per-epoch
     * tables already in shuffled order (entry i belongs to perm[i]), so a mini-batch reads
     * [k * batch_stride, +B) directly -- no index -> data dependent load; `values` is still
     * written through perm / row_map.  0: they are the rollout buffer's rows. */
    int32_t inputs_in_batch_order;
    float* loss_partials;         /* [2, n_wg, 8]                                          */
    double* totals;               /* [9] sums of the 8 loss scalars + mini-batch count     */
    /* mini-batch index = *cursor + mb_offset; the Adam launch adds cursor_advance to *cursor.
     * Eager use: (0, 1).  A captured chain of n mini-batches bakes mb_offset = 0..n-1 into its nodes
     * and advances the cursor once, by n, in the last one: the cursor word is then rewritten once per
     * chain instead of once per mini-batch, so all but the first read of it hit the reader's L2. */
    int64_t mb_offset, cursor_advance;
    /* split-wgrad chain (NULL: weight-gradient slabs + ppoaf_ppo_update_reduce).  With a workspace of
     * ppoaf_ppo_update_split_workspace_bytes() bytes (256-byte aligned) fwd_bwd computes no hidden-layer weight gradient:
     * every workgroup publishes its 16 rows of the inputs, hidden activations and dLoss/dz of the mini-batch there, and
     * ppoaf_ppo_update_wgrad forms the complete gradients from those panels (then ppoaf_ppo_update_adam, compute_norms 3). */
    void* split_workspace; int64_t split_workspace_bytes;
    /* 0: fwd_bwd's workgroups use every XCD (actor on 0-3, critic on 4-7; default).  1 / 2: they sit on XCDs 0-3 / 4-7 only
     * (actor on the half's first two XCDs, critic on the other two; the launch is twice as wide and the workgroups
     * dispatched to the other half return at once): when another update chain (K14: ppoaf_icm_update_args_t.xcd_half)
     * runs on a second stream, each keeps its weights and panels in its own four L2s -- C3: +2 % env-steps/s; alone the
     * confinement costs 1 % (C2).  Placement only changes speed. */
    int32_t xcd_half;
    /* ABI 6.  1 (split-wgrad chain only; ignored otherwise): the 16-row tiles of a 256-wide network of depth 2 .. 4 run on
     * PAIRS of workgroups that split every hidden layer pass by output columns and exchange the halves through tagged
     * records behind the panels of split_workspace (csrc/ppo_update_rowpair.hpp) -- ppoaf_ppo_update_split_workspace_bytes()
     * then includes the record region.  Results are BITWISE those of row_pairs = 0.  The tag is the mini-batch index + 1:
     * the caller zeroes split_workspace before the first launch and whenever mini-batch indices restart (every epoch).
     * The region's first 32-bit word (byte offset ppoaf_ppo_update_row_pairs_error_offset()) is non-zero after a launch in
     * which a partner did not answer within 2 s; that launch's results are invalid. */
    int32_t row_pairs;
} ppoaf_ppo_update_args_t;

int ppoaf_ppo_update_fwd_bwd(const ppoaf_ppo_update_args_t* args, ppoaf_stream_t stream);
/* same launch with the kernel's own begin / end stamped into two events (as ppoaf_gae_rtg_tmajor_timed) */
int ppoaf_ppo_update_fwd_bwd_timed(const ppoaf_ppo_update_args_t* args, void* start_event, void* stop_event,
                                   ppoaf_stream_t stream);
int ppoaf_ppo_update_reduce(const ppoaf_ppo_update_args_t* args, int compute_norms, ppoaf_stream_t stream);
/* Split-wgrad chain: fwd_bwd (args->split_workspace set) -> ppoaf_ppo_update_wgrad -> ppoaf_ppo_update_adam(compute_norms 3).
 * The weight gradients of ppo.py:2443 (loss.backward()) are formed ONCE per mini-batch over all B rows -- one workgroup
 * per 16 x 16 tile of dW_l = dz_l^T h_{l-1} on f32 MFMA with K = B, db_l as column sums of dz_l, the output layer's
 * per-block partials folded in block order -- instead of 16-row partials written to B/16 slabs of the whole bucket and
 * summed by ppoaf_ppo_update_reduce: 8.7 MB of slab traffic per mini-batch at C2 become 1.6 MB of panels.  The same
 * launch folds the loss partials into totals, advances the step counters and leaves one pair of squared-norm partials
 * per workgroup in norm_scratch[6 ..] (needs 6 + 2 * ppoaf_ppo_update_split_blocks(args) doubles).  Sums run in MFMA K
 * order: float32-rounding-level differences from the slab chain, bitwise reproducible run to run. */
int ppoaf_ppo_update_split_workspace_bytes(const ppoaf_ppo_update_args_t* args, int64_t* bytes_out);
int ppoaf_ppo_update_split_blocks(const ppoaf_ppo_update_args_t* args);
/* byte offset of the row-pair error word inside split_workspace; -1 in *offset_out when args (row_pairs, shapes) select no pairs */
int ppoaf_ppo_update_row_pairs_error_offset(const ppoaf_ppo_update_args_t* args, int64_t* offset_out);
int ppoaf_ppo_update_wgrad(const ppoaf_ppo_update_args_t* args, ppoaf_stream_t stream);
/* Fused tail of the split-wgrad chain (ABI 5; csrc/ppo_update_tail.hip): fwd_bwd (args->split_workspace set) ->
 * ppoaf_ppo_update_wgrad_adam -- TWO launches per mini-batch.  Everything from loss.backward()'s weight gradients to
 * optimizer.step() (ppo.py:2443-2444, ppo_policy.py:1032-1055: backward, clip_grad_norm_, Adam, both networks) in one
 * launch: a workgroup forms one 16 x 32 piece of a layer's dW over all B rows (ppoaf_ppo_update_wgrad's jobs, bit for bit),
 * publishes its squared-norm partial as one tagged 16-byte record, waits for the records of all workgroups, and applies
 * clip + Adam to exactly its elements (their optimiser state was requested beside the MFMA operands).  Parameters, moments,
 * gradient bucket, totals, step counters and cursor end BITWISE as after ppoaf_ppo_update_wgrad + ppoaf_ppo_update_adam(3).
 * ctl: ppoaf_ppo_update_tail_ctl_bytes() bytes of device memory, 64-byte aligned, zeroed ONCE by the caller and then
 * kept across launches (it carries the launch tag); its third 32-bit word is non-zero after a launch in which a wait
 * ran out of wait_seconds (the workgroups were not all resident: the results of that launch are then invalid and later
 * launches do not wait again).  All 8 * per_xcd + 1 workgroups must fit on the device together (checked). */
int ppoaf_ppo_update_tail_ctl_bytes(const ppoaf_ppo_update_args_t* args, int64_t* bytes_out);
int ppoaf_ppo_update_wgrad_adam(const ppoaf_ppo_update_args_t* args, void* ctl, double wait_seconds, ppoaf_stream_t stream);
/* same launch with the kernel's own begin / end stamped into two events */
int ppoaf_ppo_update_wgrad_adam_timed(const ppoaf_ppo_update_args_t* args, void* ctl, double wait_seconds,
                                      void* start_event, void* stop_event, ppoaf_stream_t stream);
int ppoaf_ppo_update_adam(const ppoaf_ppo_update_args_t* args, int compute_norms, ppoaf_stream_t stream);


/* ------------------------------------------------------------------------ *
 * K6+K7  one rollout step for all E envs of the rank in one launch
 * replaces PPOPolicy.get_rollout_actions            policies/ppo_policy.py:729-794
 *          PPO.get_policy_values + denormalisation  ppo.py:1030-1075, utils/misc.py:113-128
 *          the per-env EpisodeInfo.add_info calls   policies/ppo_policy.py:638-651
 * Actor and critic MLP forward (same tile code as the update kernel), sampling
 * (Philox4x32-10, counter = offset + env), log-prob, value (denormalised with the
 * running stats when normalize_values), written straight into row t of the
 * rollout buffer.  obs / critic_obs: [E, in_dim] float32; outputs are the row-t
 * slices of the buffer ([E,1] int64 actions for the categorical head, [E,D]
 * float32 for the Gaussian head).  *_copy_out (optional) receive the
 * observation rows.  Network descriptors as for K12.
 * ------------------------------------------------------------------------ */
typedef struct {
    ppoaf_mlp_desc_t actor, critic;
    const float* params;
    const float* obs; const float* critic_obs; int64_t E;
    int32_t head_kind; float min_std;
    const float* act_lo; const float* act_hi;   /* Gaussian head: bounds per action dimension (device float32[D]); both NULL = [-1,1] */
    const void* forced_raw_action;              /* NULL: sample.  Else the raw actions to log instead of sampling
                                                   ([E] int64 / [E,D] float32): replay of a recorded rollout */
    uint64_t seed, offset;
    int32_t normalize_values, _pad;
    const float* vn_mean; const float* vn_var;
    void* raw_action_out; void* action_out; float* logp_out; float* value_out;
    float* obs_copy_out; float* critic_obs_copy_out;
} ppoaf_policy_step_args_t;

int ppoaf_policy_step(const ppoaf_policy_step_args_t* args, ppoaf_stream_t stream);


/* ------------------------------------------------------------------------ *
 * K8  ICM forward-model loss and intrinsic reward
 * replaces the tail of ICM.forward  networks/ppo_networks/icm.py:421-430
 *   intr[i]  = (reward_scale / 2) * sum_d (pred[i,d] - enc2[i,d])^2
 *   f_loss   = 0.5 * mean((pred - enc2)^2)
 * pred, enc2: [n, D] float32; rowsum_scratch: float32[n] owned by the caller;
 * intr_out / f_loss_out may be NULL.  bwd: d_pred = grad_f_loss * (pred-enc2)/(n*D),
 * d_enc2 = -d_pred (NULL to skip); grad_f_loss is a device scalar.
 * ------------------------------------------------------------------------ */
int ppoaf_icm_forward_loss_fwd(const float* pred, const float* enc2, int64_t n, int32_t D,
                               float reward_scale, float* rowsum_scratch, float* intr_out,
                               float* f_loss_out, ppoaf_stream_t stream);
int ppoaf_icm_forward_loss_bwd(const float* pred, const float* enc2, int64_t n, int32_t D,
                               const float* grad_f_loss, float* d_pred, float* d_enc2,
                               ppoaf_stream_t stream);


/* ------------------------------------------------------------------------ *
 * K9  multi-agent-transformer attention core (f32 MFMA)
 * replaces the middle of SelfAttention.forward  networks/attention.py:94-103
 *   att = softmax(q k^T / sqrt(D) [causal mask]);  y = att v
 * q, k, v, y, dy, dq, dk, dv: [n_seq, L, D] float32 contiguous (heads folded into
 * n_seq), L <= 16 agents, D a multiple of 16; probs: [n_seq, L, L] (saved by fwd,
 * consumed by bwd).  floor(16/L) sequences are packed per 16-row MFMA tile.
 * ------------------------------------------------------------------------ */
int ppoaf_mat_attention_fwd(const float* q, const float* k, const float* v, int64_t n_seq,
                            int32_t L, int32_t D, int masked, float* y_out, float* probs_out,
                            ppoaf_stream_t stream);
int ppoaf_mat_attention_bwd(const float* q, const float* k, const float* v, const float* probs,
                            const float* dy, int64_t n_seq, int32_t L, int32_t D,
                            float* dq, float* dk, float* dv, ppoaf_stream_t stream);

/* (n, mean, M2) float64 records of the rewards-to-go of every mini-batch of an
 * epoch: records[k] covers perm[k*B : min((k+1)*B, n_perm)]  (ppo.py:2299-2303,
 * utils/stats.py:52-54 batched).  One workgroup per mini-batch. */
int ppoaf_minibatch_moments(const float* data, const int64_t* perm, const int32_t* row_map,
                            int64_t n_perm, int64_t B, double* records /* [ceil(n/B), 3] */,
                            ppoaf_stream_t stream);

/* ------------------------------------------------------------------------ *
 * K13  environment filters: running observation / reward normalisation and clipping
 * replaces ObservationNormalizer.step/reset        environments/filter_wrappers.py:155-268
 *          RewardNormalizer.step                   environments/filter_wrappers.py:388-458
 *          ObservationClipper / RewardClipper      environments/filter_wrappers.py:583-719
 * in the wiring order of wrapper_utils.py:81-111 (normalise, then clip).
 *
 * One env step of G agents x n envs (rows agent-major: group g owns rows [g*n, (g+1)*n)).
 * Two launches: `moments` reduces this rank's batch to a float64 record, the records of the
 * R ranks are all-gathered by the caller (R = 1: pass the record itself), `apply` merges
 * them into the running state and writes the filtered outputs.
 *
 * Record layout (float64, PPOAF_ENV_FILTER_RECORD_LEN(G, W_o, W_c, has_reward) values):
 *   [0] n | obs mean[G*W_o] | obs M2[G*W_o] | critic mean[G*W_c] | critic M2[G*W_c]
 *       | reward S1[G] | reward S2[G]
 * Observation stats are per (agent, feature): mean/var float32, count float64 (one copy per
 * column so no column races on it; initialise 0 / 1 / 1e-4, stats.py:25-27).
 * Reward stats per agent are float64 (the reference's become float64 on the first update).
 * The reference updates the reward stats n times per step, once after each env's running reward
 * is advanced, each time with the whole half-updated vector (filter_wrappers.py:412-418); the
 * n Chan merges equal one merge of the pooled n*n values, which is what S1/S2 carry:
 *   S1 = sum_i (n-i)(new_i - m) + i (old_i - m),  S2 likewise with squares, m = running mean.
 * A NULL filter pointer skips that stream.
 * ------------------------------------------------------------------------ */
typedef struct {
    const float* x;        /* [G, n, W] raw */
    float*       out;      /* [G, n, W] filtered (may alias x) */
    float*       mean;     /* [G, W] */
    float*       var;      /* [G, W] */
    double*      count;    /* [G, W] */
    int32_t      W;
    int32_t      normalize;  /* 0: clip only (mean/var/count may be NULL) */
    int32_t      update;     /* integrate this step into the running stats */
    int32_t      has_clip;
    float        clip_lo, clip_hi;
    float        eps;        /* 1e-8 */
} ppoaf_obs_filter_t;

typedef struct {
    const float*   reward;          /* [G, n] raw */
    const uint8_t* done;            /* [G, n] terminated */
    const uint8_t* done2;           /* [G, n] truncated, OR-ed with `done`; may be NULL */
    float*         out;             /* [G, n] */
    double*        running_reward;  /* [G, n] discounted running reward (state) */
    double*        mean;            /* [G] */
    double*        var;             /* [G] */
    double*        count;           /* [G] */
    int32_t        normalize;       /* 0: clip only (state pointers may be NULL) */
    int32_t        update;
    int32_t        has_clip;
    float          clip_lo, clip_hi;
    double         gamma;
    double         eps;             /* 1e-8 */
} ppoaf_reward_filter_t;

#define PPOAF_ENV_FILTER_RECORD_LEN(G, W_o, W_c, has_reward) \
    (1 + 2 * (G) * ((W_o) + (W_c) + ((has_reward) ? 1 : 0)))

int ppoaf_env_filter_moments(const ppoaf_obs_filter_t* obs, const ppoaf_obs_filter_t* critic_obs,
                             const ppoaf_reward_filter_t* reward, int32_t G, int64_t n,
                             double* record, ppoaf_stream_t stream);
int ppoaf_env_filter_apply(const ppoaf_obs_filter_t* obs, const ppoaf_obs_filter_t* critic_obs,
                           const ppoaf_reward_filter_t* reward, int32_t G, int64_t n,
                           const double* records /* [R, record_len] */, int32_t R,
                           ppoaf_stream_t stream);

/* ------------------------------------------------------------------------ *
 * K14  fused ICM mini-batch update
 * replaces one iteration of PPO._icm_batch_train   ppo.py:2487-2567
 *          ICM.forward (encoder x2, inverse model, forward model, both losses)
 *                                                  networks/ppo_networks/icm.py:375-430
 *          LinearObservationEncoder.forward        networks/encoders.py:40-56
 *          mpi_avg_gradients + Adam (no clipping)  ppo.py:2556-2562
 * for the reference's default ICM topology with one width H everywhere:
 *   encoder  O -> H -> H -> H -> H (last layer linear)
 *   inverse  2H -> H (x depth_inv hidden layers) -> A          (softmax + CE for discrete,
 *                                                               MSE for continuous actions)
 *   forward  H + A_in -> H (x depth_fwd hidden layers) -> H     (A_in = classes / action dims)
 * Parameters live in one flat bucket in module order (weight, bias per Linear, each
 * padded to 4 floats): encoder at enc_offset, then inverse, then forward model.
 *
 * `fwd_bwd` enqueues three launches -- encoder forward for both observations (activations
 * to `act_scratch`), inverse / forward model forward + losses + backward (encoding
 * gradients to `denc_scratch`), encoder backward -- each on 2 * ceil(B/16) workgroups of
 * 16 rows, every HxH layer on v_mfma_f32_16x16x4_f32; weight gradients go to per-workgroup
 * slabs.  `reduce` sums the slabs in a fixed order into `grads`, folds the loss into
 * totals[0] (icm_loss = (1-beta) f_loss + beta inv_loss) / totals[1] (count), advances
 * the cursor and, with fused_adam, applies Adam in the same launch (the single-rank path;
 * with more ranks the caller all-reduces `grads` and runs K11).
 * Rows of mini-batch k are perm[k*batch_stride + 0..B) with k read from `cursor`.
 * ------------------------------------------------------------------------ */
typedef struct {
    int32_t obs_dim, hidden, action_dim, fwd_action_dim, depth_inv, depth_fwd, activation, discrete;
    int64_t enc_offset, inv_offset, fwd_offset, bucket_total;
    const float* params; float* grads; float* exp_avg; float* exp_avg_sq;
    float* slabs;                    /* [2*ceil(B/16), bucket_total]                           */
    int64_t* step_count;             /* [1] Adam step counter (advanced when fused_adam)        */
    const float* lr;                 /* [1] device scalar                                       */
    float beta1, beta2, adam_eps, grad_scale;
    const float* obs;                /* [n_rows, obs_dim]                                       */
    const float* next_obs;           /* [n_rows, obs_dim]                                       */
    const void* actions;             /* int64 [n_rows] (discrete) / float32 [n_rows, action_dim]*/
    const int64_t* perm; const int32_t* row_map; int64_t n_rows;
    int64_t* cursor; int64_t B, batch_stride;
    float icm_beta; int32_t fused_adam;
    float* act_scratch;              /* [2, 4, 16*ceil(B/16), hidden]                           */
    float* denc_scratch;             /* [2, 2, 16*ceil(B/16), hidden]                           */
    float* loss_partials;            /* [ceil(B/16) + 1, 2] (last row: this step's Adam constants)*/
    double* totals;                  /* [2]                                                     */
    /* 1: obs / next_obs / actions are per-epoch tables already in shuffled order (entry i belongs to
     * perm[i]); mini-batch k reads [k * batch_stride, +B) directly, perm / row_map are not consulted */
    int32_t inputs_in_batch_order, _pad;
    /* split-wgrad chain (NULL: every workgroup writes weight-gradient slabs of the bucket).  With a workspace of
     * ppoaf_icm_update_split_workspace_bytes() bytes (256-byte aligned) the three fwd_bwd kernels form NO weight
     * gradient: they publish each layer's dLoss/dz rows (and the inputs the scratch does not hold already), and
     * ppoaf_icm_update_reduce forms every dW = dz^T x once per mini-batch over all rows (both observation streams for the
     * encoder) on f32 MFMA, biases as column sums of dz, [+ Adam on the spot with fused_adam].  Same call sequence and
     * outputs; sums run in MFMA K order (float32-rounding-level differences from the slab form, bitwise reproducible). */
    void* split_workspace; int64_t split_workspace_bytes;
    /* 0: the fwd_bwd launches use every XCD (default).  1 / 2: their workgroups sit on XCDs 0-3 / 4-7 only (the launch is
     * twice as wide and the workgroups dispatched to the other half return at once; workgroup b is dispatched to XCD b % 8):
     * for the PPO update running on the other half at the same time (ppoaf_ppo_update_args_t.xcd_half). */
    int32_t xcd_half;
    /* ABI 6.  1 (with split_workspace, hidden 128; ignored otherwise): ppoaf_icm_update_fwd_bwd issues ONE launch instead of
     * three -- workgroup (tile, stream) runs its stream's encoder, swaps encodings with its partner, runs one of the two
     * models, swaps the encoding gradients and runs its encoder's backward pass; the exchanges are tagged records at the
     * START of split_workspace (ppoaf_icm_update_split_workspace_bytes() includes them), tagged with *cursor + 1: zero the
     * workspace before the first launch and whenever the cursor restarts.  Bitwise the three launches' results.  Word 0 of
     * the workspace is non-zero after a launch in which a partner did not answer within 2 s (results invalid).
     * ppoaf_icm_update_fuses_kernels(args): 1 when these arguments take the single launch. */
    int32_t fuse_kernels;
} ppoaf_icm_update_args_t;

int ppoaf_icm_update_fwd_bwd(const ppoaf_icm_update_args_t* args, ppoaf_stream_t stream);
int ppoaf_icm_update_reduce(const ppoaf_icm_update_args_t* args, ppoaf_stream_t stream);
int ppoaf_icm_update_split_workspace_bytes(const ppoaf_icm_update_args_t* args, int64_t* bytes_out);
int ppoaf_icm_update_fuses_kernels(const ppoaf_icm_update_args_t* args);
/* Rollout-time intrinsic reward of a whole env batch (PPOPolicy.get_intrinsic_reward,
 * policies/ppo_policy.py:954-1007 -> ICM.forward icm.py:375-430 without the inverse model):
 * intr_out[i] = scale * sum_d (forward_model(enc(obs_i), action_i) - enc(next_obs_i))_d^2 with
 * scale = intr_reward_weight * reward_scale / 2.  Uses the same args struct: obs / next_obs /
 * actions hold the B = batch rows directly (perm NULL, fused_adam 0); params, topology and
 * act_scratch as for the update; two launches (encoder for both observations, forward model). */
int ppoaf_icm_intrinsic_reward(const ppoaf_icm_update_args_t* args, float scale, float* intr_out,
                               ppoaf_stream_t stream);

/* ------------------------------------------------------------------------ *
 * K15  fused multi-agent-transformer mini-batch update
 * replaces one iteration of PPO._ppo_batch_train for a MATPolicy  ppo.py:2292-2469
 *          MATPolicy.evaluate (token block, teacher forcing)      policies/mat_policy.py:378-439,628-658
 *          MATActorCritic.forward / MATCritic / MATActor          networks/actor_critic/multi_agent_transformer.py:22-373
 *          SelfAttention / Encoding / Decoding blocks             networks/attention.py:13-257
 *          the loss, backward and (through K11) the one clip + Adam over the shared bucket  mat_policy.py:677-699
 * for the default topology: embedding 64, one block, one head, GELU, Discrete actions (<= 8),
 * obs_dim <= 64, num_agents <= 16.  `offsets` are the 63 parameter tensors of MATActorCritic in
 * module order (each padded to 4 floats) relative to `params`.  Dataset rows are envs carrying
 * num_agents tokens: critic_obs [n_rows, A, O], raw_actions int64 [n_rows, A], advantages /
 * old_log_probs / rewards_to_go / values [n_rows, A].  adv_records / vn_records are (n, mean, M2)
 * float64 per mini-batch over its B*A values ([n_batches, 3] and [n_batches, n_ranks, 3]).
 * `fwd_bwd` = one launch on ceil(B / floor(16/A)) workgroups writing slabs [that many, bucket_total];
 * `reduce` sums them into `grads`, folds the 8 loss scalars + count into totals[9] and advances
 * the cursor; the caller then all-reduces `grads` (N > 1) and runs ppoaf_clip_adam_step.
 * ------------------------------------------------------------------------ */
typedef struct {
    int32_t obs_dim, num_agents, num_actions, embedding;
    int64_t offsets[64];
    int64_t bucket_total;
    const float* params; float* grads; float* slabs;
    const float* critic_obs; const int64_t* raw_actions;
    const float* advantages; const float* old_log_probs; const float* rewards_to_go; float* values;
    const int64_t* perm; const int32_t* row_map; int64_t n_rows;
    int64_t* cursor; int64_t B, batch_stride;
    int32_t normalize_values, n_ranks, normalize_adv, use_huber;
    float* vn_mean; float* vn_var; double* vn_count;     /* [2] double-buffered by mini-batch parity, as K12 */
    const double* vn_records; const double* adv_records;
    float surr_clip, entropy_weight, kl_loss_weight, huber_delta;
    float* loss_partials;            /* [n_workgroups, 8] */
    double* totals;                  /* [9] */
    /* single-rank fusion of K11's norm pass: with fuse_norm the reduce launch also leaves one partial of ||grads||^2
     * per workgroup -- ppoaf_mat_update_norm_partials(args) of them: ceil(bucket_total / 1024) in the slab form, one per
     * wgrad workgroup with split_workspace -- in norm_scratch[2 ..] (float64[2 + that many], at least
     * PPOAF_NORM_SCRATCH_DOUBLES) and advances step_count, so the caller follows with ppoaf_adam_step_prenormed
     * (n_norm_partials = that count) instead of ppoaf_clip_adam_step */
    double* norm_scratch; int64_t* step_count; int32_t fuse_norm;
    /* 1: critic_obs / raw_actions / advantages / old_log_probs / rewards_to_go are per-epoch tables already in shuffled
     * order (entry i belongs to perm[i]): a mini-batch reads [k * batch_stride, +B) directly, its loads depend on the cursor
     * only; `values` is still written through perm / row_map.  0: they are the rollout buffer's rows. */
    int32_t inputs_in_batch_order;
    /* split-wgrad chain (NULL: every workgroup writes a slab of the whole bucket).  With a workspace of
     * ppoaf_mat_update_split_workspace_bytes() bytes (256-byte aligned) fwd_bwd forms NO weight gradient of the 18
     * 64 x 64 linears -- a third of its MFMA work, all on the one CU that owns a token tile -- and publishes each
     * linear's input and dLoss/dz tile there instead; ppoaf_mat_update_reduce then forms dW = dz^T x once per mini-batch
     * over all tokens (K = every token row, f32 MFMA, one workgroup per 16 x 16 tile), the biases as column sums of dz,
     * and reduces the (now 13 KB) slabs of the small tensors.  Same call sequence, same outputs; sums run in MFMA K order
     * (float32-rounding-level differences from the slab form, bitwise reproducible).  With fuse_norm the launch leaves
     * ppoaf_mat_update_norm_partials(args) partials in norm_scratch[2 ..]. */
    void* split_workspace; int64_t split_workspace_bytes;
    /* mini-batch index = *cursor + mb_offset; the reduce launch adds cursor_advance to *cursor (as K12: eager use (0, 1);
     * a captured chain of n mini-batches bakes mb_offset = 0..n-1 into its nodes and advances the cursor once, by n, in
     * the last one, so all but the first read of the cursor word hit the reader's L2) */
    int64_t mb_offset, cursor_advance;
} ppoaf_mat_update_args_t;

int ppoaf_mat_update_fwd_bwd(const ppoaf_mat_update_args_t* args, ppoaf_stream_t stream);
int ppoaf_mat_update_fwd_bwd_timed(const ppoaf_mat_update_args_t* args, void* start_event, void* stop_event,
                                   ppoaf_stream_t stream);
int ppoaf_mat_update_reduce(const ppoaf_mat_update_args_t* args, ppoaf_stream_t stream);
int ppoaf_mat_update_split_workspace_bytes(const ppoaf_mat_update_args_t* args, int64_t* bytes_out);
/* Fused tail of K15's split-wgrad chain (ABI 5; single rank): ppoaf_mat_update_fwd_bwd -> ppoaf_mat_update_wgrad_adam -- TWO
 * launches per mini-batch.  The weight gradients of the 18 linears and the small tensors (ppoaf_mat_update_reduce's jobs, bit
 * for bit), the clip norm from tagged per-workgroup records every workgroup waits for, and clip + Adam on the workgroup's
 * own elements (mat_policy.py:677-699: one optimiser over actor + critic; ppoaf_adam_step_prenormed's arithmetic) in one
 * launch.  args->fuse_norm must be set (the launch advances step_count); exp_avg / exp_avg_sq / lr / betas / eps / max_norm /
 * grad_norm_out as for ppoaf_adam_step_prenormed.  ctl: ppoaf_mat_update_tail_ctl_bytes() bytes of device memory, 64-byte
 * aligned, zeroed once and kept across launches; its third 32-bit word is non-zero after a launch in which a wait ran out
 * of wait_seconds (results invalid).  All workgroups must be resident together (checked). */
int ppoaf_mat_update_tail_ctl_bytes(const ppoaf_mat_update_args_t* args, int64_t* bytes_out);
int ppoaf_mat_update_wgrad_adam(const ppoaf_mat_update_args_t* args, void* ctl, float* exp_avg, float* exp_avg_sq,
                                const float* lr, float beta1, float beta2, float eps, float grad_scale, float max_norm,
                                float* grad_norm_out, double wait_seconds, ppoaf_stream_t stream);
/* number of squared-norm partials a fuse_norm reduce launch leaves (n_norm_partials of ppoaf_adam_step_prenormed;
 * norm_scratch must hold 2 + that many doubles); -1 on invalid args */
int ppoaf_mat_update_norm_partials(const ppoaf_mat_update_args_t* args);

/* ------------------------------------------------------------------------ *
 * K16  one rollout step of a MATPolicy for all E envs of the rank in one launch
 * replaces MATPolicy.get_rollout_actions / _get_autoregressive_actions  policies/mat_policy.py:441-519,587-626
 *          MATPolicy.get_critic_values + denormalisation               mat_policy.py:660-675, utils/misc.py:124-128
 *          the per-env add_info calls                                   policies/ppo_policy.py:638-651
 * critic_obs: [E, A, obs_dim] (agents of an env side by side, in the policy's slot order); the
 * encoder runs once, the decoder A times (agent i's sampled action is agent i+1's token).
 * Outputs are the step's row of the rollout buffer: actions int64 [E, A], log-probs and
 * (denormalised) values [E, A]; *_copy_out receive the observation rows (actor_obs [E, A,
 * actor_obs_dim], or critic_obs again when NULL).  offsets / limits as K15.
 * ------------------------------------------------------------------------ */
typedef struct {
    int32_t obs_dim, num_agents, num_actions, embedding, actor_obs_dim, normalize_values;
    int64_t offsets[64];
    const float* params;
    const float* critic_obs; const float* actor_obs; int64_t E;
    uint64_t seed, offset;
    const float* vn_mean; const float* vn_var;
    int64_t* action_out; int64_t* raw_action_out; float* logp_out; float* value_out;
    float* critic_obs_copy_out; float* obs_copy_out;
    const int64_t* forced_action;   /* NULL: sample.  Else [E, A] actions to log instead of sampling (replay) */
} ppoaf_mat_step_args_t;

int ppoaf_mat_policy_step(const ppoaf_mat_step_args_t* args, ppoaf_stream_t stream);

/* ------------------------------------------------------------------------ *
 * K17  gradient exchange between the ranks of one node over peer mappings (xGMI)
 * replaces, inside the per-mini-batch update chain, the comm.Allreduce of
 *          mpi_avg_gradients                       utils/mpi_utils.py:65-86  (called from ppo.py:2443-2448,
 *          2557-2558 once per mini-batch; the division by num_procs stays in the optimiser's grad_scale)
 * One process per GPU.  Every rank creates an exchange for its flat gradient bucket, exports an opaque
 * blob (IPC handle of its exchange slots + flag words), the host all-gathers the blobs in rank order over
 * any control plane (torch.distributed, MPI, a file) and every rank connects.  After a host barrier,
 * ppoaf_peer_exchange_allreduce is ONE kernel launch with fixed arguments (hipGraph-capturable):
 *   dst[i] = sum over ranks r = 0..n-1, in that order, of src_r[i]          (bitwise identical on every rank)
 *   norm_out[0] = sum_{i <  split} (norm_scale * dst[i])^2,  norm_out[1] = the same for i >= split
 *                 (double, added in a fixed order; NULL to skip; split_floats >= bucket size: one segment and
 *                 only norm_out[0] is written)
 * dst may alias src.  All ranks must issue the same sequence of exchanges.  A rank waits at most
 * wait_seconds for its peers inside the kernel; on expiry the launch drains, dst is undefined and the error
 * word (status out[1]) holds the sequence number that timed out -- it never hangs.
 * ppoaf_peer_exchange_status: out[0] = exchanges completed, out[1] = 0 or the sequence number of a timed-out
 * wait, out[2] = 1 uncached / 2 fine-grained / 3 coarse-grained exchange memory, out[3] = n_ranks (synchronises the device).
 * ------------------------------------------------------------------------ */
#define PPOAF_PEER_EXCHANGE_MAX_RANKS 16
#define PPOAF_PEER_EXCHANGE_BLOB_BYTES 128
typedef struct ppoaf_peer_exchange ppoaf_peer_exchange_t;

/* memory_kind of the exchange slots: 0 auto (uncached, else fine-grained), 1 uncached, 2 fine-grained, 3 coarse-grained */
int ppoaf_peer_exchange_create(int rank, int n_ranks, int64_t bucket_floats, int memory_kind,
                               ppoaf_peer_exchange_t** out);
int ppoaf_peer_exchange_export(ppoaf_peer_exchange_t* x, void* blob /* host, PPOAF_PEER_EXCHANGE_BLOB_BYTES */);
int ppoaf_peer_exchange_connect(ppoaf_peer_exchange_t* x, const void* all_blobs /* host, n_ranks blobs in rank order */);
int ppoaf_peer_exchange_allreduce(ppoaf_peer_exchange_t* x, const float* src, float* dst, int64_t split_floats,
                                  float norm_scale, double* norm_out, double wait_seconds, ppoaf_stream_t stream);
int ppoaf_peer_exchange_status(ppoaf_peer_exchange_t* x, int64_t* out /* host [4] */);
/* K12 + K17 in one launch: ppoaf_ppo_update_reduce (slabs -> gradient, bookkeeping) whose column sums go from
 * registers into the exchange slot, followed by the cross-rank sum into args->grads; the two clip norms
 * (scaled by args->grad_scale) are left as per-workgroup partials inside the exchange object and are added, in
 * a fixed order, by ppoaf_ppo_update_adam_exchanged -- the Adam launch that must follow on the same stream
 * (ppoaf_ppo_update_adam with the norms taken from the exchange instead of args->norm_scratch).
 * The exchange must have been created for args->bucket_total floats (at most 256 * 1024 of them). */
int ppoaf_ppo_update_reduce_exchange(const ppoaf_ppo_update_args_t* args, ppoaf_peer_exchange_t* x,
                                     double wait_seconds, ppoaf_stream_t stream);
int ppoaf_ppo_update_adam_exchanged(const ppoaf_ppo_update_args_t* args, ppoaf_peer_exchange_t* x,
                                    ppoaf_stream_t stream);
/* The fused tail launch on N > 1 ranks of one node (ABI 5): ppoaf_ppo_update_wgrad_adam with the K17 exchange as a phase
 * of every weight-gradient job (mpi_avg_gradients, utils/mpi_utils.py:89-111, at its call sites ppo_policy.py:1035,1048):
 * workgroup b of every rank forms the same 16 x 32 piece, writes its sums into this rank's slot (16-byte system-scope
 * stores, job-major tiles), flags its peers' group b, reads their tiles back and adds them IN RANK ORDER, then takes part in
 * the launch-wide norm wait and applies clip + Adam (grad_scale = 1 / ranks) to its elements -- two launches per mini-batch,
 * bitwise the same parameters on every rank.  xchg: created for ppoaf_ppo_update_tail_exchange_floats() floats and
 * used by this entry point only (its slot layout is the job list's); not coarse-grained memory (no fences are used);
 * at most 256 workgroups (ppoaf_ppo_update_split_blocks) -- wider shapes use ppoaf_ppo_update_wgrad ->
 * ppoaf_peer_exchange_allreduce -> ppoaf_ppo_update_adam(args, 2, stream).  A peer that does not show up within
 * xchg_wait_seconds sets the exchange's error word (ppoaf_peer_exchange_status) and the launch drains. */
int ppoaf_ppo_update_tail_exchange_floats(const ppoaf_ppo_update_args_t* args, int64_t* floats_out);
int ppoaf_ppo_update_wgrad_adam_exchange(const ppoaf_ppo_update_args_t* args, void* ctl, double wait_seconds,
                                         ppoaf_peer_exchange_t* xchg, double xchg_wait_seconds, ppoaf_stream_t stream);
int ppoaf_peer_exchange_destroy(ppoaf_peer_exchange_t* x);

/* ------------------------------------------------------------------------ *
 * Rank collectives (one communicator per process, one process per GPU; RCCL bound at run time)
 * replaces broadcast_model_parameters   utils/mpi_utils.py:50-63    -> ppoaf_bcast_f32 (flat bucket, once)
 *          mpi_avg / mpi_avg_gradients  utils/mpi_utils.py:65-111   -> ppoaf_allreduce_avg_f32 (sum, then / world)
 *          the raw-data allgather of RunningMeanStd.update  utils/stats.py:47-50
 *                                                                   -> ppoaf_allgather_moments of (n, mean, M2) records
 * Rank 0 creates the id (ppoaf_comm_unique_id), the host distributes its PPOAF_COMM_UNIQUE_ID_BYTES bytes
 * (MPI_Bcast / a file / torch.distributed), every rank calls ppoaf_comm_init (collective).  All calls are
 * asynchronous on `stream` and in place; out of ppoaf_allgather_moments is [world, n_doubles].  They return
 * PPOAF_E_INVALID with a message when no librccl can be bound.  The per-mini-batch gradient exchange inside
 * the update loop has the lower-latency K17 path above; ppo_and_friends_amd's own Python host uses
 * torch.distributed (the same RCCL) for these collectives.
 * ------------------------------------------------------------------------ */
#define PPOAF_COMM_UNIQUE_ID_BYTES 128
typedef struct ppoaf_comm ppoaf_comm_t;

int ppoaf_comm_unique_id(void* out /* host, PPOAF_COMM_UNIQUE_ID_BYTES */);
int ppoaf_comm_init(int rank, int world, const void* unique_id, ppoaf_comm_t** out);
int ppoaf_allreduce_avg_f32(ppoaf_comm_t* comm, float* buf, int64_t n, ppoaf_stream_t stream);
int ppoaf_allreduce_sum_f32(ppoaf_comm_t* comm, float* buf, int64_t n, ppoaf_stream_t stream);   /* sum only */
int ppoaf_bcast_f32(ppoaf_comm_t* comm, float* buf, int64_t n, int root, ppoaf_stream_t stream);
int ppoaf_allgather_moments(ppoaf_comm_t* comm, const double* record, int64_t n_doubles, double* out,
                            ppoaf_stream_t stream);
int ppoaf_comm_destroy(ppoaf_comm_t* comm);
/* mpi_avg_gradients at its per-mini-batch call site (ppo.py:2443-2448) when the K17 peer exchange is not available:
 * fwd_bwd -> reduce -> RCCL sum all-reduce of the gradient bucket -> norms + clip + Adam for n_minibatches consecutive
 * mini-batches (cursor .. cursor + n - 1; the cursor advances by n), all launches issued from this one call (the
 * 1 / world factor is args->grad_scale, as in the three-launch chain).  mb_offset must be 0. */
int ppoaf_ppo_update_chain_allreduce(const ppoaf_ppo_update_args_t* args, ppoaf_comm_t* comm, int64_t n_minibatches,
                                     ppoaf_stream_t stream);
/* The same for the ICM update (ppo.py:2556-2562: averaged gradients, Adam without clipping; args->fused_adam must be 0;
 * norm_scratch as for ppoaf_clip_adam_step) and for the MAT update (mat_policy.py:692-699: one clip + Adam over the
 * shared bucket; args->fuse_norm must be 0, args->norm_scratch / step_count are the optimiser's). */
int ppoaf_icm_update_chain_allreduce(const ppoaf_icm_update_args_t* args, ppoaf_comm_t* comm, int64_t n_minibatches,
                                     double* norm_scratch, float* grad_norm_out /* NULL ok */, ppoaf_stream_t stream);
int ppoaf_mat_update_chain_allreduce(const ppoaf_mat_update_args_t* args, ppoaf_comm_t* comm, int64_t n_minibatches,
                                     float* exp_avg, float* exp_avg_sq, const float* lr /* device [1] */,
                                     float beta1, float beta2, float eps, float grad_scale, float max_norm,
                                     float* grad_norm_out /* NULL ok */, ppoaf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PPOAF_HIP_H */
