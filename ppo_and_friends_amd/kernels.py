"""
Tensor-level wrappers over the C ABI (include/ppoaf_hip.h).

Every function validates shapes / dtypes / devices on the host BEFORE the
launch (a kernel is never started on operands it was not written for), then
enqueues on torch's current HIP stream.  Nothing here computes on the CPU.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import PpoafError, check, ptr, stream


def _req(cond, msg):
    if not cond:
        raise PpoafError(msg)


def _f32(t, name):
    _req(t.dtype == torch.float32, f"{name}: expected float32, got {t.dtype}")
    return t


def _clip_args(bootstrap_clip):
    if bootstrap_clip is None:
        return 0, 0.0, 0.0
    return 1, float(bootstrap_clip[0]), float(bootstrap_clip[1])


# --------------------------------------------------------------------------
# K1
# --------------------------------------------------------------------------
def gae_rtg_tmajor(rewards, values, boot_value, boot_reward, end_kind=None,
                   gamma=0.99, lambd=0.95, bootstrap_clip=(-100.0, 100.0),
                   use_gae=True, adv_out=None, rtg_out=None, timing_events=None):
    """[T,E] time-major GAE + rewards-to-go (utils/episode_info.py:419-465).
    timing_events = (start, stop) from event_create(): the kernel's own begin / end are stamped."""
    _req(rewards.dim() == 2, "rewards must be [T,E]")
    T, E = rewards.shape
    _f32(rewards, "rewards"); _f32(values, "values")
    _f32(boot_value, "boot_value"); _f32(boot_reward, "boot_reward")
    _req(values.shape == (T, E), "values must be [T,E]")
    if end_kind is None:
        _req(boot_value.numel() == E and boot_reward.numel() == E,
             "fixed-length form: boot_value / boot_reward must be [E]")
    else:
        _req(end_kind.dtype == torch.int8 and end_kind.shape == (T, E), "end_kind must be int8 [T,E]")
        _req(boot_value.shape == (T, E) and boot_reward.shape == (T, E),
             "dense form: boot_value / boot_reward must be [T,E]")
    if adv_out is None:
        adv_out = torch.empty_like(rewards)
    if rtg_out is None:
        rtg_out = torch.empty_like(rewards)
    _req(adv_out.shape == (T, E) and rtg_out.shape == (T, E), "outputs must be [T,E]")
    _f32(adv_out, "adv_out"); _f32(rtg_out, "rtg_out")
    hc, lo, hi = _clip_args(bootstrap_clip)
    e0, e1 = (None, None) if timing_events is None else timing_events
    check(_lib.load().ppoaf_gae_rtg_tmajor_timed(
        ptr(rewards), ptr(values), ptr(boot_value), ptr(boot_reward), ptr(end_kind),
        T, E, float(gamma), float(lambd), hc, lo, hi, int(bool(use_gae)),
        ptr(adv_out), ptr(rtg_out), e0, e1, stream()), "gae_rtg_tmajor")
    return adv_out, rtg_out


def event_create():
    e = _lib.load().ppoaf_event_create()
    _req(e is not None, "hipEventCreate failed")
    return C.c_void_p(e)


def event_elapsed_ms(start, stop):
    ms = C.c_float(0.0)
    check(_lib.load().ppoaf_event_elapsed_ms(start, stop, C.byref(ms)), "event_elapsed_ms")
    return ms.value


_stream_pairs = {}


def concurrent_stream_pair(device=None):
    """
    Two streams of `device` that really run side by side.  The HIP runtime multiplexes its streams onto a few
    hardware queues (least-used queue first, ties broken arbitrarily), so two streams taken one after the other
    from torch's pool can share a queue and serialise -- measured on MI355X: the 3rd and 4th streams a process
    uses do, which cost the overlapped PPO/ICM epochs 30%.  The pair is therefore chosen by measurement, once
    per device: a short spin kernel on each candidate pair, first pair whose wall time is that of one kernel.
    """
    import time
    import itertools
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev in _stream_pairs:
        return _stream_pairs[dev]
    cycles = 3_000_000                                         # ~1.3 ms
    with torch.cuda.device(dev):
        cands = [torch.cuda.Stream() for _ in range(5)]

        def wall(streams):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for s in streams:
                with torch.cuda.stream(s):
                    torch.cuda._sleep(cycles)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        for s in cands:                                        # bind every candidate to its queue first
            wall([s])
        one = min(wall([cands[0]]) for _ in range(3))
        best, best_t = (cands[0], cands[1]), float("inf")
        for a, b in itertools.combinations(cands, 2):
            t = min(wall([a, b]) for _ in range(2))
            if t < best_t:
                best, best_t = (a, b), t
            if t < 1.35 * one:
                break
    _stream_pairs[dev] = best
    return best


def gae_rtg_traj(rewards, values, ending_value, ending_reward, traj_start, traj_len,
                 gamma=0.99, lambd=0.95, bootstrap_clip=(-100.0, 100.0), use_gae=True,
                 adv_out=None, rtg_out=None, compute_rtg=True):
    """Flat episode-major [N] trajectories (one wave each)."""
    N = rewards.numel()
    n_traj = traj_start.numel()
    _f32(rewards, "rewards"); _f32(values, "values")
    _f32(ending_value, "ending_value"); _f32(ending_reward, "ending_reward")
    _req(values.numel() == N, "values must match rewards")
    _req(traj_start.dtype == torch.int64 and traj_len.dtype == torch.int32,
         "traj_start int64 / traj_len int32")
    _req(traj_len.numel() == n_traj and ending_value.numel() == n_traj
         and ending_reward.numel() == n_traj, "per-trajectory arrays must have n_traj entries")
    if n_traj:
        # bounds are validated on the host: a bad trajectory table must never reach the kernel
        ends = traj_start + traj_len.to(torch.int64)
        _req(int(traj_start.min()) >= 0 and int(ends.max()) <= N and int(traj_len.min()) >= 0,
             "trajectory table exceeds the flat buffer")
    if adv_out is None:
        adv_out = torch.zeros_like(rewards)
    if rtg_out is None and (compute_rtg or not use_gae):
        rtg_out = torch.zeros_like(rewards)
    if N == 0:                       # only zero-length trajectories: nothing to scan (and no storage to point at)
        return adv_out, rtg_out
    hc, lo, hi = _clip_args(bootstrap_clip)
    check(_lib.load().ppoaf_gae_rtg_traj(
        ptr(rewards), ptr(values), ptr(ending_value), ptr(ending_reward), ptr(traj_start),
        ptr(traj_len), n_traj, float(gamma), float(lambd), hc, lo, hi, int(bool(use_gae)),
        ptr(adv_out), ptr(rtg_out), stream()), "gae_rtg_traj")
    return adv_out, rtg_out


# --------------------------------------------------------------------------
# K2 + K3
# --------------------------------------------------------------------------
SC_SURR, SC_ACTOR, SC_CRITIC, SC_ENTROPY, SC_KL, SC_ADV_MEAN, SC_ADV_STD, SC_BAD = range(8)


def ppo_loss_fwd_bwd(cur_logp, old_logp, adv, entropy, values, rtg, normalize_adv=True,
                     surr_clip=0.2, entropy_weight=0.01, kl_loss_weight=0.0, use_huber=False,
                     huber_delta=10.0, scalars=None, d_logp=None, d_entropy=None, d_values=None,
                     need_grads=True):
    B = cur_logp.numel()
    for n, t in (("cur_logp", cur_logp), ("old_logp", old_logp), ("adv", adv),
                 ("values", values), ("rtg", rtg)):
        _f32(t, n)
        _req(t.numel() == B, f"{n}: expected {B} elements, got {t.numel()}")
    if entropy is not None:
        _f32(entropy, "entropy")
        _req(entropy.numel() == B, "entropy length")
    dev = cur_logp.device
    if scalars is None:
        scalars = torch.empty(8, dtype=torch.float32, device=dev)
    _req(scalars.numel() == 8 and scalars.dtype == torch.float32, "scalars must be float32[8]")
    if need_grads:
        d_logp = torch.empty(B, dtype=torch.float32, device=dev) if d_logp is None else d_logp
        d_entropy = torch.empty(B, dtype=torch.float32, device=dev) if d_entropy is None else d_entropy
        d_values = torch.empty(B, dtype=torch.float32, device=dev) if d_values is None else d_values
        for n, t in (("d_logp", d_logp), ("d_entropy", d_entropy), ("d_values", d_values)):
            _req(t.numel() == B and t.dtype == torch.float32, f"{n} must be float32[{B}]")
    check(_lib.load().ppoaf_ppo_loss_fwd_bwd(
        ptr(cur_logp), ptr(old_logp), ptr(adv), ptr(entropy), ptr(values), ptr(rtg), B,
        int(bool(normalize_adv)), float(surr_clip), float(entropy_weight), float(kl_loss_weight),
        int(bool(use_huber)), float(huber_delta), ptr(scalars), ptr(d_logp), ptr(d_entropy),
        ptr(d_values), stream()), "ppo_loss_fwd_bwd")
    return scalars, d_logp, d_entropy, d_values


# --------------------------------------------------------------------------
# K4
# --------------------------------------------------------------------------
def minibatch_gather(pairs, perm, row_map=None):
    """pairs: list of (src [n_rows, ...], dst [B, ...]); rows picked by perm (int64 [B])."""
    _req(1 <= len(pairs) <= _lib.MAX_GATHER_FIELDS, "1..8 fields per gather launch")
    _req(perm.dtype == torch.int64 and perm.dim() == 1, "perm must be int64 [B]")
    B = perm.numel()
    if row_map is not None:
        _req(row_map.dtype == torch.int32, "row_map must be int32")
        n_rows = row_map.numel()
    else:
        n_rows = pairs[0][0].shape[0]
    arr = (_lib.GatherField * len(pairs))()
    for i, (src, dst) in enumerate(pairs):
        _req(src.dtype == dst.dtype, f"field {i}: dtype mismatch")
        rb = src[0].numel() * src.element_size() if src.shape[0] else 0
        _req(rb > 0 and rb % 4 == 0, f"field {i}: row bytes {rb} must be a positive multiple of 4")
        _req(dst.shape[0] == B and dst[0].numel() * dst.element_size() == rb,
             f"field {i}: dst must be [B, same row]")
        if row_map is None:
            _req(src.shape[0] == n_rows, f"field {i}: row count mismatch")
        arr[i].src = src.data_ptr()
        arr[i].dst = dst.data_ptr()
        arr[i].row_bytes = rb
        ptr(src); ptr(dst)      # device / contiguity checks
    check(_lib.load().ppoaf_minibatch_gather(arr, len(pairs), ptr(perm), ptr(row_map),
                                             n_rows, B, stream()), "minibatch_gather")


def scatter_rows_f32(src, perm, dst, row_map=None):
    B = perm.numel()
    _req(src.numel() == B and src.dtype == torch.float32 and dst.dtype == torch.float32,
         "scatter_rows_f32: src float32[B], dst float32[n_rows]")
    n_rows = row_map.numel() if row_map is not None else dst.numel()
    check(_lib.load().ppoaf_scatter_rows_f32(ptr(src), ptr(perm), ptr(row_map), n_rows, B,
                                             ptr(dst), stream()), "scatter_rows_f32")


# --------------------------------------------------------------------------
# K5
# --------------------------------------------------------------------------
def batch_moments(data, W=1, out=None):
    _f32(data, "data")
    n = data.numel() // W
    _req(n * W == data.numel() and n >= 1, "data must be [n, W] with n >= 1")
    if out is None:
        out = torch.empty(1 + 2 * W, dtype=torch.float64, device=data.device)
    _req(out.dtype == torch.float64 and out.numel() == 1 + 2 * W, "moments must be float64[1+2W]")
    check(_lib.load().ppoaf_batch_moments(ptr(data), n, W, ptr(out), stream()), "batch_moments")
    return out


def running_moments_integrate(moments, mean, var, count):
    W = mean.numel()
    _req(moments.dtype == torch.float64 and moments.numel() % (1 + 2 * W) == 0,
         "moments must be float64[R, 1+2W]")
    R = moments.numel() // (1 + 2 * W)
    _req(mean.dtype == torch.float32 and var.dtype == torch.float32 and var.numel() == W
         and count.dtype == torch.float64 and count.numel() == 1, "state dtypes/shapes")
    check(_lib.load().ppoaf_running_moments_integrate(ptr(moments), R, W, ptr(mean), ptr(var),
                                                      ptr(count), stream()),
          "running_moments_integrate")


def normalize(x, mean, var, eps=1e-8, clip=None, out=None):
    _f32(x, "x")
    W = mean.numel()
    _req(x.numel() % W == 0, "x must be [n, W]")
    out = torch.empty_like(x) if out is None else out
    hc, lo, hi = (0, 0.0, 0.0) if clip is None else (1, float(clip[0]), float(clip[1]))
    check(_lib.load().ppoaf_normalize(ptr(x), x.numel() // W, W, ptr(mean), ptr(var), float(eps),
                                      lo, hi, hc, ptr(out), stream()), "normalize")
    return out


def denormalize(x, mean, var, eps=1e-8, out=None):
    _f32(x, "x")
    W = mean.numel()
    _req(x.numel() % W == 0, "x must be [n, W]")
    out = torch.empty_like(x) if out is None else out
    check(_lib.load().ppoaf_denormalize(ptr(x), x.numel() // W, W, ptr(mean), ptr(var), float(eps),
                                        ptr(out), stream()), "denormalize")
    return out


# --------------------------------------------------------------------------
# K6
# --------------------------------------------------------------------------
def categorical_sample(logits, seed, offset, action_out=None, logp_out=None, probs_out=None):
    _f32(logits, "logits")
    _req(logits.dim() == 2, "logits must be [n,K]")
    n, K = logits.shape
    dev = logits.device
    action_out = torch.empty(n, dtype=torch.int64, device=dev) if action_out is None else action_out
    logp_out = torch.empty(n, dtype=torch.float32, device=dev) if logp_out is None else logp_out
    _req(action_out.dtype == torch.int64 and action_out.numel() == n, "action_out int64[n]")
    _req(logp_out.dtype == torch.float32 and logp_out.numel() == n, "logp_out float32[n]")
    check(_lib.load().ppoaf_categorical_sample(ptr(logits), n, K, int(seed), int(offset),
                                               ptr(action_out), ptr(logp_out), ptr(probs_out),
                                               stream()), "categorical_sample")
    return action_out, logp_out


def categorical_eval_fwd(logits, actions, want_probs=True):
    _f32(logits, "logits")
    n, K = logits.shape
    _req(actions.dtype == torch.int64 and actions.numel() == n, "actions must be int64[n]")
    dev = logits.device
    logp = torch.empty(n, dtype=torch.float32, device=dev)
    ent = torch.empty(n, dtype=torch.float32, device=dev)
    probs = torch.empty(n, K, dtype=torch.float32, device=dev) if want_probs else None
    check(_lib.load().ppoaf_categorical_eval_fwd(ptr(logits), ptr(actions), n, K, ptr(logp),
                                                 ptr(ent), ptr(probs), stream()),
          "categorical_eval_fwd")
    return logp, ent, probs


def categorical_eval_bwd(probs, actions, d_logp, d_entropy):
    n, K = probs.shape
    d_logits = torch.empty_like(probs)
    check(_lib.load().ppoaf_categorical_eval_bwd(ptr(probs), ptr(actions), ptr(d_logp),
                                                 ptr(d_entropy), n, K, ptr(d_logits), stream()),
          "categorical_eval_bwd")
    return d_logits


def gaussian_tanh_eval_fwd(mean, log_std, x, min_std=0.01):
    _f32(mean, "mean"); _f32(log_std, "log_std"); _f32(x, "x")
    n, D = mean.shape
    _req(x.shape == (n, D) and log_std.numel() == D, "x [n,D], log_std [D]")
    logp = torch.empty(n, dtype=torch.float32, device=mean.device)
    ent = torch.empty(n, dtype=torch.float32, device=mean.device)
    check(_lib.load().ppoaf_gaussian_tanh_eval_fwd(ptr(mean), ptr(log_std), ptr(x), n, D,
                                                   float(min_std), ptr(logp), ptr(ent), stream()),
          "gaussian_tanh_eval_fwd")
    return logp, ent


def gaussian_tanh_eval_bwd(mean, log_std, x, d_logp, d_entropy, min_std=0.01):
    n, D = mean.shape
    d_mean = torch.empty_like(mean)
    d_log_std = torch.empty_like(log_std)
    check(_lib.load().ppoaf_gaussian_tanh_eval_bwd(ptr(mean), ptr(log_std), ptr(x), ptr(d_logp),
                                                   ptr(d_entropy), n, D, float(min_std),
                                                   ptr(d_mean), ptr(d_log_std), stream()),
          "gaussian_tanh_eval_bwd")
    return d_mean, d_log_std


def gaussian_tanh_sample(mean, log_std, seed, offset, min_std=0.01, act_lo=None, act_hi=None):
    """act_lo / act_hi: float32[D] device tensors (bounds per action dimension) or both None for the unit box."""
    _f32(mean, "mean"); _f32(log_std, "log_std")
    n, D = mean.shape
    _req((act_lo is None) == (act_hi is None), "gaussian_tanh_sample: give both bounds or neither")
    if act_lo is not None:
        _f32(act_lo, "act_lo"); _f32(act_hi, "act_hi")
        _req(act_lo.numel() == D and act_hi.numel() == D, "gaussian_tanh_sample: bounds must hold one value per action dimension")
    raw = torch.empty_like(mean)
    act = torch.empty_like(mean)
    logp = torch.empty(n, dtype=torch.float32, device=mean.device)
    check(_lib.load().ppoaf_gaussian_tanh_sample(ptr(mean), ptr(log_std), n, D, float(min_std),
                                                 ptr(act_lo), ptr(act_hi), int(seed),
                                                 int(offset), ptr(raw), ptr(act), ptr(logp),
                                                 stream()), "gaussian_tanh_sample")
    return raw, act, logp


# --------------------------------------------------------------------------
# K11
# --------------------------------------------------------------------------
NORM_SCRATCH_DOUBLES = 66          # PPOAF_NORM_SCRATCH_DOUBLES


def clip_adam_step(params, grads, exp_avg, exp_avg_sq, step_count, lr, norm_scratch,
                   beta1=0.9, beta2=0.999, eps=1e-5, grad_scale=1.0, max_norm=0.5,
                   grad_norm_out=None):
    n = params.numel()
    for nme, t in (("params", params), ("grads", grads), ("exp_avg", exp_avg),
                   ("exp_avg_sq", exp_avg_sq)):
        _f32(t, nme)
        _req(t.numel() == n and t.dim() == 1, f"{nme} must be flat float32[{n}]")
    _req(step_count.dtype == torch.int64 and step_count.numel() == 1, "step_count int64[1]")
    _req(lr.dtype == torch.float32 and lr.numel() == 1, "lr float32[1] on the device")
    _req(norm_scratch.dtype == torch.float64 and norm_scratch.numel() >= NORM_SCRATCH_DOUBLES,
         f"norm_scratch float64[>={NORM_SCRATCH_DOUBLES}] (norm + per-workgroup partials)")
    check(_lib.load().ppoaf_clip_adam_step(
        ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), n, ptr(step_count), ptr(lr),
        float(beta1), float(beta2), float(eps), float(grad_scale),
        float(max_norm if max_norm is not None else 0.0), ptr(norm_scratch), ptr(grad_norm_out),
        stream()), "clip_adam_step")


# --------------------------------------------------------------------------
# K12: fused mini-batch update
# --------------------------------------------------------------------------
ACT_RELU, ACT_LEAKY_RELU, ACT_TANH = 0, 1, 2
HEAD_CATEGORICAL, HEAD_GAUSSIAN = 0, 1
UPDATE_ROWS_PER_WG = 16


def minibatch_moments(data_flat, perm, row_map, B, out=None):
    """[ceil(n/B), 3] float64 (n, mean, M2) records of data_flat[row_map[perm]] per mini-batch."""
    _f32(data_flat, "data")
    _req(perm.dtype == torch.int64 and perm.dim() == 1, "perm must be int64 [n]")
    n = perm.numel()
    nb = (n + B - 1) // B
    if out is None:
        out = torch.empty(nb, 3, dtype=torch.float64, device=data_flat.device)
    _req(out.dtype == torch.float64 and out.numel() == nb * 3, "records must be float64 [nb,3]")
    if row_map is not None:
        _req(row_map.dtype == torch.int32 and row_map.numel() <= data_flat.numel() and n <= row_map.numel(),
             "row_map must be int32 and cover perm")
    check(_lib.load().ppoaf_minibatch_moments(ptr(data_flat), ptr(perm), ptr(row_map), n, int(B),
                                              ptr(out), stream()), "minibatch_moments")
    return out


def ppo_update_fwd_bwd(args):
    check(_lib.load().ppoaf_ppo_update_fwd_bwd(C.byref(args), stream()), "ppo_update_fwd_bwd")


def ppo_update_reduce(args, compute_norms):
    check(_lib.load().ppoaf_ppo_update_reduce(C.byref(args), int(compute_norms), stream()),
          "ppo_update_reduce")


def ppo_update_adam(args, compute_norms):
    check(_lib.load().ppoaf_ppo_update_adam(C.byref(args), int(compute_norms), stream()),
          "ppo_update_adam")


def policy_step(args):
    check(_lib.load().ppoaf_policy_step(C.byref(args), stream()), "policy_step")


# --------------------------------------------------------------------------
# K8: ICM forward-model loss / intrinsic reward
# --------------------------------------------------------------------------
def icm_forward_loss_fwd(pred, enc2, reward_scale, want_loss=True):
    _f32(pred, "pred"); _f32(enc2, "enc2")
    _req(pred.dim() == 2 and pred.shape == enc2.shape, "pred / enc2 must both be [n, D]")
    n, D = pred.shape
    dev = pred.device
    rowsum = torch.empty(n, dtype=torch.float32, device=dev)
    intr = torch.empty(n, dtype=torch.float32, device=dev)
    f_loss = torch.empty(1, dtype=torch.float32, device=dev) if want_loss else None
    check(_lib.load().ppoaf_icm_forward_loss_fwd(ptr(pred), ptr(enc2), n, D, float(reward_scale),
                                                 ptr(rowsum), ptr(intr), ptr(f_loss), stream()),
          "icm_forward_loss_fwd")
    return intr, f_loss


def icm_forward_loss_bwd(pred, enc2, grad_f_loss, want_enc2_grad=True):
    n, D = pred.shape
    _req(grad_f_loss.dtype == torch.float32 and grad_f_loss.numel() == 1, "grad_f_loss must be a float32 scalar")
    d_pred = torch.empty_like(pred)
    d_enc2 = torch.empty_like(enc2) if want_enc2_grad else None
    check(_lib.load().ppoaf_icm_forward_loss_bwd(ptr(pred), ptr(enc2), n, D, ptr(grad_f_loss.reshape(1)),
                                                 ptr(d_pred), ptr(d_enc2), stream()), "icm_forward_loss_bwd")
    return d_pred, d_enc2


# --------------------------------------------------------------------------
# K9: MAT attention core
# --------------------------------------------------------------------------
def mat_attention_fwd(q, k, v, masked):
    for n, t in (("q", q), ("k", k), ("v", v)):
        _f32(t, n)
        _req(t.dim() == 3 and t.shape == q.shape, f"{n} must be [n_seq, L, D]")
    n_seq, L, D = q.shape
    y = torch.empty_like(q)
    probs = torch.empty(n_seq, L, L, dtype=torch.float32, device=q.device)
    check(_lib.load().ppoaf_mat_attention_fwd(ptr(q), ptr(k), ptr(v), n_seq, L, D, int(bool(masked)),
                                              ptr(y), ptr(probs), stream()), "mat_attention_fwd")
    return y, probs


def mat_attention_bwd(q, k, v, probs, dy):
    n_seq, L, D = q.shape
    _req(dy.shape == q.shape and probs.shape == (n_seq, L, L), "dy [n_seq,L,D], probs [n_seq,L,L]")
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    check(_lib.load().ppoaf_mat_attention_bwd(ptr(q), ptr(k), ptr(v), ptr(probs), ptr(dy), n_seq, L, D,
                                              ptr(dq), ptr(dk), ptr(dv), stream()), "mat_attention_bwd")
    return dq, dk, dv


# --------------------------------------------------------------------------
# K13: environment filters (observation / reward normalisation + clipping)
# --------------------------------------------------------------------------
def env_filter_record_len(G, W_o, W_c, has_reward):
    """PPOAF_ENV_FILTER_RECORD_LEN of include/ppoaf_hip.h."""
    return 1 + 2 * G * (W_o + W_c + (1 if has_reward else 0))


def obs_filter(x, out, G, n, stats=None, update=True, clip=None, eps=1e-8):
    """
    ppoaf_obs_filter_t for x/out [G*n, W] float32; stats = (mean [G*W] f32, var [G*W] f32,
    count [G*W] f64) or None (clip only).  The tensors must outlive the launches.
    """
    _req(x.is_cuda and x.is_contiguous() and out.is_contiguous() and x.dim() == 2, "obs_filter: x must be [G*n, W] on the device")
    _f32(x, "obs_filter x"); _f32(out, "obs_filter out")
    W = x.shape[1]
    _req(x.shape[0] == G * n and out.shape == x.shape and W >= 1, f"obs_filter: shape {tuple(x.shape)} vs G={G} n={n}")
    f = _lib.ObsFilter()
    f.x, f.out, f.W = ptr(x), ptr(out), W
    f.normalize = 0
    if stats is not None:
        mean, var, count = stats
        _req(mean.dtype == torch.float32 and var.dtype == torch.float32 and count.dtype == torch.float64,
             "obs_filter: stats dtypes (f32, f32, f64)")
        _req(mean.numel() == G * W and var.numel() == G * W and count.numel() == G * W, "obs_filter: stats size")
        _req(mean.is_cuda and var.is_cuda and count.is_cuda, "obs_filter: stats must be on the device")
        f.mean, f.var, f.count = ptr(mean), ptr(var), ptr(count)
        f.normalize = 1
    f.update = 1 if (update and stats is not None) else 0
    f.has_clip = 0
    if clip is not None:
        f.has_clip, f.clip_lo, f.clip_hi = 1, float(clip[0]), float(clip[1])
    f.eps = float(eps)
    return f


def reward_filter(reward, terminated, truncated, out, G, n, state=None, update=True, clip=None,
                  gamma=0.99, eps=1e-8):
    """
    ppoaf_reward_filter_t for reward/out [G*n] float32, terminated/truncated [G*n] bool;
    state = (running_reward [G*n] f64, mean [G] f64, var [G] f64, count [G] f64) or None.
    """
    _req(reward.is_cuda and reward.is_contiguous() and out.is_contiguous(), "reward_filter: device contiguous")
    _f32(reward, "reward_filter reward"); _f32(out, "reward_filter out")
    _req(reward.numel() == G * n and out.numel() == G * n, "reward_filter: size")
    f = _lib.RewardFilter()
    f.reward, f.out = ptr(reward), ptr(out)
    f.normalize = 0
    if state is not None:
        rr, mean, var, count = state
        for t in (rr, mean, var, count):
            _req(t.dtype == torch.float64 and t.is_cuda, "reward_filter: state must be float64 on the device")
        _req(rr.numel() == G * n and mean.numel() == G and var.numel() == G and count.numel() == G,
             "reward_filter: state size")
        _req(terminated is not None and terminated.numel() == G * n and terminated.element_size() == 1
             and terminated.is_contiguous(), "reward_filter: terminated must be [G*n] bool/uint8")
        f.done = ptr(terminated)
        if truncated is not None:
            _req(truncated.numel() == G * n and truncated.element_size() == 1 and truncated.is_contiguous(),
                 "reward_filter: truncated must be [G*n] bool/uint8")
            f.done2 = ptr(truncated)
        f.running_reward, f.mean, f.var, f.count = ptr(rr), ptr(mean), ptr(var), ptr(count)
        f.normalize = 1
    f.update = 1 if (update and state is not None) else 0
    f.has_clip = 0
    if clip is not None:
        f.has_clip, f.clip_lo, f.clip_hi = 1, float(clip[0]), float(clip[1])
    f.gamma, f.eps = float(gamma), float(eps)
    return f


def _ref(s):
    return C.byref(s) if s is not None else None


def env_filter_moments(obs_f, cobs_f, rew_f, G, n, record):
    _req(record.dtype == torch.float64 and record.is_cuda and record.is_contiguous(), "env_filter_moments: record")
    need = env_filter_record_len(G, obs_f.W if obs_f is not None else 0, cobs_f.W if cobs_f is not None else 0,
                                 rew_f is not None)
    _req(record.numel() >= need, f"env_filter_moments: record holds {record.numel()} < {need}")
    check(_lib.load().ppoaf_env_filter_moments(_ref(obs_f), _ref(cobs_f), _ref(rew_f), int(G), int(n),
                                               ptr(record), stream()), "env_filter_moments")
    return record


def env_filter_apply(obs_f, cobs_f, rew_f, G, n, records=None):
    R = 0
    if records is not None:
        need = env_filter_record_len(G, obs_f.W if obs_f is not None else 0,
                                     cobs_f.W if cobs_f is not None else 0, rew_f is not None)
        _req(records.dtype == torch.float64 and records.is_cuda and records.is_contiguous(), "env_filter_apply: records")
        _req(records.numel() % need == 0 and records.numel() >= need, "env_filter_apply: records size")
        R = records.numel() // need
    check(_lib.load().ppoaf_env_filter_apply(_ref(obs_f), _ref(cobs_f), _ref(rew_f), int(G), int(n),
                                             ptr(records) if records is not None else None, R, stream()),
          "env_filter_apply")
