"""
TEST INFRASTRUCTURE (see oracle/__init__.py) -- numpy restatement of the
reference's rollout buffer arithmetic.  PINNED by tests/golden/g1..g3.

Follows /root/reference/utils/episode_info.py:
  discounted_sums            <- EpisodeInfo.compute_discounted_sums  :223-262
  gae_advantages             <- EpisodeInfo._compute_gae_advantages  :264-293
  end_episode                <- EpisodeInfo.end_episode              :419-465
                                (+ compute_advantages :401-417, :295-301)
  rollout_to_dataset         <- the order PPO.rollout closes episodes
                                (ppo.py:1804-1819, 1863-1938) feeding
                                PPODataset.build (:745-914)
  recalculate_advantages     <- PPODataset.recalculate_advantages    :721-743

Rounding points restated exactly:
  * rewards-to-go: inputs are a float32 array (rewards + clipped ending reward,
    :456-457).  Under the reference's pinned numpy<1.24 the Python-float
    `gamma * d_sum` promotes the accumulator to float64 (accum="float64", the
    default and the parity target); under NumPy>=2 it stays float32
    (accum="float32", what the reference does when run in this container).
  * GAE: delta = r(f64) + f64(fl32(gamma * V[t+1])) - f64(V[t])  (:289-290, the
    product gamma*padded_values stays float32), scan in float64 (:293).
  * ending_value is rounded to float32 (:410-411), and is NOT clipped;
    ending_reward IS clipped (:450-454) then rounded to float32 (:456-457).
"""
import numpy as np


def discounted_sums(array, gamma, accum="float64"):
    """Reverse discounted cumulative sum d_t = x_t + gamma*d_{t+1} (:254-262).

    Output is float64 (np.zeros, :254) whatever the accumulator type."""
    out = np.zeros(len(array))
    if accum == "float32":
        d = np.float32(0.0)
        g = np.float32(gamma)   # python float is "weak" under NEP 50
        for i in range(len(array) - 1, -1, -1):
            d = np.float32(np.float32(array[i]) + np.float32(g * d))
            out[i] = d
    else:
        d = 0.0
        g = float(gamma)
        for i in range(len(array) - 1, -1, -1):
            d = float(array[i]) + g * d
            out[i] = d
    return out


def gae_advantages(padded_values, rewards, gamma, lambd):
    """GAE reverse scan (:264-293).  padded_values float32 [L+1]; rewards float64-able [L]."""
    pv = np.asarray(padded_values, dtype=np.float32)
    if np.isinf(pv).any():
        raise FloatingPointError("inf encountered in padded values (episode_info.py:283-287)")
    r = np.asarray(rewards, dtype=np.float64)
    gv = (np.float32(gamma) * pv[1:]).astype(np.float32)          # stays float32
    deltas = r + gv.astype(np.float64) - pv[:-1].astype(np.float64)
    return discounted_sums(deltas, float(gamma) * float(lambd), "float64")


def end_episode(rewards, values, ending_value, ending_reward, gamma, lambd,
                bootstrap_clip=(-100.0, 100.0), use_gae=True,
                rtg_accum="float64"):
    """Returns (advantages f64[L], rewards_to_go f64[L]) as EpisodeInfo holds them (:419-465)."""
    if bootstrap_clip is not None:
        ending_reward = float(np.clip(ending_reward, bootstrap_clip[0], bootstrap_clip[1]))
    padded_rewards = np.array(list(np.asarray(rewards, dtype=np.float64)) + [ending_reward],
                              dtype=np.float32)
    rtg = discounted_sums(padded_rewards, gamma, rtg_accum)[:-1]
    v32 = np.asarray(values).astype(np.float32)
    if use_gae:
        padded_values = np.concatenate((v32, (ending_value,))).astype(np.float32)
        adv = gae_advantages(padded_values, rewards, gamma, lambd)
    else:
        adv = rtg - v32                                              # :300
    return adv, rtg


def segments_from_end_kind(end_kind):
    """
    end_kind int8 [T,E]: 0 = episode continues, 1 = terminal end, 2 = bootstrapped
    (maxed / truncated / rollout end) end AFTER step t.  The last row must be
    non-zero everywhere (ppo.py:1870-1871 closes every open episode).
    Returns the episode list in the reference's completion order: for each t,
    terminal envs ascending (ppo.py:1810-1819), then maxed envs ascending
    (ppo.py:1873-1877,1932-1938).  Each entry: (env, t_start, t_end_inclusive, kind).
    """
    T, E = end_kind.shape
    assert (end_kind[T - 1] != 0).all(), "every env must be closed at rollout end"
    start = np.zeros(E, dtype=np.int64)
    segs = []
    for t in range(T):
        for kind in (1, 2):
            for e in np.where(end_kind[t] == kind)[0]:
                segs.append((int(e), int(start[e]), t, kind))
                start[e] = t + 1
    return segs


def rollout_to_dataset(rewards, values, boot_value, boot_reward, end_kind,
                       gamma=0.99, lambd=0.95, bootstrap_clip=(-100.0, 100.0),
                       use_gae=True, rtg_accum="float64", extra=None):
    """
    Dense [T,E] rollout -> flattened dataset in the reference's episode order.

    rewards f32/f64 [T,E], values f32 [T,E]; boot_value / boot_reward f32 [T,E]
    are read only where end_kind == 2 (terminal ends use 0, 0: ppo.py:1818-1819).
    extra: dict name -> [T,E,...] arrays gathered in the same order.
    Returns dict(adv f32[N], rtg f32[N], values f32[N], ep_lens, flat_t, flat_e, **extra).
    The float32 casts are PPODataset.build's torch.tensor(..., float32) (:868-887).
    """
    segs = segments_from_end_kind(end_kind)
    adv_l, rtg_l, ft, fe, lens = [], [], [], [], []
    for (e, t0, t1, kind) in segs:
        r = rewards[t0:t1 + 1, e]
        v = values[t0:t1 + 1, e]
        ev = 0.0 if kind == 1 else float(boot_value[t1, e])
        er = 0.0 if kind == 1 else float(boot_reward[t1, e])
        a, g = end_episode(r, v, ev, er, gamma, lambd, bootstrap_clip, use_gae, rtg_accum)
        adv_l.append(a)
        rtg_l.append(g)
        ft.append(np.arange(t0, t1 + 1))
        fe.append(np.full(t1 + 1 - t0, e))
        lens.append(t1 + 1 - t0)
    flat_t = np.concatenate(ft)
    flat_e = np.concatenate(fe)
    out = dict(adv=np.concatenate(adv_l).astype(np.float32),
               rtg=np.concatenate(rtg_l).astype(np.float32),
               values=np.asarray(values, dtype=np.float32)[flat_t, flat_e],
               ep_lens=np.array(lens), flat_t=flat_t, flat_e=flat_e, segs=segs)
    if extra:
        for k, arr in extra.items():
            out[k] = np.asarray(arr)[flat_t, flat_e]
    return out


def recalculate_advantages(flat_rewards, new_flat_values, ep_lens, ending_values,
                           gamma=0.99, lambd=0.95, use_gae=True, flat_rtg=None):
    """
    PPODataset.recalculate_advantages (:721-743): per episode, overwrite values
    with the dataset's (float32) values and re-run compute_advantages with the
    episode's stored ending_value.  rtg is NOT recomputed.
    """
    out = []
    pos = 0
    for L, ev in zip(ep_lens, ending_values):
        v = np.asarray(new_flat_values[pos:pos + L], dtype=np.float32)
        if use_gae:
            pv = np.concatenate((v, (float(ev),))).astype(np.float32)
            out.append(gae_advantages(pv, flat_rewards[pos:pos + L], gamma, lambd))
        else:
            out.append(np.asarray(flat_rtg[pos:pos + L], dtype=np.float64) - v)
        pos += L
    return np.concatenate(out).astype(np.float32)
