"""
TEST INFRASTRUCTURE (see oracle/__init__.py) -- literal restatement of the per-step statistics
bookkeeping of the reference's rollout (ppo.py:1600-1631 initial values, :1756-1790 ranges and running
scores, :1805-1851 terminations, :1863-1976 cuts and the end of the rollout, :1978-2099 the final block),
for one policy on one rank (one agent per env, or A agents sharing the policy: trailing agent axis).
PINNED by the rollout status blocks of fixtures g12_* (tests/test_oracle_update_golden.py), recorded from the
unmodified reference's PPO.rollout.
"""
import numpy as np


def rollout_statistics_loop(reward, nat, intr, next_obs_min, next_obs_max, term, boot, next_reward, max_ts=None):
    """
    reward, nat, intr [T, E]; next_obs_min/max [T] (extrema of the post-step observation);
    term, boot [T, E] bool; next_reward [T, E]: the critic's bootstrap rewards of the whole batch at step t.
    """
    if reward.ndim == 3:                       # A agents of one policy: per-env scores sum the agents (ppo.py:1782-1787),
        A = reward.shape[2]                    # extrema run over all of them, bootstraps count once per agent (:1901-1912)
        r_all, n_all, i_all = reward, nat, intr
        reward, nat, intr = reward.sum(2), nat.sum(2), intr.sum(2)
    else:
        A = 1
        r_all, n_all, i_all = reward, nat, intr
    T, E = reward.shape
    fmax = np.finfo(np.float32).max
    top_rollout_score, top_reward = -fmax, -fmax
    mx_r, mn_r, mx_n, mn_n, mx_i, mn_i, mx_o, mn_o = -fmax, fmax, -fmax, fmax, -fmax, fmax, -fmax, fmax
    ep_nat, ep_sc, ep_in = np.zeros(E), np.zeros(E), np.zeros(E)
    tot_nat, tot_sc, tot_in = np.zeros(E), np.zeros(E), np.zeros(E)
    bs_min, bs_max, bs_sum, total_bs = fmax, -fmax, np.zeros(E), 0
    episode_lengths = np.zeros(E, dtype=np.int64)
    total_episodes = 0.0
    longest_run, shortest_run, avg_run = 0, T, T
    for t in range(T):
        episode_lengths += 1
        mx_r, mn_r = max(mx_r, r_all[t].max()), min(mn_r, r_all[t].min())
        mx_n, mn_n = max(mx_n, n_all[t].max()), min(mn_n, n_all[t].min())
        mx_i, mn_i = max(mx_i, i_all[t].max()), min(mn_i, i_all[t].min())
        mx_o, mn_o = max(mx_o, next_obs_max[t]), min(mn_o, next_obs_min[t])
        ep_sc += reward[t]; ep_nat += nat[t]; ep_in += intr[t]
        top_reward = max(top_reward, n_all[t].max())
        where_term = np.where(term[t])[0]
        where_not_term = np.where(~term[t])[0]
        if where_term.size > 0:
            top_rollout_score = max(top_rollout_score, ep_nat[where_term].max())
            tot_nat[where_term] += ep_nat[where_term]
            tot_in[where_term] += ep_in[where_term]
            tot_sc[where_term] += ep_sc[where_term]
            ep_sc[where_term] = 0; ep_nat[where_term] = 0; ep_in[where_term] = 0
            longest_run = max(longest_run, episode_lengths[where_term].max())
            shortest_run = min(shortest_run, episode_lengths[where_term].min())
            avg_run = episode_lengths[where_term].mean()
            episode_lengths[where_term] = 0
            total_episodes += where_term.size
        if boot[t].any():
            bs_min, bs_max = min(bs_min, float(next_reward[t].min())), max(bs_max, float(next_reward[t].max()))
            bs_sum += next_reward[t] if next_reward[t].ndim == 1 else next_reward[t].sum(1)
            total_bs += A
        if t == T - 1:
            combined = episode_lengths.sum()
            ts_before = max(T * E - combined, 0)
            cur = total_episodes if total_episodes != 0 else 1.0
            avg_ep_len = combined / E if ts_before == 0 else ts_before / cur
            total_episodes += (episode_lengths / avg_ep_len).sum()
            tot_nat += ep_nat; tot_sc += ep_sc; tot_in += ep_in
            if where_not_term.size > 0:
                top_rollout_score = max(top_rollout_score, ep_nat[where_not_term].max())
        longest_run = max(longest_run, episode_lengths.max())
    if total_episodes < 1.0:
        top_rollout_score = max(top_rollout_score, ep_nat.max())
    return {"total episodes": total_episodes, "score avg": tot_sc.sum() / total_episodes,
            "natural score avg": tot_nat.sum() / total_episodes, "top score": top_rollout_score,
            "reward range": (mn_r, mx_r), "natural reward range": (mn_n, mx_n), "obs range": (mn_o, mx_o),
            "bootstrap range": (bs_min, bs_max), "bootstrap avg": 0.0 if total_bs == 0 else bs_sum.sum() / total_bs,
            "longest episode": longest_run, "shortest episode": shortest_run, "average episode": avg_run,
            "intrinsic score avg": tot_in.sum() / (total_episodes / E), "intr reward range": (mn_i, mx_i),
            "top natural reward": top_reward}
