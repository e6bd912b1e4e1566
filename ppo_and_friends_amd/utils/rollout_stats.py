"""
The statistics block that closes PPO.rollout in the reference (ppo.py:1796-2100), from the arrays the
rollout left on the device instead of per-step Python bookkeeping: one pass of torch reductions,
one host read.

Definitions kept (all per policy; E envs, the A agents of a policy are summed per env as the
reference's `ep_scores[policy_id] += reward[agent_id]` does):
  * an episode's score runs from the last TERMINATION of its env (cuts at max_ts_per_ep do not reset
    it, ppo.py:1837-1851 vs :1965);
  * total episodes = terminated episodes + the fractional count of the episodes still running at
    the end, `episode_lengths / avg_ep_len` (ppo.py:1940-1952);
  * "score avg" / "natural score avg" = every reward of the rollout / total episodes (:2031-2037);
  * "top score" = best natural score of a terminated episode, or of an env still running at the
    end (:1822-1824, 1960-1963, 1974-1976);
  * bootstrap range / avg over the critic's next-state rewards of the WHOLE env batch at every step
    where some env was cut, avg = sum / (steps x agents) (:1903-1911, 2058-2066);
  * longest / shortest / average episode (:1841-1847, 1967-1968, 2089-2099).
Ranks are combined like the reference's allreduces (SUM / MAX / MIN).
"""
import numpy as np
import torch

from . import mpi_utils

_FMAX = float(np.finfo(np.float32).max)


def rollout_statistics(r_env, nat_env, term, boot, boot_reward, num_agents, reward_range, nat_range,
                       obs_range, default_run, intr_env=None, intr_range=None):
    """
    r_env, nat_env [T, E] float: reward (incl. intrinsic) and natural reward per env (agents summed);
    term, boot [T, E] bool: terminal / bootstrapped episode ends; boot_reward [T, C] (every agent column);
    *_range: (min, max) device scalars or tensors.  -> dict of Python numbers (after the cross-rank reductions).
    """
    T, E = r_env.shape
    dev = r_env.device
    f64 = torch.float64
    t_idx = torch.arange(T, device=dev, dtype=torch.int64)[:, None].expand(T, E)
    # index of the last termination strictly before t (-1: none)
    last_incl = torch.cummax(torch.where(term, t_idx, torch.full_like(t_idx, -1)), dim=0).values
    last_before = torch.cat([torch.full((1, E), -1, dtype=torch.int64, device=dev), last_incl[:-1]], dim=0)
    seg_len = (t_idx - last_before).to(f64)                                   # episode_lengths after step t
    c = torch.cumsum(nat_env.to(f64), dim=0)
    base = torch.where(last_before >= 0, torch.gather(c, 0, last_before.clamp(min=0)), torch.zeros_like(c))
    seg_nat = c - base                                                        # ep_nat_scores after step t
    neg = torch.full((), -_FMAX, dtype=f64, device=dev)
    pos = torch.full((), _FMAX, dtype=f64, device=dev)
    n_term = term.sum().to(f64)
    # fractional episodes at the end (ppo.py:1940-1952)
    ep_len_end = (T - 1 - last_incl[T - 1]).to(f64)
    combined = ep_len_end.sum()
    ts_before = torch.clamp(float(T * E) - combined, min=0.0)
    cur_total = torch.where(n_term == 0, torch.ones_like(n_term), n_term)
    avg_len = torch.where(ts_before == 0, combined / E, ts_before / cur_total)
    total_eps = n_term + (ep_len_end / avg_len).sum()
    top_term = torch.where(term, seg_nat, neg).max()
    top_end = torch.where(~term[T - 1], seg_nat[T - 1], neg).max()
    top = torch.maximum(top_term, top_end)
    longest = seg_len.max()
    shortest = torch.minimum(torch.where(term, seg_len, pos).min(), torch.tensor(float(default_run), dtype=f64, device=dev))
    # average run: mean length of the episodes that terminated at the LAST step with a termination
    any_term_t = term.any(dim=1)
    last_t = torch.where(any_term_t, torch.arange(T, device=dev), torch.full((T,), -1, device=dev)).max()
    lt = last_t.clamp(min=0)
    avg_run = torch.where(last_t >= 0, (seg_len[lt] * term[lt]).sum() / term[lt].sum().clamp(min=1),
                          torch.tensor(float(default_run), dtype=f64, device=dev))
    cut_t = boot.any(dim=1)
    n_cut = cut_t.sum().to(f64)
    br = boot_reward.reshape(T, -1).to(f64)
    bs_min = torch.where(cut_t[:, None], br, pos).min()
    bs_max = torch.where(cut_t[:, None], br, neg).max()
    bs_sum = (br * cut_t[:, None]).sum()
    zero = torch.zeros((), dtype=f64, device=dev)
    sums = torch.stack([total_eps, n_cut * num_agents, r_env.sum(dtype=f64), nat_env.sum(dtype=f64), bs_sum,
                        zero if intr_env is None else intr_env.sum(dtype=f64), avg_run])
    maxs = torch.stack([top, reward_range[1].to(f64), nat_range[1].to(f64), obs_range[1].to(f64), bs_max, longest,
                        neg if intr_range is None else intr_range[1].to(f64)])
    mins = torch.stack([reward_range[0].to(f64), nat_range[0].to(f64), obs_range[0].to(f64), bs_min, shortest,
                        pos if intr_range is None else intr_range[0].to(f64)])
    if mpi_utils.distributed_path():
        import torch.distributed as dist
        stage = mpi_utils._needs_staging(sums)
        for t, op in ((sums, dist.ReduceOp.SUM), (maxs, dist.ReduceOp.MAX), (mins, dist.ReduceOp.MIN)):
            if stage:
                h = t.cpu(); dist.all_reduce(h, op=op); t.copy_(h)
            else:
                dist.all_reduce(t, op=op)
    s, mx, mn = sums.cpu().numpy(), maxs.cpu().numpy(), mins.cpu().numpy()
    total_episodes, total_bs = float(s[0]), float(s[1])
    out = {"total episodes": total_episodes,
           "score avg": float(s[2] / total_episodes), "natural score avg": float(s[3] / total_episodes),
           "top score": float(mx[0]), "reward range": (float(mn[0]), float(mx[1])),
           "natural reward range": (float(mn[1]), float(mx[2])), "obs range": (float(mn[2]), float(mx[3])),
           "bootstrap range": (float(mn[3]), float(mx[4])),
           "bootstrap avg": 0.0 if total_bs == 0 else float(s[4] / total_bs),
           "longest episode": float(mx[5]), "shortest episode": float(mn[4]),
           "average episode": float(s[6] / mpi_utils.get_num_procs())}
    if intr_env is not None:
        out["intrinsic score avg"] = float(s[5] / (total_episodes / E))
        out["intr reward range"] = (float(mn[5]), float(mx[6]))
    return out
