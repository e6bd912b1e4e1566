"""Edge case found by the derandomised fuzz of tests/test_gpu_end_to_end.py: E = 1, T = 1, max_ts_per_ep = 1."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
sys.path.insert(0, "tests")
import test_gpu_end_to_end as t
for E, T, term, max_ts in ((1, 1, 0.9, 1), (1, 1, 0.0, 1), (3, 1, 0.9, 1), (1, 2, 0.9, 1), (1, 1, 0.9, 200), (2, 3, 0.9, 1)):
    ppo = t._make(E, T, 32, 1, term, max_ts, update_mode="torch")
    cpu = t._oracle_like(ppo, 32)
    ds = ppo.rollout()
    env = ppo.env; buf = ppo.policies["p"].buffer
    t_tab = None if env.term_table is None else env.term_table.cpu().numpy()
    ref = cpu.rollout(env.obs_table.cpu().numpy(), env.reward_table.cpu().numpy(), actions=buf.actions[..., 0].cpu().numpy(), term_table=t_tab, max_ts_per_ep=max_ts)
    print(f"E {E} T {T} term {term} max_ts {max_ts}: term_table {None if t_tab is None else t_tab.reshape(-1)[:6]}  end_kind {buf.end_kind.cpu().numpy().reshape(-1)[:6]} fixed_length {buf.fixed_length}")
    print("   rtg got", ds.rewards_to_go.cpu().numpy()[:6], "want", ref.rewards_to_go.numpy()[:6])
    print("   boot_value", buf.boot_value.cpu().numpy().reshape(-1)[:6], "boot_reward", buf.boot_reward.cpu().numpy().reshape(-1)[:6], "rewards", buf.rewards.cpu().numpy().reshape(-1)[:6])
