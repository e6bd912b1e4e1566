"""Shared by make_golden_state.py (build container, next to the reference) and tests/test_checkpoint_crossread.py: the
product-side configuration of the checkpoint cross-read fixtures (g14_*)."""
import numpy as np
import torch

E, T, O, NA, B, SEED = 4, 8, 4, 2, 16, 140


def product_ppo(state_path, device="cpu", load_state=False):
    """The product configured like the reference run of make_golden_state.reference_ppo (same shapes, same switches)."""
    from ppo_and_friends_amd.ppo import PPO
    from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
    from ppo_and_friends_amd.spaces import Box, Discrete
    dev = torch.device(device)
    env_gen = lambda: SyntheticFixedLengthEnv(E, O, Discrete(NA), T, dev, reward="uniform", seed=SEED)
    sp = Box(-np.inf, np.inf, (O,), np.float32)
    return PPO(env_gen, {"agent": (None, sp, sp, Discrete(NA), dict(actor_kw_args=dict(hidden_size=32), critic_kw_args=dict(hidden_size=32)))},
               device=dev, random_seed=SEED, envs_per_proc=E, ts_per_rollout=T, batch_size=B, epochs_per_iter=2,
               normalize_obs=True, normalize_rewards=True, obs_clip=(-5.0, 5.0), reward_clip=(-5.0, 5.0),
               state_path=state_path, load_state=load_state)


def fill_product_state(ppo):
    """A recognisable, seed-determined state: weights as initialised, Adam moments / statistics from a fixed generator."""
    pol = ppo.policies["agent"]
    g = torch.Generator().manual_seed(SEED + 1)
    pol.policy_exp_avg.copy_(torch.randn(pol.policy_exp_avg.numel(), generator=g) * 1e-2)
    pol.policy_exp_avg_sq.copy_(torch.rand(pol.policy_exp_avg_sq.numel(), generator=g) * 1e-3)
    pol.policy_step_counts.fill_(37)
    pol.actor_optim.set_lr(2.5e-4); pol.critic_optim.set_lr(2.5e-4)
    rs = ppo.value_normalizers["agent"].running_stats
    rs.mean_t.fill_(0.75); rs.var_t.fill_(2.25); rs.count_t.fill_(321.0001)
    ppo.status_dict["global status"]["iteration"] = 5
    ppo.status_dict["global status"]["timesteps"] = 5 * E * T
    return ppo
