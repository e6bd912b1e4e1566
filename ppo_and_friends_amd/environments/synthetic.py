"""
Batched, device-resident environments for the rollout engine.

The reference steps `envs_per_proc` Python simulators one after another
(environments/ppo_env_wrappers.py:1122-1148).  Third-party simulators are out of
scope; what the hot path needs from an environment is the
step / auto-reset contract of VectorizedEnv.batch_step (:1075-1156):

    obs, critic_obs, reward [E], terminated [E], truncated [E], terminal_obs = env.step(action)

with `obs` already the reset observation for environments that finished and
`terminal_obs` the observation they finished on.  Everything is a device tensor.

SyntheticFixedLengthEnv is the measurement workload of BASELINE.md §3 /
SURVEY.md §8(d): pre-generated N(0,1) observations `[T+1, E, O]`, constant (or
U(-1,1)) rewards, no terminations -> fixed-length trajectories.
"""
import numpy as np
import torch

from ..spaces import Box, Discrete


class SyntheticFixedLengthEnv:

    def __init__(self, num_envs, obs_dim, action_space, horizon, device, reward="ones",
                 seed=1234, rank=0, term_prob=0.0):
        self.num_envs = int(num_envs)
        self.obs_dim = int(obs_dim)
        self.horizon = int(horizon)
        self.device = torch.device(device)
        self.observation_space = Box(-np.inf, np.inf, (obs_dim,), np.float32)
        self.action_space = action_space
        rng = np.random.default_rng(seed + rank)
        obs = rng.standard_normal((horizon + 1, num_envs, obs_dim), dtype=np.float32)
        self.obs_table = torch.from_numpy(obs).to(self.device)
        if reward == "ones":
            rew = np.ones((horizon, num_envs), dtype=np.float32)        # CartPole: +1 per step
        else:
            rew = rng.uniform(-1.0, 1.0, (horizon, num_envs)).astype(np.float32)
        self.reward_table = torch.from_numpy(rew).to(self.device)
        self.term_prob = float(term_prob)
        if term_prob > 0.0:                                             # correctness-only variant
            self.term_table = torch.from_numpy(rng.uniform(0, 1, (horizon, num_envs)) < term_prob).to(self.device)
        else:
            self.term_table = None
        self._false = torch.zeros(num_envs, dtype=torch.bool, device=self.device)
        self.t = 0

    def get_batch_size(self):
        return self.num_envs

    def reset(self):
        self.t = 0
        o = self.obs_table[0]
        return o, o

    soft_reset = reset

    def step(self, action):
        t = self.t % self.horizon
        nxt = self.obs_table[t + 1]
        rew = self.reward_table[t]
        term = self._false if self.term_table is None else self.term_table[t]
        self.t += 1
        return nxt, nxt, rew, term, self._false, nxt
