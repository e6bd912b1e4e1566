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
    """
    num_agents > 1 gives the multi-agent shape of the MAPPO configs: every env
    holds A agents that share one policy; rows are agent-major columns c = a * E + e (the order the
    reference logs agents in, ppo.py:1730-1752); critic_view = "policy" feeds the critic the
    concatenated observations of all agents of the env (O_c = A * O), "local" the agent's own.
    All agents of an env terminate together (the reference death-masks, ppo.py:1686-1691).
    """

    def __init__(self, num_envs, obs_dim, action_space, horizon, device, reward="ones",
                 seed=1234, rank=0, term_prob=0.0, num_agents=1, critic_view="local"):
        self.num_envs = int(num_envs)
        self.num_agents = int(num_agents)
        self.agent_ids = [f"agent{a}" for a in range(self.num_agents)]
        self.obs_dim = int(obs_dim)
        self.horizon = int(horizon)
        self.device = torch.device(device)
        A, E, O = self.num_agents, self.num_envs, self.obs_dim
        self.critic_obs_dim = O * A if (critic_view == "policy" and A > 1) else O
        self.observation_space = Box(-np.inf, np.inf, (O,), np.float32)
        self.critic_observation_space = Box(-np.inf, np.inf, (self.critic_obs_dim,), np.float32)
        self.action_space = action_space
        rng = np.random.default_rng(seed + rank)
        obs = rng.standard_normal((horizon + 1, A * E, O), dtype=np.float32)
        self.obs_table = torch.from_numpy(obs).to(self.device)
        if self.critic_obs_dim != O:
            # [T+1, A, E, O] -> per env the agents' observations side by side, repeated for every agent
            v = self.obs_table.view(horizon + 1, A, E, O).permute(0, 2, 1, 3).reshape(horizon + 1, 1, E, A * O)
            self.critic_obs_table = v.expand(horizon + 1, A, E, A * O).reshape(horizon + 1, A * E, A * O).contiguous()
        else:
            self.critic_obs_table = self.obs_table
        if reward == "ones":
            rew = np.ones((horizon, A * E), dtype=np.float32)           # CartPole: +1 per step
        else:
            rew = rng.uniform(-1.0, 1.0, (horizon, A * E)).astype(np.float32)
        self.reward_table = torch.from_numpy(rew).to(self.device)
        self.term_prob = float(term_prob)
        if term_prob > 0.0:                                             # correctness-only variant
            term_env = rng.uniform(0, 1, (horizon, E)) < term_prob
            self.term_table = torch.from_numpy(np.tile(term_env, (1, A))).to(self.device)   # per column
        else:
            self.term_table = None
        self._false = torch.zeros(A * E, dtype=torch.bool, device=self.device)
        self.t = 0

    def get_batch_size(self):
        return self.num_envs

    def reset(self):
        self.t = 0
        return self.obs_table[0], self.critic_obs_table[0]

    def soft_reset(self):
        """The observation the env stopped on (ppo_env_wrappers.py:1170-1183); a hard reset before any step."""
        if self.t == 0:
            return self.reset()
        t = (self.t - 1) % self.horizon + 1
        return self.obs_table[t], self.critic_obs_table[t]

    def step(self, action):
        t = self.t % self.horizon
        nxt = self.obs_table[t + 1]
        rew = self.reward_table[t]
        term = self._false if self.term_table is None else self.term_table[t]
        self.t += 1
        return nxt, self.critic_obs_table[t + 1], rew, term, self._false, nxt


class SyntheticMixedAgentsEnv:
    """
    Agents with DIFFERENT observation / action spaces in one env (the shape of the reference's
    multi-policy baselines, e.g. baselines/pettingzoo/mpe_simple_adversary.py: an adversary and a
    team of good agents): every quantity is a dict keyed by agent id, as the reference's multi-agent
    wrappers hand them over (environments/ppo_env_wrappers.py:1075-1156), each entry a device tensor
    over the E envs of the rank.  agent_specs: [(agent_id, obs_dim, action_space), ...].  All agents
    of an env terminate together.
    """

    def __init__(self, num_envs, agent_specs, horizon, device, reward="uniform", seed=1234, rank=0, term_prob=0.0):
        self.num_envs, self.horizon, self.device = int(num_envs), int(horizon), torch.device(device)
        self.agent_ids = [a for a, _, _ in agent_specs]
        self.num_agents = len(self.agent_ids)
        E = self.num_envs
        rng = np.random.default_rng(seed + rank)
        self.observation_space, self.critic_observation_space, self.action_space = {}, {}, {}
        self.obs_table, self.reward_table = {}, {}
        for agent_id, obs_dim, action_space in agent_specs:
            sp = Box(-np.inf, np.inf, (int(obs_dim),), np.float32)
            self.observation_space[agent_id] = self.critic_observation_space[agent_id] = sp
            self.action_space[agent_id] = action_space
            self.obs_table[agent_id] = torch.from_numpy(
                rng.standard_normal((horizon + 1, E, int(obs_dim)), dtype=np.float32)).to(self.device)
            rew = np.ones((horizon, E), dtype=np.float32) if reward == "ones" else \
                rng.uniform(-1.0, 1.0, (horizon, E)).astype(np.float32)
            self.reward_table[agent_id] = torch.from_numpy(rew).to(self.device)
        self.term_table = None
        if term_prob > 0.0:
            self.term_table = torch.from_numpy(rng.uniform(0, 1, (horizon, E)) < term_prob).to(self.device)
        self._false = torch.zeros(E, dtype=torch.bool, device=self.device)
        self.t = 0

    def get_batch_size(self):
        return self.num_envs

    def _obs(self, t):
        o = {a: tab[t] for a, tab in self.obs_table.items()}
        return o, o

    def reset(self):
        self.t = 0
        return self._obs(0)

    def step(self, action):
        assert set(action.keys()) == set(self.agent_ids)
        t = self.t % self.horizon
        obs, cobs = self._obs(t + 1)
        rew = {a: tab[t] for a, tab in self.reward_table.items()}
        term = self._false if self.term_table is None else self.term_table[t]
        self.t += 1
        done = {a: term for a in self.agent_ids}
        return obs, cobs, rew, done, {a: self._false for a in self.agent_ids}, obs
