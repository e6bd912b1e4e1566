"""
Batched CartPole on the device -- the dynamics of gymnasium's CartPole-v0 / v1 (classic_control/cartpole.py:
Euler integration, tau = 0.02, force 10 N, termination at |x| > 2.4 or |theta| > 12 deg, reward 1 per step,
truncation after 200 / 500 steps) for E environments at once, in the env contract of this package
(environments/synthetic.py):

    obs, critic_obs, reward [E], terminated [E], truncated [E], terminal_obs = env.step(action)

`obs` is already the reset observation for environments that finished and `terminal_obs` the observation
they finished on (VectorizedEnv.batch_step, ppo_env_wrappers.py:1075-1156).  It is the learning check of
the reference's own test suite (test/tests/train/test_gymnasium.py:3-49: CartPole reaches a score of 200
within 70 000 timesteps) made runnable without gymnasium; the simulator itself stays torch ops -- env
adapters are outside the hot path.
"""
import math

import numpy as np
import torch

from ..spaces import Box, Discrete


class BatchedCartPoleEnv:

    gravity, masscart, masspole, length, force_mag, tau = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    x_threshold, theta_threshold = 2.4, 12 * 2 * math.pi / 360

    def __init__(self, num_envs, device, seed=0, max_episode_steps=200):
        self.num_envs, self.num_agents = int(num_envs), 1
        self.agent_ids = ["agent0"]
        self.device = torch.device(device)
        self.max_episode_steps = int(max_episode_steps)
        hi = np.array([4.8, np.inf, 0.42, np.inf], dtype=np.float32)
        self.observation_space = Box(-hi, hi, (4,), np.float32)
        self.critic_observation_space = self.observation_space
        self.action_space = Discrete(2)
        self.term_table = True                    # "may end early" marker read by PPO.rollout
        self.gen = torch.Generator(device=self.device).manual_seed(int(seed))
        self.state = torch.zeros(self.num_envs, 4, dtype=torch.float32, device=self.device)
        self.steps = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
        self._ones = torch.ones(self.num_envs, dtype=torch.float32, device=self.device)

    def get_batch_size(self):
        return self.num_envs

    def _fresh(self, n):
        return torch.rand(n, 4, generator=self.gen, device=self.device) * 0.1 - 0.05

    def reset(self):
        self.state = self._fresh(self.num_envs)
        self.steps.zero_()
        return self.state, self.state

    def soft_reset(self):
        return self.state, self.state

    def step(self, action):
        a = action.reshape(self.num_envs).to(torch.float32)
        x, x_dot, th, th_dot = self.state.unbind(1)
        force = (2.0 * a - 1.0) * self.force_mag
        total_mass = self.masspole + self.masscart
        pml = self.masspole * self.length
        cos, sin = torch.cos(th), torch.sin(th)
        temp = (force + pml * th_dot * th_dot * sin) / total_mass
        th_acc = (self.gravity * sin - cos * temp) / (self.length * (4.0 / 3.0 - self.masspole * cos * cos / total_mass))
        x_acc = temp - pml * th_acc * cos / total_mass
        nxt = torch.stack([x + self.tau * x_dot, x_dot + self.tau * x_acc,
                           th + self.tau * th_dot, th_dot + self.tau * th_acc], dim=1)
        self.steps += 1
        terminated = (nxt[:, 0].abs() > self.x_threshold) | (nxt[:, 2].abs() > self.theta_threshold)
        truncated = (~terminated) & (self.steps >= self.max_episode_steps)
        done = terminated | truncated
        fresh = self._fresh(self.num_envs)
        self.state = torch.where(done[:, None], fresh, nxt)
        self.steps = torch.where(done, torch.zeros_like(self.steps), self.steps)
        return self.state, self.state, self._ones, terminated, truncated, nxt
