import sys, time; sys.path.insert(0, '.')
import torch, torch.nn as nn
from ppo_and_friends_amd.ppo import PPO
from ppo_and_friends_amd.environments.cartpole import BatchedCartPoleEnv
from ppo_and_friends_amd.spaces import Discrete
dev = torch.device("cuda", 0); E = int(sys.argv[1]) if len(sys.argv) > 1 else 16
env_gen = lambda: BatchedCartPoleEnv(E, dev, seed=0)
probe = env_gen(); act = dict(activation=nn.LeakyReLU())
ppo = PPO(env_gen, {"p": (None, probe.observation_space, probe.observation_space, Discrete(2),
                          dict(lr=2e-3, actor_kw_args=act, critic_kw_args=dict(act)))},
          device=dev, random_seed=2, envs_per_proc=E, ts_per_rollout=256, max_ts_per_ep=32, batch_size=256,
          obs_clip=(-10.0, 10.0), reward_clip=(-10.0, 10.0), save_state=False)
t0 = time.time()
for it in range(40):
    ppo.learn(E * 256)
    gs, sd = ppo.status_dict["global status"], ppo.status_dict["p"]
    print(f"it {it:2d} timesteps {gs['timesteps']:7d} natural score avg {sd['natural score avg']:7.2f} top {sd['top score']:6.1f} "
          f"kl {sd['kl avg']:.4f} wall {time.time() - t0:5.1f}s")
    if sd["natural score avg"] >= 199.0: break
