"""Host-side timeline of one fused-update epoch on C2 (where does wall time go besides kernels?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ppo_and_friends_amd.ppo import PPO, PermutationLoader
from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
from ppo_and_friends_amd.spaces import Box, Discrete
dev = torch.device("cuda", 0); E, T, O = 4096, 128, 4
env_gen = lambda: SyntheticFixedLengthEnv(E, O, Discrete(2), T, dev)
sp = Box(-np.inf, np.inf, (O,), np.float32)
ppo = PPO(env_gen, {"p": (None, sp, sp, Discrete(2), {})}, device=dev, random_seed=1, envs_per_proc=E,
          ts_per_rollout=T, batch_size=256, epochs_per_iter=1)
ppo.rollout(); pol = ppo.policies["p"]
loader = PermutationLoader(pol.dataset, 256, ppo.loader_generator, ppo._perm_cache)
f = ppo._fused_updater("p", 256)
sync = torch.cuda.synchronize
for ep in range(4):
    sync(); t0 = time.perf_counter()
    perm = loader.epoch_permutation(); sync(); t1 = time.perf_counter()
    f.begin_epoch(perm); sync(); t2 = time.perf_counter()
    f.run_epoch(); t3 = time.perf_counter()
    loader.prefetch(); t4 = time.perf_counter()
    sync(); t5 = time.perf_counter()
    f.end_epoch(); t6 = time.perf_counter()
    print(f"epoch {ep}: perm {1e3*(t1-t0):.2f} ms | begin_epoch {1e3*(t2-t1):.2f} | enqueue {1e3*(t3-t2):.2f} | "
          f"prefetch {1e3*(t4-t3):.2f} | gpu drain {1e3*(t5-t4):.2f} | end_epoch {1e3*(t6-t5):.2f} | total {1e3*(t6-t0):.2f}")

# eager vs graph replay, per mini-batch
args = f._args_for(256)
for mode in ("eager", "graph", "eager", "graph"):
    f.begin_epoch(loader.epoch_permutation()); sync()
    t0 = time.perf_counter()
    if mode == "eager":
        for _ in range(2048):
            f._one(args)
    else:
        f.ppo.use_graphs = True
        f.run_epoch()
    t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    print(f"{mode}: enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms -> {1e3*(t2-t0)/2048*1e3:.1f} us per mini-batch")
