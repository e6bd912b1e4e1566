"""Diagnostic: per-phase cycle counts of the persistent update kernel (worker 0), C2 shapes.
Build first:  hipcc ... -DPPOAF_PERSIST_STAMPS -shared csrc/*.hip -o tools/libppoaf_hip_pstamps.so"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ppo_and_friends_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "libppoaf_hip_pstamps.so")
import argparse
ap = argparse.ArgumentParser(); ap.add_argument("--config", default="C2"); a = ap.parse_args()
sys.argv = [sys.argv[0]]
import bench
args = bench.parse(); args.config = a.config
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ppo, pol, d = bench.build_config(a.config, args, dev, 0)
ppo.epochs_per_iter = 1
ppo.rollout(); ppo.train_on_rollout()
ppo.rollout(); ppo.train_on_rollout()
f = [x for x in ppo._fused.values() if x is not None][0]
ctl = f._persist_ctl.cpu().numpy()
off = (32 * 4 + 32 * 4 + 64 * 8) // 4
ticks = ctl[off:off + 16].view(np.uint64)[:6]
n = f.n_full
names = ["fwd_bwd call", "barrier 1", "reduce + bookkeeping", "barrier 2", "adam", "barrier 3"]
tot = ticks.sum()
print(f"{a.config}: {n} mini-batches in the launch; s_memtime ticks per mini-batch (100 MHz constant clock: 10 ns each)")
for k, t in zip(names, ticks):
    print(f"  {k:22s} {t / n:9.1f} ticks = {t / n * 0.01:7.2f} us")
print(f"  total                  {tot / n:9.1f} ticks = {tot / n * 0.01:7.2f} us")
