"""Diagnostic: per-phase time of the weight-stationary persistent update kernel (worker 0 of one network).
Build first (never shipped):
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DPPOAF_WS_STAMPS [-DPPOAF_WS_STAMP_NET=1] \
        -shared ppo_and_friends_amd/csrc/*.hip -o tools/libppoaf_hip_wsstamps.so"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ppo_and_friends_amd import _lib
import argparse
ap = argparse.ArgumentParser(); ap.add_argument("--config", default="C2"); ap.add_argument("--lib", default="libppoaf_hip_wsstamps.so")
a = ap.parse_args()
_lib.LIB_PATH = os.path.join(ROOT, "tools", a.lib)
os.environ["PPOAF_WS"] = "1"
sys.argv = [sys.argv[0]]
import bench
args = bench.parse(); args.config = a.config
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ppo, pol, d = bench.build_config(a.config, args, dev, 0)
ppo.epochs_per_iter = 1
ppo.overlap_icm = False
ppo.rollout(); ppo.train_on_rollout()
ppo.rollout(); ppo.train_on_rollout()
f = [x for x in ppo._fused.values() if x is not None][0]
ctl = f._ws_ctl.cpu().numpy()
off = (32 * 4 + 2 * 32 * 4 + 2 * 32 * 8) // 4
ticks = ctl[off:off + 32].view(np.uint64)[:12]
n = f.n_full
names = ["forward tiles (all layers)", "  barriers after them", "head / loss / D_last", "  barrier", "bwd first (dgrad+wgrad+out)", "  barrier",
         "bwd middle layers", "  barriers", "W0 + bookkeeping + norm", "  barrier", "adam", "  barrier",
         "    (fwd: layer-0 fills / pre-issue)", "    (fwd l>=1: address + issue)", "    (fwd l>=1: wait + LDS stores)", "    (fwd: sync + MFMA + store + sync)"]
tot = ticks.sum()
print(f"{a.config} [{a.lib}]: {n} mini-batches in the launch; 10 ns ticks of s_memtime")
for k, t in zip(names, ticks):
    print(f"  {k:40s} {t / n * 0.01:7.2f} us")
print(f"  total                          {tot / n * 0.01:7.2f} us   (train_s {ppo.status_dict['global status']['train time']:.4f})")
