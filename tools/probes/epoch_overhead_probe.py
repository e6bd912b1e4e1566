"""Where does an epoch's wall time go at C2 besides the update kernel?  Host timestamps (NO added device syncs) around
the pieces of the training loop; end_epoch contains the loop's own synchronisation with the device."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
sys.argv = [sys.argv[0]]
import bench
from ppo_and_friends_amd import fused_update as fu
from ppo_and_friends_amd import ppo as ppo_mod
args = bench.parse()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ppo, pol, d = bench.build_config("C2", args, dev, 0)
acc, marks = {}, []
def timed(cls, name):
    orig = getattr(cls, name)
    def wrap(self, *a, **k):
        t0 = time.perf_counter()
        r = orig(self, *a, **k)
        t1 = time.perf_counter()
        acc[name] = acc.get(name, 0.0) + t1 - t0
        marks.append((name, t0, t1))
        return r
    setattr(cls, name, wrap)
for n in ("begin_epoch", "run_epoch", "end_epoch"):
    timed(fu.FusedPolicyUpdate, n)
for n in ("epoch_permutation", "prefetch"):
    timed(ppo_mod.PermutationLoader, n)
for it in range(3):
    acc.clear(); marks.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ppo.rollout()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    ppo.train_on_rollout()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iteration {it}: rollout {1e3*(t1-t0):.1f} ms, train {1e3*(t2-t1):.1f} ms; host time inside train (10 epochs): " +
          ", ".join(f"{k} {1e3*v:.1f} ms" for k, v in acc.items()) +
          f"; outside these calls {1e3*((t2-t1)-sum(acc.values())):.1f} ms", flush=True)
    if it == 2:
        base = marks[0][1]
        for name, a, b in marks[:12]:
            print(f"    {name:18s} {1e3*(a-base):8.2f} -> {1e3*(b-base):8.2f} ms")
