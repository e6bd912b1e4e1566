"""Which statement of PermutationLoader._draw takes the time while the update kernel runs?"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
sys.argv = [sys.argv[0]]
import bench
from ppo_and_friends_amd import ppo as ppo_mod
args = bench.parse()
if os.environ.get("PROBE_THREADS"):
    torch.set_num_threads(int(os.environ["PROBE_THREADS"]))
import threading
print("intra-op threads", torch.get_num_threads(), "python threads", [t.name for t in threading.enumerate()])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ppo, pol, d = bench.build_config("C2", args, dev, 0)
log = []
def _draw(self, n):
    g = self.generator
    t = [time.perf_counter()]
    torch.empty((), dtype=torch.int64).random_(generator=g); t.append(time.perf_counter())
    c = self.prefetch_cache
    buf, work = c.get("pinned"), c.get("work")
    if buf is None or buf.numel() != n:
        buf = torch.empty(n, dtype=torch.int64, pin_memory=True); work = torch.empty(n, dtype=torch.int64)
        c["pinned"], c["work"] = buf, work
    t.append(time.perf_counter())
    torch.randperm(n, generator=g, out=work); t.append(time.perf_counter())
    buf.copy_(work); t.append(time.perf_counter())
    torch.randperm(n, generator=g, out=work); t.append(time.perf_counter())
    log.append([round(1e3 * (b - a), 2) for a, b in zip(t[:-1], t[1:])])
    return buf
ppo_mod.PermutationLoader._draw = _draw
for it in range(3):
    log.clear()
    ppo.rollout(); ppo.train_on_rollout()
    torch.cuda.synchronize()
    print("iteration", it, "per draw [seed draw, buffers, randperm -> work, work -> pinned copy, second randperm] ms:")
    worst = max(max(r) for r in log)
    print("    python threads", [t.name for t in threading.enumerate()], "worst statement", worst, "ms; total", round(sum(sum(r) for r in log), 1))
