"""Is a host-side torch.randperm slow while a long GPU kernel is running?"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
n = 524288
g = torch.Generator().manual_seed(1)
work = torch.empty(n, dtype=torch.int64)
def draw(label):
    t0 = time.perf_counter(); torch.randperm(n, generator=g, out=work); t1 = time.perf_counter()
    print(f"{label:55s} randperm {1e3*(t1-t0):7.2f} ms", flush=True)
print("threads", torch.get_num_threads(), "cpus", len(os.sched_getaffinity(0)))
for _ in range(3): draw("before any GPU use")
torch.cuda.init(); x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for _ in range(3): draw("GPU initialised, idle")
torch.cuda._sleep(int(2.4e9 * 0.3))          # ~0.3 s spin kernel
for _ in range(3): draw("while a spin kernel runs")
torch.cuda.synchronize()
torch.set_num_threads(1)
torch.cuda._sleep(int(2.4e9 * 0.3))
for _ in range(3): draw("while a spin kernel runs, 1 intra-op thread")
torch.cuda.synchronize()
for _ in range(2): draw("idle again, 1 thread")
