"""Does a stream of EMPTY launches slow a running update chain down?  C3's PPO epoch (split-wgrad launch chain, hipGraph
replay) alone, then beside a second stream that replays a hipGraph of N one-thread spin kernels per PPO mini-batch, paced to fill the mini-batch's time (each launch
boundary writes back / invalidates the XCDs' L2 and costs the command processor a dispatch) -- the ICM chain's 4 launches
per mini-batch without its work.   python tools/probes/launch_boundary_probe.py [launches_per_minibatch ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("PPOAF_WS", "0")
import torch
import bench
from ppo_and_friends_amd import kernels as K

rates = [int(x) for x in sys.argv[1:]] or [0, 4, 8]
sys.argv = [sys.argv[0]]
args = bench.parse()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ppo, pol, d = bench.build_config("C3", args, dev, 0)
ppo.rollout(); pol.train()
fused = ppo._fused_updater("cartpole", args.batch_size)
N = len(pol.dataset)
n_mb = N // args.batch_size
tiny = torch.zeros(64, device=dev)
main = torch.cuda.current_stream()
sa, sb = K.concurrent_stream_pair(dev)

def epoch(per_mb):
    fused.begin_epoch(torch.randperm(N, device=dev))
    g = None
    if per_mb:
        chunk = 128 * per_mb
        spin = int(50e-6 / per_mb * 2.0e9)               # ~50 us of spinning per mini-batch in all: the stream keeps pace with the chain
        s = torch.cuda.Stream(); s.wait_stream(main)
        with torch.cuda.stream(s):
            for _ in range(8): torch.cuda._sleep(spin)
        main.wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(chunk): torch.cuda._sleep(spin)
    torch.cuda.synchronize()
    sa.wait_stream(main); sb.wait_stream(main)
    t0 = time.perf_counter()
    with torch.cuda.stream(sa):
        fused.run_epoch()
    if g is not None:
        with torch.cuda.stream(sb):
            for _ in range(n_mb // 128): g.replay()
    sa.synchronize()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    fused.end_epoch()
    return (t1 - t0) / n_mb * 1e6

for r in rates:
    epoch(r)                                         # graphs captured, caches warm
    us = [epoch(r) for _ in range(3)]
    print(f"{r} empty launches per mini-batch on a second stream: PPO chain {min(us):.1f} us per mini-batch (runs {['%.1f' % u for u in us]})", flush=True)
