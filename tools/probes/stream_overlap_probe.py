"""Do two fresh torch streams run concurrently?  Pair after pair from torch's pool, one spin kernel on each."""
import time
import torch

torch.cuda.init()
cyc = 40_000_000
torch.cuda._sleep(cyc); torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(cyc); torch.cuda.synchronize(); one = time.perf_counter() - t0
print(f"one spin kernel: {one*1e3:.1f} ms")
for i in range(24):
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(sa):
        torch.cuda._sleep(cyc)
    with torch.cuda.stream(sb):
        torch.cuda._sleep(cyc)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"pair {i:2d}  streams {sa.cuda_stream:#x} {sb.cuda_stream:#x}  wall {dt*1e3:6.1f} ms  ratio {dt/one:.2f}", flush=True)
    if i % 3 == 2:
        extra = torch.cuda.Stream()          # shift the pool's parity
        with torch.cuda.stream(extra):
            torch.zeros(1, device="cuda")
