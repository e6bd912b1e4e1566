"""Diagnostic: launch-shape / unroll / cache-policy variants of the GAE streaming kernel at 2^28 transitions
(build: hipcc ... -DPPOAF_GAE_SWEEP -> tools/libppoaf_hip_sweep.so), plus array-placement skews.  Never shipped."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ppo_and_friends_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "libppoaf_hip_sweep.so")
from ppo_and_friends_amd import kernels as K

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
T = 128
Es = (1 << 28) // T
NAMES = {0: "U2 T1024 (shipped)", 1: "U4 T256", 2: "U8 T256 (round 1)", 4: "U8 T512", 13: "U4 T1024", 14: "U2 T1024", 15: "U1 T1024",
         16: "U2 T1024 nt", 17: "U2 T1024 G2", 18: "U1 T1024 G2", 19: "U2 T512 G2", 20: "U1 T1024 G4", 21: "U3 T1024", 22: "U2 T768"}


def run(r, v, b, adv, rtg, reps=20):
    """bench.py's protocol (round 4): 3 warm-ups, 20 timed launches -> (best, median, worst) GB/s."""
    for _ in range(3):
        K.gae_rtg_tmajor(r, v, b, b, None, adv_out=adv, rtg_out=rtg)
    evs = [(K.event_create(), K.event_create()) for _ in range(reps)]
    for ev in evs:
        K.gae_rtg_tmajor(r, v, b, b, None, adv_out=adv, rtg_out=rtg, timing_events=ev)
    torch.cuda.synchronize()
    s = sorted(K.event_elapsed_ms(a, c) for a, c in evs)
    g = lambda ms: 16.0 * T * Es / (ms * 1e-3) / 1e9
    return g(s[0]), g(0.5 * (s[reps // 2 - 1] + s[reps // 2])), g(s[-1])


n = T * Es
pad = 1 << 22
big = torch.empty(4 * (n + pad), dtype=torch.float32, device=dev)
b = torch.randn(Es, device=dev)
for skew in [int(x) for x in os.environ.get("SKEWS", "0,64,1088,16640").split(",")]:   # floats between the arrays' natural positions
    views = [big[i * (n + skew):i * (n + skew) + n].view(T, Es) for i in range(4)]
    views[0].uniform_(); views[1].normal_()
    for var in [int(x) for x in os.environ.get("VARIANTS", "0,1,13,14").split(",")]:
        os.environ["PPOAF_GAE_VARIANT"] = str(var)
        best, med, worst = run(views[0], views[1], b, views[2], views[3])
        print(f"skew {skew:6d}  variant {var:2d} {NAMES[var]:18s} best {best:7.1f}  median {med:7.1f}  worst {worst:7.1f} GB/s", flush=True)
# copy reference: float4 copy of the same volume (read 2 arrays, write 2)
x = big[:2 * n]; y = big[2 * (n + pad):2 * (n + pad) + 2 * n]
for _ in range(2):
    y.copy_(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); y.copy_(x); e1.record(); torch.cuda.synchronize()
print(f"torch copy of the same volume: {16.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9:7.1f} GB/s")
