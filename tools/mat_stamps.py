"""In-kernel s_memtime stamps of K15 (diagnostic build: bash tools/build_variant.sh matstamps -DPPOAF_MAT_STAMPS):
where one launch of mat_update_fwd_bwd_kernel spends its time (workgroup 0, thread 0; shader-clock cycles)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PPOAF_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libppoaf_hip_matstamps.so")
import numpy as np, torch
import bench
from ppo_and_friends_amd import _lib
sys.argv = [sys.argv[0], "--no-graphs"]
args = bench.parse()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ppo, pol, d = bench.build_config("C5", args, dev, 0)
ppo.rollout(); pol.train()
fused = ppo._fused_updater("cartpole", args.batch_size)
fused.begin_epoch(torch.randperm(len(pol.dataset), device=dev))
a = fused._args_for(args.batch_size)
for _ in range(40):
    fused._one(a)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * 64)()
lib.ppoaf_debug_read_mat_stamps.argtypes = [C.c_void_p]
assert lib.ppoaf_debug_read_mat_stamps(buf) == 0
st = np.array(list(buf), dtype=np.int64)
tick = 1.0      # stamps are s_memtime values: shader-clock cycles (not the 100 MHz counter)
names = ["start", "rows/stats", "gather", "encoder fwd", "decoder fwd", "head+loss", "actor bwd (to 2nd attention)", "att_bwd", "actor bwd rest", "critic bwd"]
print("total %.0f cycles" % ((st[9] - st[0]) * tick))
for i in range(1, 10):
    print("  %-32s %7.0f cycles" % (names[i], (st[i] - st[i - 1]) * tick))
enc = st[16:40]
enc = enc[enc > 0]
print("encoder forward, barrier to barrier (cycles):", np.round(np.diff(np.concatenate([[st[2]], enc])) * tick, 2))
