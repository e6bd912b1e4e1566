"""Launches the GAE kernels a few times (config size and saturating size) for rocprofv3 --pmc runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppo_and_friends_amd import kernels as K

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
for (T, E, reps) in ((128, 4096, 5), (128, (1 << 28) // 128, 3)):
    r = torch.rand(T, E, device=dev); v = torch.randn(T, E, device=dev); b = torch.randn(E, device=dev)
    adv = torch.empty_like(r); rtg = torch.empty_like(r)
    for _ in range(reps):
        K.gae_rtg_tmajor(r, v, b, b, None, adv_out=adv, rtg_out=rtg)
    torch.cuda.synchronize()
    del r, v, b, adv, rtg
print("done")
