"""
Launches the update kernels of one BASELINE config eagerly (no hipGraph: counters are collected per dispatch) for
rocprofv3 --pmc passes:  python3 tools/update_pmc_driver.py --config C2|C3|C4|C5 [--minibatches 16]

  C2  K12 three-launch chain  fwd_bwd<8,8> -> reduce -> adam            (16 mini-batches)
  C3  the same chain <8,16> beside K14's  encoder_fwd -> heads -> encoder_bwd -> reduce  (16 mini-batches each)
  C4  ONE launch of the two-XCD persistent kernel over --minibatches mini-batches (+ the chain for comparison)
  C5  K15  mat_update_fwd_bwd -> reduce -> clip_adam                   (16 mini-batches)
Every launch sees the weights the previous Adam launch rewrote, as in a training epoch.
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2")
ap.add_argument("--minibatches", type=int, default=16)
a = ap.parse_args()
sys.argv = [sys.argv[0], "--no-graphs"]                    # bench.py's defaults: E, T, batch 256 ...; eager launches
args = bench.parse()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ppo, pol, d = bench.build_config(a.config, args, dev, 0)
ppo.rollout()
pol.train()
B, n = args.batch_size, a.minibatches
N = pol.buffer.num_transitions if not pol.agent_grouping else len(pol.dataset)
perm = torch.randperm(len(pol.dataset), device=dev)


def eager(fused, n_mb):
    fused.begin_epoch(perm)
    keep = (fused.n_full, fused.tail)
    fused.n_full, fused.tail = n_mb, 0
    try:
        fused.run_epoch()
        torch.cuda.synchronize()
    finally:
        fused.n_full, fused.tail = keep


fused = ppo._fused_updater("cartpole", B)
assert fused is not None
eager(fused, n)
if pol.enable_icm:
    icm = ppo._fused_icm_updater("cartpole")
    icm.begin_epoch(perm)
    keep = (icm.n_full, icm.tail)
    icm.n_full, icm.tail = n, 0
    icm.run_epoch()
    torch.cuda.synchronize()
    icm.n_full, icm.tail = keep
print("done", a.config, n)
