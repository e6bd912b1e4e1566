"""Does a config's speed depend on which configs ran before it in the same process?"""
import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--no-saturating"]
args = bench.parse()
import torch
from ppo_and_friends_amd.utils import mpi_utils
mpi_utils.init_process_group_from_env()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
for name in sys.stdin.read().split():
    if name == "gc":
        gc.collect(); torch.cuda.empty_cache(); print("gc.collect()"); continue
    if name == "burn":
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            torch.zeros(1, device="cuda")
        torch.cuda.synchronize(); print("burned", hex(st.cuda_stream)); continue
    name, steps, warm, gae = (name.split(":") + ["2", "1", "0"])[:4]
    r = bench.run_config(name, args, dev, 0, 1, int(steps), int(warm), gae == "1")
    print(name, r["value"], "ms/step", r["ms_per_step"], "rollout_s", r["rollout_s"], "train_s", r["train_s"],
          "mem MB", torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, flush=True)
