import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import test_gpu_full_size as t
for ws in ("0", "1"):
    os.environ["PPOAF_WS"] = ws
    for overlap in (True, False):
        ppo, E, T, A = t._c_config("C3")
        ppo.overlap_icm = overlap
        losses = []
        for i in range(4):
            ppo.rollout(); ppo.train_on_rollout()
            losses.append(round(float(ppo.status_dict["p"]["icm loss"]), 6))
        print("WS", ws, "overlap", overlap, losses, flush=True)
