"""How far do two runs of the SAME three-launch chain drift apart at C4 size when one critic weight starts 1 ulp off?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import test_gpu_full_size as t
os.environ["PPOAF_WS"] = "0"
outs = []
for perturb in (False, True):
    ppo, E, T, A = t._c_config("C4")
    ppo.rollout()
    pol = ppo.policies["p"]
    buf = pol.buffer
    cobs = buf.critic_observations.reshape(-1, buf.critic_observations.shape[-1])[:4096].clone()
    if perturb:
        w = next(pol.critic.parameters())
        with torch.no_grad():
            w.view(-1)[7] = torch.nextafter(w.view(-1)[7], torch.tensor(10.0, device=w.device))
    ppo.train_on_rollout()
    with torch.no_grad():
        v = pol.critic(cobs).flatten().clone()
    outs.append((pol.policy_params.clone(), v, ppo.status_dict["p"]["critic loss"]))
d = (outs[0][0] - outs[1][0]).abs()
print("chain vs chain + 1 ulp on one critic weight: max |dw|", float(d.max()), "share > 1e-4", float((d > 1e-4).float().mean()),
      "max |dV|", float((outs[0][1] - outs[1][1]).abs().max()), "critic loss", outs[0][2], outs[1][2])
