import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import test_gpu_full_size as t
outs = []
for ws, mode in (("1", "auto"), ("0", "auto"), ("1", "rowtile")):
    os.environ["PPOAF_WS"] = ws; os.environ["PPOAF_WS_MODE"] = mode
    ppo, E, T, A = t._c_config("C4")
    ppo.rollout()
    pol = ppo.policies["p"]
    buf = pol.buffer
    obs = buf.observations.reshape(-1, buf.observations.shape[-1])[:4096].clone()
    cobs = buf.critic_observations.reshape(-1, buf.critic_observations.shape[-1])[:4096].clone()
    ppo.train_on_rollout()
    sd = ppo.status_dict["p"]
    with torch.no_grad():
        v = pol.critic(cobs).flatten().clone(); lg = pol.actor(obs).clone()
    outs.append((pol.policy_params.clone(), [sd[k] for k in ("actor loss", "critic loss", "kl avg", "weighted entropy")], v, lg))
    print(ws, mode, outs[-1][1])
for i, j in ((0, 1), (2, 1)):
    d = (outs[i][0] - outs[j][0]).abs()
    na = 0
    print("pair", i, j, "max |dw|", float(d.max()), "share > 1e-4", float((d > 1e-4).float().mean()),
          "max |dV|", float((outs[i][2] - outs[j][2]).abs().max()), "V scale", float(outs[j][2].abs().mean()),
          "max |dlogit|", float((outs[i][3] - outs[j][3]).abs().max()))
