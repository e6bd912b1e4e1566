"""Diagnostic: one mini-batch's gradient bucket, split-wgrad chain vs slab chain, per parameter tensor."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

def grads(split, O, NA, B, H, depth, E, T, seed=4):
    os.environ["PPOAF_SPLIT_WGRAD"] = split
    os.environ["PPOAF_WS"] = "0"
    from ppo_and_friends_amd.ppo import PPO
    from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
    from ppo_and_friends_amd.spaces import Box, Discrete
    dev = torch.device("cuda", 0)
    env_gen = lambda: SyntheticFixedLengthEnv(E, O, Discrete(NA), T, dev, reward="uniform", seed=21)
    sp = Box(-np.inf, np.inf, (O,), np.float32)
    kw = dict(hidden_size=H, hidden_depth=depth)
    ppo = PPO(env_gen, {"p": (None, sp, sp, Discrete(NA), dict(actor_kw_args=kw, critic_kw_args=dict(kw)))}, device=dev, random_seed=seed,
              normalize_obs=False, normalize_rewards=False, envs_per_proc=E, ts_per_rollout=T, batch_size=B, epochs_per_iter=1,
              update_mode="fused", save_state=False)
    ppo.rollout()
    pol = ppo.policies["p"]
    pol.train()
    fused = ppo._fused_updater("p", B)
    assert fused.split == (split == "1"), fused.split_reason
    perm = torch.randperm(len(pol.dataset), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    fused.begin_epoch(perm)
    fused.gradient_only(fused._args_for(B))
    torch.cuda.synchronize()
    names = [(f"actor.{k}", p) for k, p in pol.actor.named_parameters()] + [(f"critic.{k}", p) for k, p in pol.critic.named_parameters()]
    base = pol.policy_params.data_ptr()
    return {n: pol.policy_grads[(p.data_ptr() - base) // 4:(p.data_ptr() - base) // 4 + p.numel()].clone().reshape(p.shape) for n, p in names}, fused.totals.clone()

for cfg in (dict(O=5, NA=3, B=32, H=128, depth=3, E=12, T=20), dict(O=1, NA=2, B=33, H=32, depth=1, E=5, T=10),
            dict(O=4, NA=2, B=64, H=128, depth=3, E=16, T=16)):
    a, ta = grads("0", **cfg)
    b, tb = grads("1", **cfg)
    print(cfg)
    for k in a:
        d = (a[k] - b[k]).abs().max().item()
        print(f"   {k:40s} max|g| {a[k].abs().max().item():.3e}  max|d| {d:.3e}" + ("   <<<<" if d > 1e-5 * max(a[k].abs().max().item(), 1e-12) + 1e-9 else ""))
    print("   totals", (ta - tb).abs().max().item())
