"""Child process of test_rccl_fallback_loops_agree: one rank takes the N > 1 path (PPOAF_REHEARSE_MULTI_RANK=1, RCCL
all-reduce exchange) for two iterations and prints a digest of the result.  PPOAF_FALLBACK_KIND = ppo | icm | mat picks
the update whose fallback loop is exercised (K12; K12 + K14; K15)."""
import hashlib, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ppo_and_friends_amd.utils import mpi_utils
from ppo_and_friends_amd.ppo import PPO
from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
from ppo_and_friends_amd.spaces import Box, Discrete
from ppo_and_friends_amd import fused_update

kind = os.environ.get("PPOAF_FALLBACK_KIND", "ppo")
# (test-side knob of this helper, not a switch of the package: which of the two fallback loops issues the launches)
fused_update.FusedPolicyUpdate.rccl_loop = os.environ.get("PPOAF_TEST_RCCL_LOOP", "c")
mpi_utils.init_process_group_from_env()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
E, T, B, O = 16, 32, 64, 6
sp = Box(-np.inf, np.inf, (O,), np.float32)
if kind == "mat":
    from ppo_and_friends_amd.policies.mat_policy import MATPolicy
    env_gen = lambda: SyntheticFixedLengthEnv(E, O, Discrete(3), T, dev, reward="uniform", seed=11, num_agents=3)
    settings = {"p": (MATPolicy, sp, sp, Discrete(3), {})}
else:
    env_gen = lambda: SyntheticFixedLengthEnv(E, O, Discrete(3), T, dev, reward="uniform", seed=11, term_prob=0.05)
    settings = {"p": (None, sp, sp, Discrete(3), dict(enable_icm=kind == "icm"))}
ppo = PPO(env_gen, settings, device=dev, random_seed=4, normalize_obs=False, normalize_rewards=False,
          envs_per_proc=E, ts_per_rollout=T, batch_size=B, epochs_per_iter=2, update_mode="fused", save_state=False)
for _ in range(2):
    ppo.rollout(); ppo.train_on_rollout()
pol = ppo.policies["p"]
if kind == "mat":
    parts = [pol.actor_critic.flat_params, pol.actor_critic_optim.exp_avg]
    steps = int(pol.actor_critic_optim.step_count.item())
else:
    parts = [pol.policy_params, pol.policy_exp_avg]
    steps = int(pol.policy_step_counts[0].item())
    if kind == "icm":
        parts += [pol.icm_model.flat_params, pol.icm_optim.exp_avg]
        steps += int(pol.icm_optim.step_count.item())
h = hashlib.sha256()
for t in parts:
    h.update(t.detach().cpu().numpy().tobytes())
sd = ppo.status_dict["p"]
used_c = fused_update.FusedPolicyUpdate._rccl_comm_cache not in ("unset", None)
fused = [type(f).__name__ for f in ppo._fused.values() if f is not None]
print("RESULT " + json.dumps({"digest": h.hexdigest(), "c_loop": bool(used_c), "fused": sorted(fused),
                              "critic_loss": float(sd["critic loss"]), "steps": steps}), flush=True)
