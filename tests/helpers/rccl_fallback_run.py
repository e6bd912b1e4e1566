"""Child process of test_rccl_fallback_loops_agree: one rank takes the N > 1 path (PPOAF_REHEARSE_MULTI_RANK=1, RCCL
all-reduce exchange) for two iterations and prints a digest of the result."""
import hashlib, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ppo_and_friends_amd.utils import mpi_utils
from ppo_and_friends_amd.ppo import PPO
from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
from ppo_and_friends_amd.spaces import Box, Discrete
from ppo_and_friends_amd import fused_update

mpi_utils.init_process_group_from_env()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
E, T, B, O = 16, 32, 64, 6
env_gen = lambda: SyntheticFixedLengthEnv(E, O, Discrete(3), T, dev, reward="uniform", seed=11, term_prob=0.05)
sp = Box(-np.inf, np.inf, (O,), np.float32)
ppo = PPO(env_gen, {"p": (None, sp, sp, Discrete(3), {})}, device=dev, random_seed=4, normalize_obs=False, normalize_rewards=False,
          envs_per_proc=E, ts_per_rollout=T, batch_size=B, epochs_per_iter=2, update_mode="fused", save_state=False)
for _ in range(2):
    ppo.rollout(); ppo.train_on_rollout()
pol = ppo.policies["p"]
w = pol.policy_params.detach().cpu().numpy()
sd = ppo.status_dict["p"]
used_c = fused_update.FusedPolicyUpdate._rccl_comm_cache not in ("unset", None)
print("RESULT " + json.dumps({"digest": hashlib.sha256(w.tobytes()).hexdigest(), "c_loop": bool(used_c),
                              "critic_loss": float(sd["critic loss"]), "steps": int(pol.policy_step_counts[0].item())}), flush=True)
