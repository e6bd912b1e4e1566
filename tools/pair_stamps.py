"""In-kernel s_memtime stamps of the row-pair form of K12's fwd_bwd launch (ppo_update_rowpair.hpp), critic workgroup
(tile 0, first half), C3 / C4 critic width.  Diagnostic build: bash tools/build_variant.sh stamps4 -DPPOAF_STAMPS -DPPOAF_STAMP_BLOCK=4
and PPOAF_LIB=tools/libppoaf_hip_stamps4.so.  PAIRS=0: the one-workgroup body's stamps for the same block."""
import sys, os, ctypes as C; sys.path.insert(0, '.')
import numpy as np, torch
from ppo_and_friends_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get('PPOAF_LIB', 'tools/libppoaf_hip_stamps4.so'))
from ppo_and_friends_amd import fused_update
from ppo_and_friends_amd.ppo import PPO, PermutationLoader
from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
from ppo_and_friends_amd.spaces import Box, Discrete
pairs = os.environ.get("PAIRS", "1") == "1"
fused_update.FusedPolicyUpdate.row_pairs = pairs
dev = torch.device('cuda', 0); E, T, O = 1024, 128, 17
env_gen = lambda: SyntheticFixedLengthEnv(E, O, Discrete(5), T, dev)
sp = Box(-np.inf, np.inf, (O,), np.float32)
ppo = PPO(env_gen, {"p": (None, sp, sp, Discrete(5), dict(critic_kw_args=dict(hidden_size=256)))}, device=dev, random_seed=1,
          normalize_obs=False, normalize_rewards=False, envs_per_proc=E, ts_per_rollout=T, batch_size=256, epochs_per_iter=1, use_graphs=False)
ppo.rollout(); pol = ppo.policies["p"]
loader = PermutationLoader(pol.dataset, 256, ppo.loader_generator)
f = ppo._fused_updater("p", 256); f.begin_epoch(loader.epoch_permutation())
assert (f.pairs_reason() == "") == pairs, f.pairs_reason()
args = f._args_for(256)
for _ in range(50): f._one(args)
torch.cuda.synchronize()
lib = _lib.load(); buf = (C.c_ulonglong * 32)(); lib.ppoaf_debug_read_stamps.argtypes = [C.c_void_p]; lib.ppoaf_debug_read_stamps(buf)
st = np.array(list(buf), dtype=np.int64).reshape(2, 16)[0]
print("pairs" if pairs else "one workgroup per tile", "-- total cycles", st[9] - st[0])
if pairs:
    order = [(0, "start"), (1, "S0: requests, indices, statistics"), (2, "input rows"), (3, "layer 0 (all 256 columns)"),
             (8, "hidden 1: MFMAs done (wave 0)"), (11, "  next fragments requested, epilogue"), (10, "  barrier"),
             (12, "  send, publish, partner's half arrived"), (13, "  barrier"),
             (4, "hidden 2 incl. its exchange"), (5, "output layer"), (6, "head + losses"), (7, "output backward, dz_last"),
             (14, "dgrad 2 + epilogue + barrier"), (15, "  exchange of dz_1"), (9, "dgrad 1 + publish")]
else:
    order = [(0, "start"), (1, "S0"), (2, "input rows"), (3, "layer 0"), (4, "hidden forward"), (5, "output layer"), (6, "head + losses"),
             (7, "output backward"), (10, "dgrad 2"), (8, "rest of hidden backward"), (9, "layer 0 backward / publish")]
prev = st[0]
for k, name in order[1:]:
    print("   %-40s %7d" % (name, st[k] - prev)); prev = st[k]
