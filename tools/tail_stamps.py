"""In-kernel s_memtime stamps of the fused tail launch of K12 (csrc/ppo_update_tail.hip), diagnostic build:
    bash tools/build_variant.sh tailstamps -DPPOAF_TAIL_STAMPS [-DPPOAF_TAIL_STAMP_BLOCK=<workgroup>]
    PPOAF_LIB=tools/libppoaf_hip_tailstamps.so python tools/tail_stamps.py
Shader-clock cycles per phase of ONE workgroup of the last launch, plus the launch's own duration from begin / end events.
CRITIC_H=256 gives the C3 / C4 critic shape."""
import sys, os, ctypes as C; sys.path.insert(0, '.')
import numpy as np, torch
from ppo_and_friends_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get('PPOAF_LIB', 'tools/libppoaf_hip_tailstamps.so'))
from ppo_and_friends_amd.ppo import PPO, PermutationLoader
from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
from ppo_and_friends_amd.spaces import Box, Discrete
from ppo_and_friends_amd import kernels as K
dev = torch.device('cuda', 0); E, T, O = 4096, 128, 4
CH = int(os.environ.get('CRITIC_H', '128'))
env_gen = lambda: SyntheticFixedLengthEnv(E, O, Discrete(2), T, dev)
sp = Box(-np.inf, np.inf, (O,), np.float32)
ppo = PPO(env_gen, {"p": (None, sp, sp, Discrete(2), dict(critic_kw_args=dict(hidden_size=CH)))}, device=dev, random_seed=1,
          normalize_obs=False, normalize_rewards=False, envs_per_proc=E, ts_per_rollout=T, batch_size=256, epochs_per_iter=1, use_graphs=False)
ppo.rollout(); pol = ppo.policies["p"]
loader = PermutationLoader(pol.dataset, 256, ppo.loader_generator)
f = ppo._fused_updater("p", 256); f.begin_epoch(loader.epoch_permutation())
assert f.tail_reason() == "", f.tail_reason()
args = f._args_for(256)
lib, st = _lib.load(), K.stream()
for _ in range(50):
    f._one(args)
torch.cuda.synchronize()
ev = [(K.event_create(), K.event_create()) for _ in range(20)]
durs = []
for e0, e1 in ev:
    _lib.check(lib.ppoaf_ppo_update_fwd_bwd(C.byref(args), st), "fwd_bwd")
    _lib.check(lib.ppoaf_ppo_update_wgrad_adam_timed(C.byref(args), f._tail_ctl_ptr(args), 2.0, e0, e1, st), "tail")
torch.cuda.synchronize()
durs = sorted(K.event_elapsed_ms(e0, e1) * 1e3 for e0, e1 in ev)
print("tail launch, begin -> end events: median %.2f us  min %.2f  max %.2f" % (durs[len(durs) // 2], durs[0], durs[-1]))
ctl = f._tail_ctl.cpu().numpy()
print("error word", ctl[2], "launches completed", ctl[0])
stamps = ctl.view(np.int64)[8:8 + 24]
names = ["start -> loads issued", "loads landed + MFMA", "fold + tile to LDS (wave 0)", "wave sum of the norm partial",
         "publish + wait for all records + sums", "Adam on own elements (stores issued)"]
print("workgroup", os.environ.get("PPOAF_TAIL_STAMP_BLOCK_SHOWN", "0"), "total cycles", stamps[6] - stamps[0])
for n, d in zip(names, np.diff(stamps[:7])):
    print("   %-48s %7d" % (n, d))
