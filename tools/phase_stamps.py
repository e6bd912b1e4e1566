"""In-kernel s_memtime stamps of K12's row-tiled fwd_bwd kernel (diagnostic build: bash tools/build_stamps.sh, or
bash tools/build_variant.sh <name> -DPPOAF_STAMPS [-DPPOAF_STAMP_BLOCK=4 for a critic workgroup] and PPOAF_LIB=tools/libppoaf_hip_<name>.so):
shader-clock cycles per phase of one workgroup.  CRITIC_H=256 gives the C3 / C4 critic shape.  Note: a stamp is a global
store, and the backend drains vmcnt before the loads that follow it -- the phase after a stamp includes that drain."""
import sys, os, ctypes as C; sys.path.insert(0,'.')
import numpy as np, torch
from ppo_and_friends_amd import _lib
_lib.LIB_PATH=os.path.abspath(os.environ.get('PPOAF_LIB','tools/libppoaf_hip_stamps.so'))
from ppo_and_friends_amd.ppo import PPO, PermutationLoader
from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
from ppo_and_friends_amd.spaces import Box, Discrete
import os
dev=torch.device('cuda',0); E,T,O=4096,128,4
CH=int(os.environ.get('CRITIC_H','128'))
env_gen=lambda: SyntheticFixedLengthEnv(E,O,Discrete(2),T,dev)
sp=Box(-np.inf,np.inf,(O,),np.float32)
ppo=PPO(env_gen,{"p":(None,sp,sp,Discrete(2),dict(critic_kw_args=dict(hidden_size=CH)))},device=dev,random_seed=1,normalize_obs=False,normalize_rewards=False,envs_per_proc=E,ts_per_rollout=T,batch_size=256,epochs_per_iter=1,use_graphs=False)
ppo.rollout(); pol=ppo.policies["p"]
loader=PermutationLoader(pol.dataset,256,ppo.loader_generator)
f=ppo._fused_updater("p",256); f.begin_epoch(loader.epoch_permutation())
args=f._args_for(256)
for _ in range(50): f._one(args)
torch.cuda.synchronize()
lib=_lib.load(); buf=(C.c_ulonglong*32)(); lib.ppoaf_debug_read_stamps.argtypes=[C.c_void_p]; print(lib.ppoaf_debug_read_stamps(buf))
st=np.array(list(buf),dtype=np.int64).reshape(2,16)
names=["P0 idx/stats","P1 gatherX","P2 layer0","P3 hidden fwd","P4 out","P5 head","P6 out bwd","P7 hidden bwd","P8 layer0 bwd"]
for w in (0,1):
    d=np.diff(st[w,:10]); print("net",w,"total cycles",st[w,9]-st[w,0])
    for n,x in zip(names,d): print("   %-16s %7d"%(n,x))
    print("   first bwd layer: dgrad %d, wgrad %d, bias sums %d, barrier %d"%(st[w,10]-st[w,7], st[w,11]-st[w,10], st[w,12]-st[w,11], st[w,13]-st[w,12]))
