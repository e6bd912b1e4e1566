import sys, os, ctypes as C, time; sys.path.insert(0,'.')
import numpy as np, torch
from ppo_and_friends_amd import _lib
_lib.LIB_PATH='tools/libppoaf_hip_stamps.so'
from ppo_and_friends_amd.ppo import PPO, PermutationLoader
from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
from ppo_and_friends_amd.spaces import Box, Discrete
from ppo_and_friends_amd import kernels as K
dev=torch.device('cuda',0); E,T,O=4096,128,4
env_gen=lambda: SyntheticFixedLengthEnv(E,O,Discrete(2),T,dev)
sp=Box(-np.inf,np.inf,(O,),np.float32)
ppo=PPO(env_gen,{"p":(None,sp,sp,Discrete(2),{})},device=dev,random_seed=1,envs_per_proc=E,ts_per_rollout=T,batch_size=256,epochs_per_iter=1,use_graphs=False)
ppo.rollout(); pol=ppo.policies["p"]
loader=PermutationLoader(pol.dataset,256,ppo.loader_generator)
f=ppo._fused_updater("p",256); f.begin_epoch(loader.epoch_permutation())
args=f._args_for(256)
for dbg in (0,7,7+8,7+16,7+32,7+64,7+128,7+256,7+8+16+32+64+128+256):
    os.environ["PPOAF_DEBUG"]=str(dbg)
    for _ in range(20): K.ppo_update_fwd_bwd(args)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): K.ppo_update_fwd_bwd(args)
    e1.record(); torch.cuda.synchronize()
    print("debug=%d  fwd_bwd kernel %.2f us/launch (back-to-back, same mini-batch)"%(dbg, e0.elapsed_time(e1)*1000/200))
