import sys, os, ctypes as C, time; sys.path.insert(0,'.')
import numpy as np, torch
from ppo_and_friends_amd import _lib
_lib.LIB_PATH=os.path.abspath('tools/libppoaf_hip_stamps.so')
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
for dbg in (0,4,260,4,0):
    os.environ["PPOAF_DEBUG"]=str(dbg)
    res=[]
    for mode in ("warm","real"):
        fn = (lambda: K.ppo_update_fwd_bwd(args)) if mode=="warm" else (lambda: f._one(args))
        for _ in range(20): fn()
        torch.cuda.synchronize()
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1)*1000/300)
    print("debug=%3d  fwd_bwd alone (same mini-batch, weights L2-warm) %.2f us | fwd_bwd+reduce+adam (weights rewritten each time) %.2f us"%(dbg,res[0],res[1]))
