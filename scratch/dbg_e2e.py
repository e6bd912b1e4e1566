import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from test_gpu_end_to_end import _make, _oracle_like, _flat_params
from ppo_and_friends_amd.ppo import PermutationLoader
E,T,B=16,32,64
ppo=_make(E,T,B,2,use_graphs=False); cpu=_oracle_like(ppo,B); pol=ppo.policies["p"]
ppo.rollout(); env=ppo.env
cpu.rollout(env.obs_table.cpu().numpy(), env.reward_table.cpu().numpy(), actions=pol.buffer.actions[...,0].cpu().numpy())
loader=PermutationLoader(pol.dataset,B,ppo.loader_generator); pol.train()
for ep in range(2):
    ppo._ppo_batch_train(loader,"p"); ref=cpu.train_epoch(); sd=ppo.status_dict["p"]
    for k in ("actor loss","critic loss","kl avg","weighted entropy"):
        print(ep,k,sd[k],ref[k],abs(sd[k]-ref[k]))
    print("actor dW", np.abs(_flat_params(pol.actor)-_flat_params(cpu.actor)).max(), "critic dW", np.abs(_flat_params(pol.critic)-_flat_params(cpu.critic)).max())
