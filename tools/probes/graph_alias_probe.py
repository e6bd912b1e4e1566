"""
Diagnostic (round 3): which tensor of the torch-path mini-batch step is stale when its hipGraph is replayed?
Runs the reference fixture g12_c3_gauss through the product's torch update path twice in one process (fused first, to
leave the caching allocator with free blocks, as the test suite does), checking after every replay that the gathered
mini-batch fields equal a direct gather, and printing the per-mini-batch loss scalars of a graph run next to an eager run.
"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_reference_golden as T

def golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)

def run(name, mode, graphs, check):
    os.environ["PPOAF_GRAPHS"] = "1" if graphs else "0"
    g = golden(name)
    dev = torch.device("cuda", 0)
    ppo, pol, c, continuous = T.make_product(g, name, mode, dev)
    E, Tt, A, B = c["E"], c["T"], c["A"], c["batch_size"]
    raw = T.agent_major(g["step_raw_actions"][:Tt])
    ppo.replay_raw_actions = torch.from_numpy(raw if continuous else raw.reshape(Tt, A * E, 1)).to(dev)
    ds = ppo.rollout()
    pol.train()
    log = []
    if check:
        buf = pol.buffer
        N = buf.num_transitions
        orig_step, orig_opt = ppo._minibatch_step, ppo._optimizer_step
        capturing = lambda: torch.cuda.is_current_stream_capturing()
        def seg(t, net):
            base = pol.policy_params.data_ptr()
            return torch.cat([t[(p.data_ptr() - base) // 4:(p.data_ptr() - base) // 4 + p.numel()] for p in net.parameters()])
        def snap(tag):
            if capturing():
                return
            torch.cuda.synchronize()
            tot = ppo._graphs[("scratch", "totals")].clone().cpu().numpy()
            ls = pol.actor.distribution.log_std
            log.append((tag, tot[:5].copy(), seg(pol.policy_grads, pol.actor).double().abs().sum().item(),
                        seg(pol.policy_grads, pol.critic).double().abs().sum().item(),
                        seg(pol.policy_params, pol.actor).double().abs().sum().item(),
                        seg(pol.policy_params, pol.critic).double().abs().sum().item(),
                        ls.grad.double().abs().sum().item(), ls.double().sum().item(),
                        pol.actor_optim.grad_norm.item(), pol.critic_optim.grad_norm.item()))
        orig_roc = ppo._replay_or_capture
        def roc(key, fn):
            orig_roc(key, fn)
            snap(key[0])
        ppo._replay_or_capture = roc
        def step(*a, **k):
            r = orig_step(*a, **k)
            if not ppo.use_graphs: snap("fb")
            return r
        def opt(*a, **k):
            r = orig_opt(*a, **k)
            if not ppo.use_graphs: snap("opt")
            return r
        ppo._minibatch_step, ppo._optimizer_step = step, opt
    ppo._ppo_batch_train(T.FixedPermLoader(pol.dataset, B, g["epoch_perms"][0]), "agent")
    sd = ppo.status_dict["agent"]
    got = np.array([sd["actor loss"], sd["critic loss"], sd["kl avg"], sd["weighted entropy"]])
    print(f"{name} {mode} graphs={graphs}: epoch stats {got}  want {g['epoch_stats'][0]}  dev {np.abs(got - g['epoch_stats'][0]).max():.2e}", flush=True)
    return log

name = sys.argv[1] if len(sys.argv) > 1 else "g12_c3_gauss"
for n in ("g12_c2_cut", "g12_c2_icm", "g12_c2_term", name):
    run(n, "fused", True, False)                # what the suite ran before (allocator state)
lg = run(name, "torch", True, True)
le = run(name, "torch", False, True)
names = ["tag", "totals", "|g_a|", "|g_c|", "|p_a|", "|p_c|", "|g_ls|", "sum ls", "norm_a", "norm_c"]
for i, (a, b) in enumerate(zip(lg, le)):
    print(i, "GRAPH", a[0], np.array2string(a[1], precision=7), " ".join(f"{x:.7f}" for x in a[2:]))
    print(i, "EAGER", b[0], np.array2string(b[1], precision=7), " ".join(f"{x:.7f}" for x in b[2:]))
