"""
CPU tests: the UPDATE half of the oracle (oracle/cpu_ppo_loop.py, ppo_loss_oracle.py, rollout_stats_oracle.py, ...)
against fixtures recorded from the unmodified reference's own PPO object driven for whole iterations
(tests/golden/make_golden_update.py, fixtures g12_*): identical tables, initial weights, recorded actions and
shuffles in -> datasets, per-epoch statistics, first-mini-batch losses + raw gradients, final weights,
value-normaliser state and rollout statistics out.  This is what pins the oracle's update half.
"""
import numpy as np
import pytest
import torch

from oracle import cpu_ppo_loop
from oracle import rollout_stats_oracle as rso


def _load_seq(net, g, prefix):
    """Reference FeedForwardNetwork state_dict (keys sequential_net.*) -> the oracle's nn.Sequential."""
    sd = {}
    for k in g.files:
        if k.startswith(prefix + ".sequential_net."):
            sd[k[len(prefix) + len(".sequential_net."):]] = torch.from_numpy(g[k])
    net.load_state_dict(sd)


def _flat(net):
    return torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy()


def _flat_fixture(g, prefix, net):
    return np.concatenate([g[f"{prefix}.sequential_net.{k}"].reshape(-1) for k, _ in net.named_parameters()])


def _cfg(g):
    return dict(zip([str(x) for x in g["cfg_names"]], [int(x) for x in g["cfg"]]))


@pytest.mark.parametrize("name", ["g12_c2_term", "g12_c2_cut"])
def test_cpu_port_reproduces_the_reference_ppo_iterations(golden, name):
    g = golden(name)
    c = _cfg(g)
    E, T, B = c["E"], c["T"], c["batch_size"]
    n_act = int(g["init_actor.sequential_net.3.weight"].shape[0])
    cpu = cpu_ppo_loop.CpuPPO(c["O"], n_act, batch_size=B, seed=0, rtg_accum="float32")   # NumPy-2 run recorded the fixture
    _load_seq(cpu.actor, g, "init_actor")
    _load_seq(cpu.critic, g, "init_critic")
    obs_table = g["obs_table"][:, :, 0]
    rew_table = g["reward_table"][:, :, 0]
    term = g["term_table"]
    ep = 0
    for it in range(c["iterations"]):
        acts = g["step_actions"][it * T:(it + 1) * T, :, 0]
        ds = cpu.rollout(obs_table, rew_table, actions=acts, term_table=term if term.any() else None,
                         max_ts_per_ep=c["max_ts_per_ep"])
        pre = f"it{it}_ds_"
        np.testing.assert_array_equal(ds.observations.numpy(), g[pre + "observations"])
        np.testing.assert_array_equal(ds.next_observations.numpy(), g[pre + "next_observations"])
        np.testing.assert_array_equal(ds.actions.numpy(), g[pre + "actions"])
        np.testing.assert_array_equal([e.length for e in ds.episodes], g[pre + "ep_lens"])
        tol = dict(rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(ds.values.numpy(), g[pre + "values"], **tol)
        np.testing.assert_allclose(ds.log_probs.numpy(), g[pre + "log_probs"], **tol)
        np.testing.assert_allclose(ds.rewards_to_go.numpy(), g[pre + "rewards_to_go"], **tol)
        np.testing.assert_allclose(ds.advantages.numpy(), g[pre + "advantages"], **tol)
        for e in range(c["epochs"]):
            cpu.trace = [] if ep == 0 else None
            r = cpu.train_epoch(perm=g["epoch_perms"][ep])
            if ep == 0:      # the very first mini-batch, before any optimiser step: losses and raw gradients
                m = cpu.trace[0]
                np.testing.assert_allclose([m["actor"], m["critic"]], g["mb0_losses"], rtol=1e-6, atol=1e-9)
                np.testing.assert_allclose(m["actor_grad"], g["mb0_actor_grad"], rtol=1e-5, atol=1e-9)
                np.testing.assert_allclose(m["critic_grad"], g["mb0_critic_grad"], rtol=1e-5, atol=1e-8)
            got = np.array([r["actor loss"], r["critic loss"], r["kl avg"], r["weighted entropy"]])
            np.testing.assert_allclose(got, g["epoch_stats"][ep], rtol=2e-6, atol=1e-8, err_msg=f"iteration {it} epoch {e}")
            ep += 1
    for net, tag in ((cpu.actor, "final_actor"), (cpu.critic, "final_critic")):
        np.testing.assert_allclose(_flat(net), _flat_fixture(g, tag, net), rtol=1e-5, atol=1e-7, err_msg=tag)
    vs = cpu.value_stats
    np.testing.assert_allclose([vs.mean, vs.variance, vs.count], g["value_stats"], rtol=1e-6)
