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
from oracle import filter_oracle
from oracle import icm_oracle
from oracle import lstm_oracle
from oracle import ppo_loss_oracle as plo
from oracle import running_stats_oracle as rso
from oracle import mat_oracle
from oracle.rollout_stats_oracle import rollout_statistics_loop


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


def row_mapping(ref_obs, got_obs):
    """
    pi with got[pi[i]] == ref[i].  The reference orders the episodes of one end-of-episode event by the policy's
    agent_ids, which PPOPolicy.register_agent builds through a set of strings (policies/ppo_policy.py:375-376):
    a hash-dependent agent order, not a contract.  The port / product use agent order 0..A-1.  Rows are matched by
    their (unique, random) observation vectors; every other field is then compared under this mapping and the
    recorded shuffles are mapped through it, so mini-batches hold the same transitions.
    """
    index = {row.tobytes(): i for i, row in enumerate(np.ascontiguousarray(got_obs))}
    assert len(index) == len(got_obs), "observation rows are not unique"
    return np.array([index[row.tobytes()] for row in np.ascontiguousarray(ref_obs)], dtype=np.int64)


def agent_major(x):
    """[steps, E, A, ...] (fixture layout) -> [steps, A*E, ...] agent-major columns (the oracle's / product's rows)."""
    x = np.swapaxes(x, 1, 2)
    return x.reshape((x.shape[0], x.shape[1] * x.shape[2]) + x.shape[3:])


def policy_view(obs):
    """critic_view="policy": per env the agents' observations side by side, the same row for every agent."""
    S, E, A, O = obs.shape
    v = obs.reshape(S, 1, E, A * O)
    return np.broadcast_to(v, (S, A, E, A * O)).reshape(S, A * E, A * O)


def build_cpu_port(g, c):
    """The CPU port configured like the scenario's reference policy, holding the fixture's initial weights."""
    names = set(g.files)
    continuous = "init_actor.distribution.log_std" in names
    n_act = int(g["init_actor.sequential_net.3.weight"].shape[0])
    c_in = int(g["init_critic.sequential_net.0.weight"].shape[1])
    c_hidden = int(g["init_critic.sequential_net.0.weight"].shape[0])
    hidden = int(g["init_actor.sequential_net.0.weight"].shape[0])
    extra = {k: v for k, v in SCENARIOS[c["name"]].items() if k != "filters"}
    cpu = cpu_ppo_loop.CpuPPO(c["O"], n_act, hidden=hidden, batch_size=c["batch_size"], seed=0,
                              rtg_accum="float32",                      # the fixtures were recorded under NumPy 2
                              critic_obs_dim=c_in, critic_hidden=c_hidden, continuous=continuous,
                              enable_icm=any(k.startswith("init_icm.") for k in names), **extra)
    _load_seq(cpu.actor, g, "init_actor")
    _load_seq(cpu.critic, g, "init_critic")
    if continuous:
        with torch.no_grad():
            cpu.log_std.copy_(torch.from_numpy(g["init_actor.distribution.log_std"]))
    if cpu.enable_icm:
        cpu.icm.load_state_dict({k[len("init_icm."):]: torch.from_numpy(g[k]) for k in names if k.startswith("init_icm.")})
    return cpu


def final_params(g, tag, cpu):
    if tag == "actor":
        got = torch.cat([p.detach().reshape(-1) for p in cpu.actor_params]).numpy()
        want = [g[f"final_actor.sequential_net.{k}"].reshape(-1) for k, _ in cpu.actor.named_parameters()]
        if cpu.continuous:
            want.append(g["final_actor.distribution.log_std"])
        return got, np.concatenate(want)
    if tag == "critic":
        return _flat(cpu.critic), _flat_fixture(g, "final_critic", cpu.critic)
    got = torch.cat([p.detach().reshape(-1) for p in cpu.icm.parameters()]).numpy()
    return got, np.concatenate([g[f"final_icm.{k}"].reshape(-1) for k, _ in cpu.icm.named_parameters()])


LEAKY = dict(activation=torch.nn.LeakyReLU())
SCENARIOS = {
    "g12_c2_term": {}, "g12_c2_cut": {}, "g12_c4_mappo": LEAKY,
    "g12_c3_gauss": dict(lr=1e-4, **LEAKY),
    "g12_gauss_bounds": dict(act_low=[-1.0, -2.0, 0.0], act_high=[1.0, 2.0, 5.0]),
    "g12_c2_icm": {},
    "g12_c3_full": dict(lr=1e-4, filters=dict(obs_clip=(-2.0, 2.0), reward_clip=(-1.5, 1.5)), **LEAKY),
    # batch_size = 256, the reference's default and the metric's mini-batch shape (ppo.py:134)
    "g12_c2_b256": {}, "g12_c4_b256": LEAKY,
    "g12_c3_b256": dict(lr=1e-4, filters=dict(obs_clip=(-2.0, 2.0), reward_clip=(-1.5, 1.5)), **LEAKY),
    # KL early stop (ppo.py:2221-2232): the reference left the epoch loop after 2 of 4 epochs (and 4 / 3 in iteration 1)
    "g12_c2_klstop": dict(lr=1e-3), "g12_c2_icm_klstop": dict(lr=3e-3),
}


class RankView:
    """Rank r's arrays of a fixture that holds R ranks of the reference (keys `r<rank>.<key>`, make_golden_update.py:
    rank_scenarios) behind the key names of a single-rank fixture."""

    def __init__(self, g, rank):
        self._g, self._p = g, f"r{rank}."
        self.files = [k[len(self._p):] for k in g.files if k.startswith(self._p)]

    def __getitem__(self, k):
        return self._g[self._p + k]


def replay_fixture(g, name, comm=None):
    """
    One rank of a g12 fixture through the CPU port: the reference's tables, initial weights, recorded actions and shuffles
    in; datasets, first-mini-batch losses + raw gradients, every epoch's statistics (and the decision to stop early),
    final weights and normaliser states checked against what the reference produced.  `comm` (R > 1 fixtures): the
    collective surface of cpu_ppo_loop.ThreadComm / cpu_ddppo.GlooComm; the statistics, averaged gradients and weights
    recorded on this rank are then the all-reduced ones.
    """
    c = _cfg(g)
    c["name"] = name
    E, T, B, A = c["E"], c["T"], c["batch_size"], c["A"]
    cpu = build_cpu_port(g, c)
    c_in = cpu.critic[0].weight.shape[1]
    obs_table = agent_major(g["obs_table"])
    cobs_table = policy_view(g["obs_table"]) if c_in != c["O"] else None
    rew_table = agent_major(g["reward_table"])
    term = np.tile(g["term_table"], (1, A))
    ep = icm_ep = 0
    filters = SCENARIOS[name].get("filters")
    raw_obs, raw_rew = obs_table, rew_table
    if filters is not None:                    # wrapper_utils.py:81-111: the env's streams pass the filter stack first
        assert cobs_table is None
        orc = filter_oracle.FilteredEnvOracle(A, E, c["O"], c["O"], True, True, filters["obs_clip"], filters["reward_clip"],
                                              gamma=0.99)
    for it in range(c["iterations"]):
        if filters is not None:
            # the table env's streams do not depend on the actions: filter them up front, in the order the wrappers
            # see them (every rollout starts with a hard reset that is filtered -- and counted -- again, ppo.py:1580-1586)
            obs_table, rew_table = np.empty_like(raw_obs), np.empty_like(raw_rew)
            obs_table[0], _ = orc.filter_obs(raw_obs[0], raw_obs[0])
            for t in range(T):
                obs_table[t + 1], _, rew_table[t] = orc.filter_step(raw_obs[t + 1], raw_obs[t + 1], raw_rew[t], term[t],
                                                                    np.zeros(A * E, bool))
            np.testing.assert_allclose(obs_table[:T], agent_major(g["step_obs"][it * T:(it + 1) * T]), rtol=1e-6, atol=1e-6)
        acts = agent_major(g["step_raw_actions" if cpu.continuous else "step_actions"][it * T:(it + 1) * T])
        ds = cpu.rollout(obs_table, rew_table, actions=acts, term_table=term if term.any() else None,
                         max_ts_per_ep=c["max_ts_per_ep"], critic_obs_table=cobs_table, comm=comm)
        pre = f"it{it}_ds_"
        tol = dict(rtol=1e-6, atol=1e-6)
        if A == 1:                                                         # single agent: the order itself is the contract
            pi = np.arange(len(ds))
            np.testing.assert_allclose(ds.observations.numpy(), g[pre + "observations"], **tol)
            np.testing.assert_array_equal([e.length for e in ds.episodes], g[pre + "ep_lens"])
        else:
            pi = row_mapping(g[pre + "observations"], ds.observations.numpy())
            np.testing.assert_array_equal(np.sort([e.length for e in ds.episodes]), np.sort(g[pre + "ep_lens"]))
        np.testing.assert_allclose(ds.critic_observations.numpy()[pi], g[pre + "critic_observations"], **tol)
        np.testing.assert_allclose(ds.next_observations.numpy()[pi], g[pre + "next_observations"], **tol)
        if cpu.continuous:
            np.testing.assert_array_equal(ds.raw_actions.numpy()[pi], g[pre + "raw_actions"])
            np.testing.assert_allclose(ds.actions.numpy()[pi], g[pre + "actions"], **tol)     # tanh + per-dimension rescale
        else:
            np.testing.assert_array_equal(ds.actions.numpy()[pi], g[pre + "actions"])
        np.testing.assert_allclose(ds.values.numpy()[pi], g[pre + "values"], **tol)
        np.testing.assert_allclose(ds.log_probs.numpy()[pi], g[pre + "log_probs"], **tol)
        np.testing.assert_allclose(ds.rewards_to_go.numpy()[pi], g[pre + "rewards_to_go"], **tol)
        np.testing.assert_allclose(ds.advantages.numpy()[pi], g[pre + "advantages"], **tol)
        if cpu.enable_icm:
            k = list(g["rollout_status_keys"]).index("intrinsic score avg")
            np.testing.assert_allclose(cpu.intrinsic_score_avg, g["rollout_status"][it][k], rtol=1e-6)
        # the epoch loop with its KL early stop (ppo.py:2201-2232) is the oracle's own decision: it must run exactly the
        # epochs the reference ran (`epochs_run`; every epoch where no target_kl was set)
        target_kl = float(g["target_kl"][0]) if "target_kl" in g.files else 100.0
        want_epochs = int(g["epochs_run"][it]) if "epochs_run" in g.files else c["epochs"]
        first_ep, first_icm = ep, icm_ep
        cpu.trace = [] if ep == 0 else None

        def check_epoch(e, r):
            assert e < want_epochs, f"iteration {it}: the reference stopped after {want_epochs} epochs"
            if first_ep + e == 0:  # the very first mini-batch, before any optimiser step: losses and raw gradients
                m = cpu.trace[0]
                # (the actor loss is a mean of B O(1) surrogate terms that nearly cancel: float32 noise is ~1e-7 absolute)
                np.testing.assert_allclose([m["actor"], m["critic"]], g["mb0_losses"], rtol=1e-5, atol=1e-7)
                np.testing.assert_allclose(m["actor_grad"], g["mb0_actor_grad"], rtol=1e-5, atol=1e-7)
                np.testing.assert_allclose(m["critic_grad"], g["mb0_critic_grad"], rtol=1e-5, atol=1e-7)
                if comm is not None:   # what mpi_avg_gradients left in .grad (utils/mpi_utils.py:89-111): the ranks' mean
                    np.testing.assert_allclose(m["actor_avg_grad"], g["mb0_actor_avg_grad"], rtol=1e-5, atol=1e-7)
                    np.testing.assert_allclose(m["critic_avg_grad"], g["mb0_critic_avg_grad"], rtol=1e-5, atol=1e-7)
            cpu.trace = None
            if comm is not None:       # the value normaliser after this epoch: fed with the raw data of every rank (stats.py:47-50)
                vs_ = cpu.value_stats
                np.testing.assert_allclose([vs_.mean, vs_.variance, vs_.count], g["epoch_value_stats"][first_ep + e], rtol=2e-6, atol=1e-6)
            got = np.array([r["actor loss"], r["critic loss"], r["kl avg"], r["weighted entropy"]])
            # (the surrogate loss and the KL are means of O(1) terms that cancel: absolute floor 1e-6)
            np.testing.assert_allclose(got, g["epoch_stats"][first_ep + e], rtol=5e-6, atol=1e-6, err_msg=f"iteration {it} epoch {e}")
            if cpu.enable_icm:                                             # ppo.py:2213-2214: the ICM pass follows each PPO epoch
                np.testing.assert_allclose(r["icm loss"], g["icm_epoch_stats"][first_icm + e][0], rtol=5e-6, err_msg=f"icm loss {it}/{e}")

        pad = [None] * c["epochs"]             # (never reached: check_epoch stops a port that runs past the reference)
        ran = cpu_ppo_loop.train_on_rollout(
            cpu, c["epochs"], target_kl, on_epoch=check_epoch, comm=comm,
            perms=[pi[p] for p in g["epoch_perms"][ep:ep + want_epochs]] + pad,
            icm_perms=([pi[p] for p in g["icm_epoch_perms"][icm_ep:icm_ep + want_epochs]] + pad) if cpu.enable_icm else None)
        assert len(ran) == want_epochs, f"iteration {it}: ran {len(ran)} epochs, the reference {want_epochs}"
        ep += want_epochs
        icm_ep += want_epochs if cpu.enable_icm else 0
    for tag in ("actor", "critic") + (("icm",) if cpu.enable_icm else ()):
        got, want = final_params(g, tag, cpu)
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-7, err_msg=tag)
    vs = cpu.value_stats
    np.testing.assert_allclose([vs.mean, vs.variance, vs.count], g["value_stats"], rtol=2e-6, atol=1e-6)
    if filters is not None:                    # running statistics of the wrappers after the last rollout
        pre = "filter_ObservationNormalizer_actor_running_stats_agent0_"
        st = orc.obs_norm.stats[0]
        np.testing.assert_allclose(st.mean, g[pre + "mean"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(st.variance, g[pre + "var"], rtol=1e-6)
        np.testing.assert_allclose(st.count, g[pre + "count"][0], rtol=1e-9)
        pre = "filter_RewardNormalizer_running_stats_agent0_"
        st = orc.rew_norm.stats[0]
        np.testing.assert_allclose([st.mean, st.variance, st.count], [g[pre + "mean"], g[pre + "var"], g[pre + "count"][0]], rtol=1e-6)
        np.testing.assert_allclose(orc.rew_norm.running_reward[0], g["filter_RewardNormalizer_running_reward_agent0"], rtol=1e-6)


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_cpu_port_reproduces_the_reference_ppo_iterations(golden, name):
    replay_fixture(golden(name), name)


# R = 2 ranks of the reference under the two-process mpi4py stand-in (tests/golden/ref_import.py, make_golden_update.py:
# rank_scenarios): pins the DD-PPO arithmetic of the port -- mpi_avg_gradients (utils/mpi_utils.py:50-111), the value
# normaliser's all-gather of raw data (utils/stats.py:47-50), the all-reduced epoch totals (ppo.py:2468-2475) and the KL
# early stop they drive on both ranks (ppo.py:2221-2232)
RANK_SCENARIOS = {"g12_c2_r2": "g12_c2_b256", "g12_c4_r2": "g12_c4_b256", "g12_c2_icm_r2_klstop": "g12_c2_icm_klstop"}


@pytest.mark.parametrize("name", sorted(RANK_SCENARIOS))
def test_cpu_ddppo_port_reproduces_two_ranks_of_the_reference(golden, name):
    g = golden(name)
    R = int(g["ranks"][0])
    assert R == 2
    views = [RankView(g, r) for r in range(R)]
    # the fixture itself: the rank-0 broadcast made the initial weights identical, the ranks saw different data and
    # shuffles, and synchronous DD-PPO left them with identical weights and statistics
    for k in views[0].files:
        if k.startswith(("init_", "final_")) or k in ("epoch_stats", "epoch_value_stats", "value_stats", "mb0_actor_avg_grad", "mb0_critic_avg_grad"):
            np.testing.assert_array_equal(views[0][k], views[1][k], err_msg=k)
    assert not np.array_equal(views[0]["obs_table"], views[1]["obs_table"])
    assert not np.array_equal(views[0]["epoch_perms"], views[1]["epoch_perms"])
    np.testing.assert_array_equal(views[0]["mb0_actor_avg_grad"], (views[0]["mb0_actor_grad"] + views[1]["mb0_actor_grad"]) / 2)
    # both ranks of the port, each on its own thread, meeting in the reference's collectives
    cpu_ppo_loop.run_ranks([(lambda comm, v=v: replay_fixture(v, RANK_SCENARIOS[name], comm)) for v in views])


@pytest.mark.parametrize("name", sorted(SCENARIOS))
def test_rollout_statistics_oracle_reproduces_the_reference_status_block(golden, name):
    """ppo.py:1600-1631, 1756-2099: the literal statistics loop of the oracle on the per-step streams the reference
    saw, against the status_dict the reference published after each rollout."""
    g = golden(name)
    c = _cfg(g)
    E, T, A = c["E"], c["T"], c["A"]
    keys, rkeys, gkeys = (list(g[k]) for k in ("rollout_status_keys", "rollout_range_keys", "global_status_keys"))
    running_obs, top_nat, episodes = (np.inf, -np.inf), -np.inf, 0.0
    calls_step = g["values_calls_step"]
    for it in range(c["iterations"]):
        sl = slice(it * T, (it + 1) * T)
        term = g["term_table"]
        ep_ts, boot = np.zeros(E, np.int64), np.zeros((T, E), bool)
        for t in range(T):                                                  # ppo.py:1863-1877
            ep_ts += 1
            ep_ts[term[t]] = 0
            cut = (ep_ts >= c["max_ts_per_ep"]) | (t == T - 1)
            boot[t] = cut & ~term[t]
            ep_ts[cut] = 0
        # bootstrap rewards of the whole batch at every cut: the SECOND value call of a step (before the surprise)
        next_reward = np.zeros((T, E, A))
        seen = set()
        for i, st in enumerate(calls_step):
            if st in seen and it * T < st <= (it + 1) * T:
                next_reward[st - 1 - it * T] = g["values_calls"][i]
            seen.add(st)
        sq = (lambda x: x[..., 0]) if A == 1 else (lambda x: x)
        mm = g["step_next_obs_minmax"][sl]
        st = rollout_statistics_loop(sq(g["step_rewards"][sl]), sq(g["step_natural_rewards"][sl]), sq(g["step_intr_rewards"][sl]),
                                     mm[:, 0], mm[:, 1], term, boot, sq(next_reward), max_ts=c["max_ts_per_ep"])
        if "filters" not in SCENARIOS[name]:                                # ppo.py:2010-2017: a running range without normalize_obs
            running_obs = (min(running_obs[0], st["obs range"][0]), max(running_obs[1], st["obs range"][1]))
            st["obs range"] = running_obs
        top_nat = max(top_nat, st["top natural reward"]); st["top natural reward"] = top_nat
        episodes += st["total episodes"]
        for k in keys:
            np.testing.assert_allclose(st[k], g["rollout_status"][it][keys.index(k)], rtol=1e-6, atol=1e-9, err_msg=f"{k} it {it}")
        for k in rkeys:
            np.testing.assert_allclose(st[k], g["rollout_ranges"][it][rkeys.index(k)], rtol=1e-6, atol=1e-9, err_msg=f"{k} it {it}")
        gs = dict(zip(gkeys, g["global_status"][it]))
        np.testing.assert_allclose(episodes, gs["total episodes"], rtol=1e-9)
        for k in ("longest episode", "shortest episode", "average episode"):
            np.testing.assert_allclose(st[k], gs[k], rtol=1e-9, err_msg=f"{k} it {it}")
        assert gs["timesteps"] == (it + 1) * E * T


@pytest.mark.parametrize("name", ["g12_c5_mat", "g12_c5_b256"])
def test_cpu_mat_port_reproduces_the_reference_mat_iterations(golden, name):
    """MATPolicy (C5 shapes: 3 agents, O=18, Discrete(5), embedding 64, 1 block, 1 head) through the reference's own
    PPO object: autoregressive rollout log-probs, shared-episode dataset, teacher-forced evaluate, Huber value loss,
    one optimiser over actor + critic -- against oracle/mat_oracle.CpuMATPPO."""
    g = golden(name)
    c = _cfg(g)
    E, T, A, B = c["E"], c["T"], c["A"], c["batch_size"]
    cpu = mat_oracle.CpuMATPPO(c["O"], 5, A, batch_size=B, seed=0)
    sd = {"actor." + k[len("init_actor."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("init_actor.")}
    sd.update({"critic." + k[len("init_critic."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("init_critic.")})
    cpu.ac.load_state_dict(sd)
    ep = 0
    tol = dict(rtol=1e-6, atol=1e-6)
    for it in range(c["iterations"]):
        order = g["slot_orders"][it]                                        # slot j holds original agent order[j]
        sl = slice(it * T, (it + 1) * T)
        obs = g["obs_table"][:, :, order]
        # quirk Q14: the dataset keeps the agent order the policy had BEFORE this rollout's shuffle
        prev = np.array([int(a[len("agent"):]) for a in g["agent_ids"]]) if it == 0 else g["slot_orders"][it - 1]
        ds = cpu.rollout(obs, g["reward_table"][:, :, order], g["step_actions"][sl][:, :, order],
                         dataset_slot_of=np.argsort(order)[prev])
        pre = f"it{it}_ds_"
        np.testing.assert_array_equal(ds.obs.numpy(), g[pre + "observations"])          # rows [A, O] in slot order
        np.testing.assert_array_equal(ds.actions.numpy(), g[pre + "actions"])
        np.testing.assert_allclose(ds.values.numpy(), g[pre + "values"], **tol)
        np.testing.assert_allclose(ds.logp.numpy(), g[pre + "log_probs"], **tol)        # teacher-forced == autoregressive
        # the fixture was recorded under NumPy 2, where the reference's rewards-to-go scan accumulates in float32
        # (float64 under its pinned numpy < 1.24, which the oracle restates): 5e-6 apart after 512 steps
        long_tol = tol if T <= 64 else dict(rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(ds.rtg.numpy(), g[pre + "rewards_to_go"], **long_tol)
        np.testing.assert_allclose(ds.adv.numpy(), g[pre + "advantages"], **tol)
        for e in range(c["epochs"]):
            cpu.trace = [] if ep == 0 else None
            r = cpu.train_epoch(perm=g["epoch_perms"][ep])
            if ep == 0:
                m = cpu.trace[0]
                # (the actor loss is a mean of B O(1) surrogate terms that nearly cancel: float32 noise is ~1e-7 absolute)
                np.testing.assert_allclose([m["actor"], m["critic"]], g["mb0_losses"], rtol=1e-5, atol=1e-7)
                np.testing.assert_allclose(m["actor_grad"], g["mb0_actor_grad"], rtol=1e-5, atol=1e-7)
                np.testing.assert_allclose(m["critic_grad"], g["mb0_critic_grad"], rtol=1e-5, atol=1e-7)
            got = np.array([r["actor loss"], r["critic loss"], r["kl avg"], r["weighted entropy"]])
            np.testing.assert_allclose(got, g["epoch_stats"][ep], rtol=5e-6, atol=1e-6, err_msg=f"iteration {it} epoch {e}")
            ep += 1
    final = {"actor." + k[len("final_actor."):]: g[k] for k in g.files if k.startswith("final_actor.")}
    final.update({"critic." + k[len("final_critic."):]: g[k] for k in g.files if k.startswith("final_critic.")})
    for k, p in cpu.ac.named_parameters():
        np.testing.assert_allclose(p.detach().numpy(), final[k], rtol=1e-5, atol=1e-7, err_msg=k)
    vs = cpu.value_stats
    np.testing.assert_allclose([vs.mean, vs.variance, vs.count], g["value_stats"], rtol=2e-6, atol=1e-6)


@pytest.mark.parametrize("name,S,n_act", [("g12_lstm_term", 4, 2), ("g12_lstm_cut", 3, 3)])
def test_cpu_lstm_port_reproduces_the_reference_lstm_iterations(golden, name, S, n_act):
    """LSTMNetwork actor / critic through the reference's own PPO object: per-step hidden states logged (zeroed at
    terminations), sequence windows + terminal masks, hidden-state hand-over per mini-batch and write-back, the extra
    stateful critic step at episode cuts -- against oracle/lstm_oracle.CpuLSTMPPO."""
    g = golden(name)
    c = _cfg(g)
    E, T, B = c["E"], c["T"], c["batch_size"]
    cpu = lstm_oracle.CpuLSTMPPO(c["O"], n_act, sequence_length=S, lstm_hidden=32, ff_hidden=32, batch_size=B, seed=0,
                                 rtg_accum="float32")
    for net, tag in ((cpu.actor, "init_actor."), (cpu.critic, "init_critic.")):
        net.load_state_dict({k[len(tag):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag)})
    obs_table, rew_table, term = g["obs_table"][:, :, 0], g["reward_table"][:, :, 0], g["term_table"]
    tol = dict(rtol=1e-6, atol=1e-6)
    ep = 0
    for it in range(c["iterations"]):
        ds = cpu.rollout(obs_table, rew_table, g["step_actions"][it * T:(it + 1) * T, :, 0],
                         term_table=term if term.any() else None, max_ts_per_ep=c["max_ts_per_ep"])
        pre = f"it{it}_ds_"
        np.testing.assert_array_equal(ds.observations.numpy(), g[pre + "observations"])
        np.testing.assert_allclose(ds.values.numpy(), g[pre + "values"], **tol)
        np.testing.assert_allclose(ds.log_probs.numpy().reshape(-1), g[pre + "log_probs"].reshape(-1), **tol)
        np.testing.assert_allclose(ds.rewards_to_go.numpy(), g[pre + "rewards_to_go"], **tol)
        np.testing.assert_allclose(ds.advantages.numpy(), g[pre + "advantages"], **tol)
        for k in ("actor_hidden", "actor_cell", "critic_hidden", "critic_cell"):
            np.testing.assert_allclose(getattr(ds, k).numpy(), g[pre + k], err_msg=k, **tol)
        assert len(ds) == g["epoch_perms"].shape[1] == E * T - (S - 1)
        for e in range(c["epochs"]):
            # the recorded shuffles are the 13th tuple entries, i.e. sampler index + (S - 1) (episode_info.py:960-962)
            r = cpu.train_epoch(perm=g["epoch_perms"][ep] - (S - 1))
            got = np.array([r["actor loss"], r["critic loss"], r["kl avg"], r["weighted entropy"]])
            np.testing.assert_allclose(got, g["epoch_stats"][ep], rtol=5e-6, atol=1e-6, err_msg=f"iteration {it} epoch {e}")
            ep += 1
    for net, tag in ((cpu.actor, "final_actor."), (cpu.critic, "final_critic.")):
        for k, p in net.named_parameters():
            np.testing.assert_allclose(p.detach().numpy(), g[tag + k], rtol=1e-5, atol=1e-7, err_msg=tag + k)
    vs = cpu.value_stats
    np.testing.assert_allclose([vs.mean, vs.variance, vs.count], g["value_stats"], rtol=2e-6, atol=1e-6)


# ---------------------------------------------------------------------------------------- unit fixtures g8 / g9 / g10 / g13
@pytest.mark.parametrize("tag", ["c2", "c5", "c17"])
def test_g8_categorical_distribution(golden, tag):
    g = golden("g8_distributions")
    logits = torch.tensor(g[f"{tag}_logits"], requires_grad=True)
    actions = torch.tensor(g[f"{tag}_actions"])
    lp, ent, probs = plo.categorical_logp_entropy(logits, actions)
    np.testing.assert_allclose(probs.detach().numpy(), g[f"{tag}_probs"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(lp.detach().numpy(), g[f"{tag}_log_probs"].reshape(-1), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ent.detach().numpy(), g[f"{tag}_entropy"], rtol=1e-6, atol=1e-7)
    glp, = torch.autograd.grad(lp.sum(), logits, retain_graph=True)
    gent, = torch.autograd.grad(ent.sum(), logits)
    np.testing.assert_allclose(glp.numpy(), g[f"{tag}_dlogp_dlogits"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(gent.numpy(), g[f"{tag}_dent_dlogits"], rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(torch.argmax(probs, dim=-1).numpy(), g[f"{tag}_refined"])


@pytest.mark.parametrize("tag", ["unit", "bounds"])
def test_g8_gaussian_distribution(golden, tag):
    g = golden("g8_distributions")
    mean = torch.tensor(g[f"g_{tag}_mean"], requires_grad=True)
    log_std = torch.tensor(g[f"g_{tag}_log_std"], requires_grad=True)
    raw = torch.tensor(g[f"g_{tag}_raw"])
    np.testing.assert_allclose(plo.gaussian_dist(mean, log_std).stddev[0].detach().numpy(), g[f"g_{tag}_std"], rtol=1e-6)
    lp = plo.gaussian_tanh_logp(mean, log_std, raw)
    ent = -plo.gaussian_tanh_logp(mean, log_std, mean)                       # distributions.py:694 via ppo_policy.py:950
    np.testing.assert_allclose(lp.detach().numpy(), g[f"g_{tag}_log_probs"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ent.detach().numpy(), g[f"g_{tag}_entropy"], rtol=1e-6, atol=1e-6)
    gm, gs = torch.autograd.grad(lp.sum(), [mean, log_std], retain_graph=True)
    em, es = torch.autograd.grad(ent.sum(), [mean, log_std])
    for got, key in ((gm, "dlogp_dmean"), (gs, "dlogp_dlogstd"), (em, "dent_dmean"), (es, "dent_dlogstd")):
        np.testing.assert_allclose(got.numpy(), g[f"g_{tag}_{key}"], rtol=1e-5, atol=1e-6, err_msg=key)
    lo_, hi_ = g[f"g_{tag}_low"], g[f"g_{tag}_high"]
    np.testing.assert_allclose(plo.gaussian_refine(raw, lo_, hi_).numpy(), g[f"g_{tag}_refined_sample"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(plo.gaussian_refine(mean.detach(), lo_, hi_).numpy(), g[f"g_{tag}_refined_prediction"],
                               rtol=1e-6, atol=1e-7)


def test_g9_value_normalizer(golden):
    """RunningStatNormalizer (utils/misc.py:61-128) as the CPU port restates it (CpuPPO._norm_update / _denorm)."""
    g = golden("g9_value_normalizer")
    cpu = cpu_ppo_loop.CpuPPO(4, 2, seed=0)
    np.testing.assert_allclose(cpu._denorm(torch.tensor([0.5, -1.0, 3.0])).numpy(), g["denorm_fresh"], rtol=1e-6)
    for i in range(4):
        y = cpu._norm_update(torch.tensor(g[f"in{i}"]))
        np.testing.assert_allclose(y.numpy(), g[f"norm{i}"], rtol=1e-6, atol=1e-6)
        probe = torch.tensor(g[f"probe{i}"])
        np.testing.assert_allclose(cpu._denorm(probe).numpy(), g[f"denorm{i}"], rtol=1e-6, atol=1e-6)
        vs = cpu.value_stats
        mean, var = torch.tensor(vs.mean, dtype=torch.float32), torch.tensor(vs.variance, dtype=torch.float32)
        np.testing.assert_allclose(((probe - mean) / torch.sqrt(var + torch.tensor([1e-8]))).numpy(), g[f"norm_noupdate{i}"],
                                   rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose([vs.mean, vs.variance, vs.count], g[f"state{i}"], rtol=1e-7)


@pytest.mark.parametrize("tag", ["disc", "cont"])
def test_g10_icm_forward_and_gradients(golden, tag):
    g = golden("g10_icm")
    if tag == "disc":
        icm = icm_oracle.ICM(6, 3, discrete=True, enc=32, hidden=32)
    else:
        icm = icm_oracle.ICM(17, 6, discrete=False, enc=32, hidden=32, enc_hidden=64, inv_depth=3, fwd_depth=1)
    icm.load_state_dict({k[len(tag) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}_p_")})
    act = torch.tensor(g[f"{tag}_actions"])
    intr, inv_loss, f_loss = icm(torch.tensor(g[f"{tag}_obs1"]), torch.tensor(g[f"{tag}_obs2"]), act)
    loss = (1.0 - 0.8) * f_loss + 0.8 * inv_loss                              # ppo.py:2547-2548
    np.testing.assert_allclose(intr.detach().numpy(), g[f"{tag}_intr"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose([inv_loss.item(), f_loss.item(), loss.item()], g[f"{tag}_losses"], rtol=1e-6)
    names = [str(n) for n in g[f"{tag}_names"]]
    assert names == [k for k, _ in icm.named_parameters()]
    grads = torch.autograd.grad(loss, list(icm.parameters()))
    for k, gr in zip(names, grads):
        np.testing.assert_allclose(gr.numpy(), g[f"{tag}_g_{k}"], rtol=1e-5, atol=1e-8, err_msg=k)


def test_g13_filter_wrappers(golden):
    """The wrapper stack of wrapper_utils.py:81-111 stand-alone: 2 agents sharing a policy ("policy" critic view),
    terminations, two passes (the second starts with a hard reset that is filtered and counted again)."""
    g = golden("g13_filters")
    E, T, A, O = (int(x) for x in g["cfg"])
    orc = filter_oracle.FilteredEnvOracle(A, E, O, A * O, True, True, (-1.5, 1.5), (-1.0, 1.0), gamma=0.99)
    raw_obs, raw_cobs = agent_major(g["obs_table"]), policy_view(g["obs_table"])
    raw_rew, term = agent_major(g["reward_table"]), np.tile(g["term_table"], (1, A))
    tol = dict(rtol=1e-6, atol=1e-6)
    for p in range(2):
        o, c = orc.filter_obs(raw_obs[0], raw_cobs[0])
        np.testing.assert_allclose(o, agent_major(g[f"p{p}_obs"])[0], **tol)
        np.testing.assert_allclose(c, agent_major(g[f"p{p}_critic_obs"])[0], **tol)
        for t in range(T):
            # a finished env "resets" onto the observation it stopped on; the wrappers filter what reset() returned
            o, c, r = orc.filter_step(raw_obs[t + 1], raw_cobs[t + 1], raw_rew[t], term[t], np.zeros(A * E, bool))
            np.testing.assert_allclose(o, agent_major(g[f"p{p}_obs"])[t + 1], err_msg=f"obs pass {p} step {t}", **tol)
            np.testing.assert_allclose(c, agent_major(g[f"p{p}_critic_obs"])[t + 1], err_msg=f"critic obs pass {p} step {t}", **tol)
            np.testing.assert_allclose(r, agent_major(g[f"p{p}_rewards"])[t], err_msg=f"reward pass {p} step {t}", **tol)
        np.testing.assert_array_equal(agent_major(g[f"p{p}_natural"]), raw_rew.astype(np.float64))
    for a in range(A):
        for oracle_stats, key in ((orc.obs_norm.stats[a], f"ObservationNormalizer_actor_running_stats_agent{a}_"),
                                  (orc.cobs_norm.stats[a], f"ObservationNormalizer_critic_running_stats_agent{a}_"),
                                  (orc.rew_norm.stats[a], f"RewardNormalizer_running_stats_agent{a}_")):
            np.testing.assert_allclose(oracle_stats.mean, g[key + "mean"], rtol=1e-6, atol=1e-7, err_msg=key)
            np.testing.assert_allclose(oracle_stats.variance, g[key + "var"], rtol=1e-6, err_msg=key)
            np.testing.assert_allclose(oracle_stats.count, g[key + "count"][0], rtol=1e-9, err_msg=key)
