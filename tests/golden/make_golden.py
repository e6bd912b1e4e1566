#!/usr/bin/env python3
"""
Golden-vector generator (runs ONLY in the build container, never on the GPU box).

It imports the *unmodified* reference modules from /root/reference and records
input/output vectors for the hot path as small .npz fixtures next to this file.
Nothing from the reference is copied: the fixtures are data (inputs + outputs).

Import recipe: a scratch directory on sys.path holding
  (i)  a symlink  ppo_and_friends -> /root/reference   (mirrors setup.py:6-13)
  (ii) a single-rank stand-in for `mpi4py` (rank 0 of 1; collectives = identity)
The scratch directory lives under /tmp and is removed afterwards.

Importable with that recipe: utils.episode_info, utils.stats, utils.mpi_utils,
networks.attention, networks.utils, utils.schedulers (fixtures g1-g7).  Everything routed through `gymnasium`
(distributions, policies, ppo.py) is NOT importable here and is restated from
text in oracle/ (pinned by torch primitives; see DESIGN.md "Oracle").

Usage:  python tests/golden/make_golden.py
"""
import os
import shutil
import sys
import tempfile

sys.dont_write_bytecode = True

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"


def _import_reference():
    scratch = tempfile.mkdtemp(prefix="ppoaf_golden_")
    os.symlink(REFERENCE, os.path.join(scratch, "ppo_and_friends"))
    mp = os.path.join(scratch, "mpi4py")
    os.makedirs(mp)
    with open(os.path.join(mp, "__init__.py"), "w") as fh:
        fh.write(
            "class _Comm:\n"
            "    def Get_rank(self): return 0\n"
            "    def Get_size(self): return 1\n"
            "    def allreduce(self, x, op=None): return x\n"
            "    def allgather(self, x): return [x]\n"
            "    def Bcast(self, buf, root=0): return None\n"
            "    def barrier(self): return None\n"
            "    def Abort(self, code=1): raise RuntimeError('MPI Abort')\n"
            "class MPI:\n"
            "    COMM_WORLD = _Comm()\n"
            "    SUM = 'sum'; MAX = 'max'; MIN = 'min'\n")
    sys.path.insert(0, scratch)
    from ppo_and_friends.utils import episode_info as ei     # noqa
    from ppo_and_friends.utils import stats as st            # noqa
    from ppo_and_friends.networks import attention as at     # noqa
    return scratch, ei, st, at


def _drive_episode(ei, rewards, values, obs, next_obs, actions, log_probs,
                   critic_obs, starting_ts, terminal, ending_value,
                   ending_reward, use_gae, gamma, lambd, bootstrap_clip):
    """Drive one EpisodeInfo exactly as policies/ppo_policy.py:638-651,684-691 do."""
    ep = ei.EpisodeInfo(starting_ts=starting_ts, use_gae=use_gae, gamma=gamma,
                        lambd=lambd, bootstrap_clip=bootstrap_clip)
    L = len(rewards)
    for t in range(L):
        ep.add_info(observation=obs[t], next_observation=next_obs[t],
                    raw_action=actions[t], action=actions[t],
                    value=values[t].item(), log_prob=log_probs[t],
                    reward=rewards[t].item(), critic_observation=critic_obs[t])
    ep.end_episode(ending_ts=starting_ts + L, terminal=terminal,
                   ending_value=float(ending_value),
                   ending_reward=float(ending_reward))
    return ep


def gen_g1(ei, out):
    """G1: EpisodeInfo.end_episode -> advantages / rewards_to_go (episode_info.py:419-465)."""
    rng = np.random.default_rng(20241022)
    cases = []
    case_id = 0
    for T in (1, 2, 32, 128):
        for (gamma, lambd) in ((0.99, 0.95), (1.0, 1.0), (0.9, 0.0)):
            for use_gae in (True, False):
                for clip in ((-100.0, 100.0), (-0.5, 0.5), None):
                    for ending in ("terminal", "bootstrap"):
                        for rew_kind in ("ones", "uniform"):
                            if rew_kind == "ones":
                                rewards = np.ones(T, dtype=np.float64)
                            else:
                                rewards = rng.uniform(-1, 1, T)
                            values = rng.standard_normal(T).astype(np.float32)
                            if ending == "terminal":
                                ev, er = 0.0, 0.0
                            else:
                                ev = float(rng.standard_normal() * 2.0)
                                er = float(rng.standard_normal() * 2.0)
                            obs = rng.standard_normal((T, 4)).astype(np.float32)
                            lp = [torch.tensor([0.0])] * T
                            act = np.zeros((T, 1), dtype=np.int64)
                            ep = _drive_episode(
                                ei, rewards, values, obs, obs, act, lp, obs,
                                0, ending == "terminal", ev, er, use_gae,
                                gamma, lambd, clip)
                            # fp64-accumulated rtg: what the reference's pinned
                            # numpy<1.24 scalar promotion yields (episode_info.py:254-262).
                            er_c = er if clip is None else float(np.clip(er, clip[0], clip[1]))
                            padded = np.array(list(rewards) + [er_c], dtype=np.float32)
                            rtg64 = ep.compute_discounted_sums(
                                padded.astype(np.float64), gamma)[:-1]
                            pre = f"c{case_id}_"
                            out[pre + "rewards"] = rewards
                            out[pre + "values"] = values
                            out[pre + "params"] = np.array(
                                [gamma, lambd, float(use_gae),
                                 np.nan if clip is None else clip[0],
                                 np.nan if clip is None else clip[1],
                                 ev, er, float(ending == "terminal")], dtype=np.float64)
                            out[pre + "adv"] = np.asarray(ep.advantages, dtype=np.float64)
                            out[pre + "rtg_np2"] = np.asarray(ep.rewards_to_go, dtype=np.float64)
                            out[pre + "rtg_f64"] = np.asarray(rtg64, dtype=np.float64)
                            cases.append(case_id)
                            case_id += 1
    out["n_cases"] = np.array([case_id])


def _mixed_rollout(ei, rng, E, T, O, term_p, use_gae=True, gamma=0.99,
                   lambd=0.95, clip=(-100.0, 100.0), max_ts_per_ep=None):
    """
    Drive E envs for T steps the way PPO.rollout does (ppo.py:1646-1983):
    per step add_info for every env; terminal envs end with (0, 0); at rollout
    end (or max_ts_per_ep) every open episode ends with a critic bootstrap.
    Returns the dataset plus the dense [T,E] arrays the build's buffer uses.
    """
    ds = ei.PPODataset(device=torch.device("cpu"), action_dtype="discrete",
                       sequence_length=1)
    obs = rng.standard_normal((T + 1, E, O)).astype(np.float32)
    rewards = rng.uniform(-1, 1, (T, E))
    values = rng.standard_normal((T, E)).astype(np.float32)
    boot_v = (rng.standard_normal((T, E)) * 1.5).astype(np.float32)
    logp = (-np.abs(rng.standard_normal((T, E)))).astype(np.float32)
    actions = rng.integers(0, 2, (T, E)).astype(np.int64)
    term = rng.uniform(0, 1, (T, E)) < term_p
    episodes = [ei.EpisodeInfo(0, use_gae, gamma, lambd, clip) for _ in range(E)]
    ep_len = np.zeros(E, dtype=np.int64)
    ep_ts = np.zeros(E, dtype=np.int64)
    end_kind = np.zeros((T, E), dtype=np.int8)   # 0 none, 1 terminal, 2 bootstrapped
    order = []
    for t in range(T):
        ep_len += 1
        ep_ts += 1
        for e in range(E):
            episodes[e].add_info(
                observation=obs[t, e], next_observation=obs[t + 1, e],
                raw_action=actions[t, e:e + 1], action=actions[t, e:e + 1],
                value=values[t, e].item(),
                log_prob=torch.tensor([logp[t, e]]),
                reward=rewards[t, e].item(), critic_observation=obs[t, e])
        where_term = np.where(term[t])[0]
        for e in where_term:
            episodes[e].end_episode(ending_ts=ep_len[e], terminal=True,
                                    ending_value=0.0, ending_reward=0.0)
            ds.add_episode(episodes[e])
            order.append((t, e))
            end_kind[t, e] = 1
            episodes[e] = ei.EpisodeInfo(0, use_gae, gamma, lambd, clip)
            ep_len[e] = 0
            ep_ts[e] = 0
        if t == T - 1:
            where_maxed = np.arange(E)
        elif max_ts_per_ep is not None:
            where_maxed = np.where(ep_ts >= max_ts_per_ep)[0]
        else:
            where_maxed = np.array([], dtype=np.int64)
        where_maxed = np.setdiff1d(where_maxed, where_term)
        for e in where_maxed:
            episodes[e].end_episode(ending_ts=ep_len[e], terminal=False,
                                    ending_value=boot_v[t, e].item(),
                                    ending_reward=boot_v[t, e].item())
            ds.add_episode(episodes[e])
            order.append((t, e))
            end_kind[t, e] = 2
            episodes[e] = ei.EpisodeInfo(ep_len[e], use_gae, gamma, lambd, clip)
            ep_ts[e] = 0
    ds.build()
    dense = dict(obs=obs, rewards=rewards, values=values, boot_v=boot_v,
                 logp=logp, actions=actions, end_kind=end_kind)
    return ds, dense


def gen_g2(ei, out):
    """G2: PPODataset.build / __getitem__ / recalculate_advantages (episode_info.py:721-987)."""
    rng = np.random.default_rng(7)
    for tag, (E, T, term_p, max_ts) in {
            "a": (3, 12, 0.15, None),
            "b": (8, 32, 0.05, 8),
            "c": (5, 16, 0.0, None)}.items():
        ds, dense = _mixed_rollout(ei, rng, E, T, 4, term_p, max_ts_per_ep=max_ts)
        pre = f"{tag}_"
        for k, v in dense.items():
            out[pre + "in_" + k] = v
        out[pre + "cfg"] = np.array([E, T, -1 if max_ts is None else max_ts])
        out[pre + "obs"] = ds.observations.numpy()
        out[pre + "next_obs"] = ds.next_observations.numpy()
        out[pre + "critic_obs"] = ds.critic_observations.numpy()
        out[pre + "actions"] = ds.actions.numpy()
        out[pre + "raw_actions"] = ds.raw_actions.numpy()
        out[pre + "adv"] = ds.advantages.numpy().copy()
        out[pre + "logp"] = ds.log_probs.numpy()
        out[pre + "rtg"] = ds.rewards_to_go.numpy()
        out[pre + "values"] = ds.values.numpy().copy()   # copy: overwritten below
        out[pre + "ep_lens"] = ds.ep_lens
        out[pre + "len"] = np.array([len(ds)])
        item = ds[len(ds) // 2]
        out[pre + "item_idx"] = np.array([item[12]])
        out[pre + "item_obs"] = item[1].numpy()
        out[pre + "item_adv"] = np.array([item[5].item()])
        # recalculate_advantages after the trainer overwrites dataset.values
        # (ppo.py:2340 then ppo.py:2207-2208).
        new_vals = torch.tensor(
            rng.standard_normal(len(ds)).astype(np.float32))
        ds.values[:] = new_vals
        ds.recalculate_advantages()
        out[pre + "new_values"] = new_vals.numpy()
        out[pre + "adv_recalc"] = ds.advantages.numpy()


def gen_g3(ei, out):
    """G3: AgentSharedEpisode / PPOSharedEpisodeDataset stacking, A=3 (episode_info.py:485-644,990-1084)."""
    rng = np.random.default_rng(11)
    agent_ids = np.array(["a0", "a1", "a2"])
    E, T, O = 2, 6, 5
    ds = ei.PPOSharedEpisodeDataset(
        num_envs=E, agent_ids=agent_ids, device=torch.device("cpu"),
        action_dtype="discrete", sequence_length=1)
    obs = rng.standard_normal((T + 1, E, 3, O)).astype(np.float32)
    rewards = rng.uniform(-1, 1, (T, E, 3))
    values = rng.standard_normal((T, E, 3)).astype(np.float32)
    logp = (-np.abs(rng.standard_normal((T, E, 3)))).astype(np.float32)
    actions = rng.integers(0, 5, (T, E, 3)).astype(np.int64)
    boot = rng.standard_normal((E, 3)).astype(np.float32)
    for e in range(E):
        for a, aid in enumerate(agent_ids):
            ep = ei.EpisodeInfo(0, True, 0.99, 0.95, (-100.0, 100.0))
            for t in range(T):
                ep.add_info(observation=obs[t, e, a], next_observation=obs[t + 1, e, a],
                            raw_action=actions[t, e, a:a + 1], action=actions[t, e, a:a + 1],
                            value=values[t, e, a].item(),
                            log_prob=torch.tensor([logp[t, e, a]]),
                            reward=rewards[t, e, a].item(),
                            critic_observation=obs[t, e, a])
            # ending_ts arrives as a numpy int in the reference (ppo.py:1631,
            # ppo_policy.py:686); AgentSharedEpisode._merge_episodes relies on it.
            ep.end_episode(ending_ts=np.int32(T), terminal=False,
                           ending_value=boot[e, a].item(),
                           ending_reward=boot[e, a].item())
            ds.add_shared_episode(ep, aid, e)
    ds.build()
    out["in_obs"] = obs
    out["in_rewards"] = rewards
    out["in_values"] = values
    out["in_logp"] = logp
    out["in_actions"] = actions
    out["in_boot"] = boot
    out["obs"] = ds.observations.numpy()
    out["actions"] = ds.actions.numpy()
    out["adv"] = ds.advantages.numpy()
    out["rtg"] = ds.rewards_to_go.numpy()
    out["values"] = ds.values.numpy()
    out["logp"] = ds.log_probs.numpy()
    out["len"] = np.array([len(ds)])


def gen_g4(st, out):
    """G4: RunningMeanStd.update sequences (utils/stats.py:9-94)."""
    rng = np.random.default_rng(5)
    # scalar-shaped stats (value normaliser: misc.py:83)
    rs = st.RunningMeanStd()
    batches = [rng.standard_normal(n).astype(np.float32) * s + m
               for (n, s, m) in ((256, 1.0, 0.0), (256, 3.0, 10.0), (17, 0.1, -4.0), (1, 1.0, 2.0))]
    for i, b in enumerate(batches):
        rs.update(b)
        out[f"s_batch{i}"] = b
        out[f"s_state{i}"] = np.array([rs.mean, rs.variance, rs.count], dtype=np.float64)
        out[f"s_mean_dtype{i}"] = np.array([str(np.asarray(rs.mean).dtype)])
    # vector-shaped stats (obs normaliser: filter_wrappers.py:155-258)
    rv = st.RunningMeanStd(shape=(6,))
    for i in range(3):
        b = (rng.standard_normal((32, 6)) * (i + 1) + i).astype(np.float32)
        rv.update(b)
        out[f"v_batch{i}"] = b
        out[f"v_mean{i}"] = np.asarray(rv.mean, dtype=np.float64)
        out[f"v_var{i}"] = np.asarray(rv.variance, dtype=np.float64)
        out[f"v_count{i}"] = np.array([rv.count], dtype=np.float64)


def gen_g5(at, out):
    """G5: SelfAttention + encoder/decoder blocks, B=4 L=3 D=64 (networks/attention.py:13-257)."""
    torch.manual_seed(1234)
    B, L, D = 4, 3, 64
    x = torch.randn(B, L, D)
    rep = torch.randn(B, L, D)
    for masked in (False, True):
        sa = at.SelfAttention(D, 1, L, internal_init=1.0, out_init=1.0, masked=masked)
        tag = "m" if masked else "u"
        for n, p in sa.named_parameters():
            out[f"sa_{tag}_{n}"] = p.detach().numpy()
        out[f"sa_{tag}_y"] = sa(x, x, x).detach().numpy()
    enc = at.SelfAttentionEncodingBlock(D, 1, L, self_atten_internal_init=1.0,
                                        self_atten_out_init=1.0, out_init=1.0)
    for n, p in enc.named_parameters():
        out[f"enc_{n}"] = p.detach().numpy()
    out["enc_y"] = enc(x).detach().numpy()
    dec = at.SelfAttentionDecodingBlock(D, 1, L, self_atten_internal_init=1.0,
                                        self_atten_out_init=1.0, out_init=1.0)
    for n, p in dec.named_parameters():
        out[f"dec_{n}"] = p.detach().numpy()
    out["dec_y"] = dec(x, rep).detach().numpy()
    out["x"] = x.numpy()
    out["rep"] = rep.numpy()


def gen_g6(nu, out):
    """
    G6: networks/utils.py -- create_sequential_network (:120-191: module tree, activation placement,
    out_init) and init_layer / init_net_parameters (:53-111: orthogonal weights with a gain, constant bias):
    parameter names + values and the forward output on a fixed input for the layer shapes of the configs.
    """
    import torch.nn as nn
    torch.manual_seed(4321)
    cases = [("c2_actor", 4, 2, 128, 3, 0.01, nn.ReLU()), ("c2_critic", 4, 1, 128, 3, 1.0, nn.ReLU()),
             ("leaky", 6, 3, 32, 2, None, nn.LeakyReLU()), ("tanh_d1", 5, 4, 16, 1, 0.5, nn.Tanh()),
             ("no_hidden", 7, 3, 0, 0, 1.0, nn.ReLU()), ("list_sizes", 9, 2, [24, 16, 8], 99, None, nn.ReLU())]
    out["n_cases"] = np.array([len(cases)])
    for i, (tag, n_in, n_out, hs, hd, out_init, act) in enumerate(cases):
        net = nu.create_sequential_network(in_size=n_in, out_size=n_out, hidden_size=hs, hidden_depth=hd,
                                           activation=act, out_init=out_init)
        x = torch.randn(11, n_in)
        names = []
        for n, p in net.named_parameters():
            out[f"c{i}_p_{n}"] = p.detach().numpy()
            names.append(n)
        out[f"c{i}_names"] = np.array(names)
        out[f"c{i}_x"] = x.numpy()
        out[f"c{i}_y"] = net(x).detach().numpy()
        out[f"c{i}_cfg"] = np.array([tag, str(n_in), str(n_out), repr(hs), str(hd), repr(out_init), type(act).__name__])
    lstm = nu.init_net_parameters(nn.LSTM(5, 8, 1))
    for n, p in lstm.named_parameters():
        out[f"lstm_{n}"] = p.detach().numpy()
    lin = nu.init_layer(nn.Linear(6, 6), gain=0.3, bias_const=0.25)
    out["lin_w"], out["lin_b"] = lin.weight.detach().numpy(), lin.bias.detach().numpy()


def scheduler_scenario():
    """The status trajectory both the reference and the product schedulers are driven through (pure data)."""
    rng = np.random.default_rng(77)
    n = 40
    timesteps = np.cumsum(rng.integers(500, 4000, n))
    scores = np.round(np.cumsum(rng.normal(3.0, 10.0, n)), 3)
    plateau = np.repeat(rng.integers(0, 5, n // 4), 4)           # a key that stays constant for stretches
    return dict(iteration=np.arange(n), timesteps=timesteps, score=scores, plateau=plateau)


def drive_schedulers(mod, sc):
    """Every schedule of utils/schedulers.py over the scenario -> dict of output arrays (also used by the test)."""
    n = len(sc["iteration"])
    status = {"global status": {"iteration": 0, "timesteps": 0}, "p": {"score avg": 0.0, "plateau": 0}}
    made = {
        "linear_ts": mod.LinearScheduler("timesteps", status_max=60000, max_value=3e-4, min_value=1e-5),
        "linear_it": mod.LinearScheduler("iteration", status_max=25, max_value=0.05, min_value=0.0),
        "log_it": mod.LogScheduler("iteration", status_max=30, max_value=1.0, min_value=0.1),
        "step_gt": mod.LinearStepScheduler(initial_value=1e-3, status_key="score avg", status_preface="p",
                                           status_triggers=[10.0, 40.0, 90.0], step_values=[5e-4, 1e-4, 1e-5]),
        "step_lt": mod.LinearStepScheduler(initial_value=0.0, status_key="score avg", status_preface="p",
                                           status_triggers=[-5.0, -20.0], step_values=[1.0, 2.0], compare_fn=np.less),
        "change": mod.ChangeInStateScheduler("plateau", status_preface="p"),
        "change_persistent": mod.ChangeInStateScheduler("score avg", status_preface="p", compare_fn=np.greater,
                                                        persistent=True),
        "const": mod.CallableValue(0.25),
    }
    for s in made.values():
        s.finalize(status)

    class FakePolicy:
        def __init__(self):
            self.frozen, self.saved = False, []
        def freeze(self): self.frozen = True
        def unfreeze(self): self.frozen = False
        def save(self, path, tag): self.saved.append(int(tag))

    policies = {k: FakePolicy() for k in ("a", "b", "c", "d")}
    cyc = mod.FreezeCyclingScheduler([["a", "b"], ["c"]], iterations=3, delay=4)
    cyc.finalize("/nonexistent", status, policies)
    out = {k: [] for k in made}
    out["frozen"], out["active_idx"] = [], []
    for i in range(n):
        status["global status"]["iteration"] = int(sc["iteration"][i])
        status["global status"]["timesteps"] = int(sc["timesteps"][i])
        status["p"]["score avg"] = float(sc["score"][i])
        status["p"]["plateau"] = int(sc["plateau"][i])
        cyc()
        for k, s in made.items():
            out[k].append(float(s()))
        out["frozen"].append([int(policies[k].frozen) for k in ("a", "b", "c", "d")])
        out["active_idx"].append(cyc.active_idx)
    res = {k: np.asarray(v) for k, v in out.items()}
    res["groups"] = np.array([",".join(g) for g in cyc.policy_groups])
    for k in ("a", "b", "c", "d"):
        res[f"saved_{k}"] = np.asarray(policies[k].saved, dtype=np.int64)
    return res


def gen_g7(sched, out):
    """G7: utils/schedulers.py -- every schedule and the freeze cycle over a 40-iteration status trajectory."""
    sc = scheduler_scenario()
    for k, v in sc.items():
        out["in_" + k] = v
    out.update(drive_schedulers(sched, sc))


def main():
    only = set(sys.argv[1:])                      # e.g. `make_golden.py g6_network_utils`: just that fixture
    scratch, ei, st, at = _import_reference()
    from ppo_and_friends.networks import utils as nu
    from ppo_and_friends.utils import schedulers as sched
    try:
        for name, fn, mod in (("g1_end_episode", gen_g1, ei),
                              ("g2_dataset", gen_g2, ei),
                              ("g3_shared", gen_g3, ei),
                              ("g4_running_stats", gen_g4, st),
                              ("g5_attention", gen_g5, at),
                              ("g6_network_utils", gen_g6, nu),
                              ("g7_schedulers", gen_g7, sched)):
            if only and name not in only:
                continue
            out = {}
            fn(mod, out)
            path = os.path.join(HERE, name + ".npz")
            np.savez_compressed(path, **out)
            print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} B")
    finally:
        sys.path.remove(scratch)
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
