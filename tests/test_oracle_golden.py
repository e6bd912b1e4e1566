"""
CPU tests: the oracle (oracle/*.py) against golden vectors recorded from the
unmodified reference (tests/golden/make_golden.py).  This is what "pins" the
oracle; the -m gpu tests then compare the HIP path with the oracle.
"""
import numpy as np
import pytest

from oracle import episode_info_oracle as eo
from oracle import running_stats_oracle as rso


def _g1_cases(g):
    n = int(g["n_cases"][0])
    for c in range(n):
        p = g[f"c{c}_params"]
        clip = None if np.isnan(p[3]) else (p[3], p[4])
        yield c, dict(rewards=g[f"c{c}_rewards"], values=g[f"c{c}_values"],
                      gamma=p[0], lambd=p[1], use_gae=bool(p[2]), clip=clip,
                      ev=p[5], er=p[6]), g[f"c{c}_adv"], g[f"c{c}_rtg_np2"], g[f"c{c}_rtg_f64"]


def test_g1_end_episode_exact(golden):
    """end_episode: advantages bit-exact in f64; rtg bit-exact in both accumulator modes."""
    g = golden("g1_end_episode")
    n = 0
    for c, k, adv, rtg_np2, rtg_f64 in _g1_cases(g):
        a64, r64 = eo.end_episode(k["rewards"], k["values"], k["ev"], k["er"],
                                  k["gamma"], k["lambd"], k["clip"], k["use_gae"], "float64")
        a32, r32 = eo.end_episode(k["rewards"], k["values"], k["ev"], k["er"],
                                  k["gamma"], k["lambd"], k["clip"], k["use_gae"], "float32")
        np.testing.assert_array_equal(r64, rtg_f64, err_msg=f"case {c} rtg f64")
        np.testing.assert_array_equal(r32, rtg_np2, err_msg=f"case {c} rtg np2/f32")
        if k["use_gae"]:
            np.testing.assert_array_equal(a64, adv, err_msg=f"case {c} adv")
        else:
            # non-GAE advantages inherit the rtg accumulator of the run that
            # recorded them (NumPy 2 here -> float32 accumulation).
            np.testing.assert_array_equal(a32, adv, err_msg=f"case {c} adv (rtg - V)")
        n += 1
    assert n == 288


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g2_dataset_order_and_values(golden, tag):
    """Dense [T,E] rollout -> flattened dataset: ordering contract + adv/rtg/values."""
    g = golden("g2_dataset")
    pre = tag + "_"
    rewards = g[pre + "in_rewards"]
    values = g[pre + "in_values"]
    boot = g[pre + "in_boot_v"]
    end_kind = g[pre + "in_end_kind"]
    extra = dict(obs=g[pre + "in_obs"][:-1], next_obs=g[pre + "in_obs"][1:],
                 logp=g[pre + "in_logp"], actions=g[pre + "in_actions"])
    d = eo.rollout_to_dataset(rewards, values, boot, boot, end_kind,
                              rtg_accum="float32", extra=extra)
    assert len(d["adv"]) == int(g[pre + "len"][0])
    np.testing.assert_array_equal(d["ep_lens"], g[pre + "ep_lens"])
    np.testing.assert_array_equal(d["obs"], g[pre + "obs"])
    np.testing.assert_array_equal(d["next_obs"], g[pre + "next_obs"])
    np.testing.assert_array_equal(d["actions"][:, None], g[pre + "actions"])
    np.testing.assert_array_equal(d["logp"], g[pre + "logp"])
    np.testing.assert_array_equal(d["values"], g[pre + "values"])
    np.testing.assert_array_equal(d["adv"], g[pre + "adv"])
    np.testing.assert_array_equal(d["rtg"], g[pre + "rtg"])
    # fp64-accumulated rtg (the pinned-numpy semantic, the parity target) stays
    # within float32 rounding of the NumPy-2 run.
    d64 = eo.rollout_to_dataset(rewards, values, boot, boot, end_kind, rtg_accum="float64")
    np.testing.assert_allclose(d64["rtg"], g[pre + "rtg"], rtol=2e-6, atol=2e-6)
    # __getitem__ contract: index -> row of the flattened tensors
    idx = int(g[pre + "item_idx"][0])
    np.testing.assert_array_equal(d["obs"][idx], g[pre + "item_obs"])
    assert d["adv"][idx] == g[pre + "item_adv"][0]


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g2_recalculate_advantages(golden, tag):
    g = golden("g2_dataset")
    pre = tag + "_"
    rewards = g[pre + "in_rewards"]
    boot = g[pre + "in_boot_v"]
    end_kind = g[pre + "in_end_kind"]
    d = eo.rollout_to_dataset(rewards, g[pre + "in_values"], boot, boot, end_kind,
                              extra=dict(rewards=rewards))
    ending_values = [0.0 if kind == 1 else float(boot[t1, e]) for (e, t0, t1, kind) in d["segs"]]
    adv = eo.recalculate_advantages(d["rewards"], g[pre + "new_values"], d["ep_lens"],
                                    ending_values)
    np.testing.assert_array_equal(adv, g[pre + "adv_recalc"])


def test_g3_shared_episode_layout(golden):
    """AgentSharedEpisode stacking: [N, A, ...], episodes env-major, agents as columns."""
    g = golden("g3_shared")
    obs, rewards, values = g["in_obs"], g["in_rewards"], g["in_values"]
    boot, logp, actions = g["in_boot"], g["in_logp"], g["in_actions"]
    T, E, A = rewards.shape
    adv = np.zeros((E * T, A), dtype=np.float32)
    rtg = np.zeros((E * T, A), dtype=np.float32)
    for e in range(E):
        for a in range(A):
            ad, rg = eo.end_episode(rewards[:, e, a], values[:, e, a], boot[e, a], boot[e, a],
                                    0.99, 0.95, (-100.0, 100.0), True, "float32")
            adv[e * T:(e + 1) * T, a] = ad
            rtg[e * T:(e + 1) * T, a] = rg
    np.testing.assert_array_equal(adv, g["adv"])
    np.testing.assert_array_equal(rtg, g["rtg"])
    exp_obs = np.concatenate([obs[:-1, e] for e in range(E)], axis=0)        # [E*T, A, O]
    np.testing.assert_array_equal(exp_obs, g["obs"])
    exp_vals = np.concatenate([values[:, e] for e in range(E)], axis=0)
    np.testing.assert_array_equal(exp_vals, g["values"])
    exp_lp = np.concatenate([logp[:, e] for e in range(E)], axis=0)
    np.testing.assert_array_equal(exp_lp, g["logp"])
    assert g["actions"].shape[:2] == (E * T, A)
    assert int(g["len"][0]) == E * T


def test_g4_running_mean_std(golden):
    g = golden("g4_running_stats")
    rs = rso.RunningMeanStd()
    for i in range(4):
        rs.update(g[f"s_batch{i}"])
        exp = g[f"s_state{i}"]
        assert np.float64(rs.mean) == exp[0]
        assert np.float64(rs.variance) == exp[1]
        assert rs.count == exp[2]
    rv = rso.RunningMeanStd(shape=(6,))
    for i in range(3):
        rv.update(g[f"v_batch{i}"])
        np.testing.assert_array_equal(np.asarray(rv.mean, dtype=np.float64), g[f"v_mean{i}"])
        np.testing.assert_array_equal(np.asarray(rv.variance, dtype=np.float64), g[f"v_var{i}"])
        assert rv.count == g[f"v_count{i}"][0]


def test_c_oracle_matches_golden_g1(golden):
    """oracle/gae_oracle.c (plain C) bit-exact against the reference's end_episode outputs."""
    from oracle import c_oracle
    g = golden("g1_end_episode")
    for c, k, adv, rtg_np2, rtg_f64 in _g1_cases(g):
        a, r = c_oracle.gae_rtg_episode(k["rewards"], k["values"], k["ev"], k["er"], k["gamma"],
                                        k["lambd"], k["clip"], k["use_gae"])
        np.testing.assert_array_equal(r, rtg_f64, err_msg=f"case {c} rtg")
        if k["use_gae"]:
            np.testing.assert_array_equal(a, adv, err_msg=f"case {c} adv")


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_c_oracle_tmajor_matches_numpy_oracle(golden, tag):
    from oracle import c_oracle
    g = golden("g2_dataset")
    pre = tag + "_"
    rew = g[pre + "in_rewards"].astype(np.float32)
    val, boot, ek = g[pre + "in_values"], g[pre + "in_boot_v"], g[pre + "in_end_kind"]
    adv, rtg = c_oracle.gae_rtg_tmajor(rew, val, boot, boot, ek)
    d = eo.rollout_to_dataset(rew, val, boot, boot, ek)
    np.testing.assert_array_equal(adv[d["flat_t"], d["flat_e"]], d["adv"])
    np.testing.assert_array_equal(rtg[d["flat_t"], d["flat_e"]], d["rtg"])


def test_attention_oracle_matches_golden_g5(golden):
    """oracle/mat_oracle.py attention blocks, loaded with the reference's seeded weights -> its outputs."""
    import torch
    from oracle import mat_oracle as mo
    g = golden("g5_attention")
    x, rep = torch.tensor(g["x"]), torch.tensor(g["rep"])

    def load(mod, prefix):
        mod.load_state_dict({k: torch.tensor(g[prefix + k]) for k in mod.state_dict()
                             if not k.endswith("mask")}, strict=False)
        return mod

    for tag, masked in (("u", False), ("m", True)):
        sa = load(mo.SelfAttention(64, 1, 3, masked=masked), f"sa_{tag}_")
        np.testing.assert_array_equal(sa(x, x, x).detach().numpy(), g[f"sa_{tag}_y"])
    np.testing.assert_array_equal(load(mo.EncodingBlock(64, 1, 3), "enc_")(x).detach().numpy(), g["enc_y"])
    np.testing.assert_array_equal(load(mo.DecodingBlock(64, 1, 3), "dec_")(x, rep).detach().numpy(), g["dec_y"])


# ---------------------------------------------------------------------------------------- G6
def _g6_cases(g):
    import torch.nn as nn
    acts = {"ReLU": nn.ReLU, "LeakyReLU": nn.LeakyReLU, "Tanh": nn.Tanh}
    for i in range(int(g["n_cases"][0])):
        tag, n_in, n_out, hs, hd, out_init, act = [str(x) for x in g[f"c{i}_cfg"]]
        yield i, tag, int(n_in), int(n_out), eval(hs), int(hd), eval(out_init), acts[act]()


def test_g6_sequential_network_structure_and_init(golden):
    """
    networks/utils.py:53-191 of the reference (recorded by tests/golden/make_golden.py): the product's
    create_sequential_network / init_layer and the oracle's make_mlp build the same module tree (parameter
    names), draw the same initial weights from the same seed, and compute the same forward output.
    """
    import torch
    from oracle.cpu_ppo_loop import make_mlp
    from ppo_and_friends_amd.networks.feed_forward import create_sequential_network, init_layer
    g = golden("g6_network_utils")
    torch.manual_seed(4321)                      # the generator's seed and call order
    for i, tag, n_in, n_out, hs, hd, out_init, act in _g6_cases(g):
        net = create_sequential_network(n_in, n_out, hs, hd, act, out_init)
        x = torch.randn(11, n_in)
        names = [n for n, _ in net.named_parameters()]
        assert names == [str(n) for n in g[f"c{i}_names"]], tag
        np.testing.assert_array_equal(x.numpy(), g[f"c{i}_x"])
        for n, p in net.named_parameters():
            np.testing.assert_array_equal(p.detach().numpy(), g[f"c{i}_p_{n}"], err_msg=f"{tag} {n}")
        np.testing.assert_array_equal(net(x).detach().numpy(), g[f"c{i}_y"], err_msg=tag)
        if isinstance(hs, int) and hs > 0:       # the oracle's MLP builder (int hidden sizes)
            with torch.random.fork_rng():        # its own initial draws must not disturb the replayed stream
                ref = make_mlp(n_in, n_out, hs, hd, out_gain=out_init, activation=act)
            assert [n for n, _ in ref.named_parameters()] == names
            ref.load_state_dict({n: torch.from_numpy(g[f"c{i}_p_{n}"]) for n in names})
            np.testing.assert_array_equal(ref(torch.from_numpy(g[f"c{i}_x"])).detach().numpy(), g[f"c{i}_y"])
    # init_net_parameters on an LSTM / init_layer with a gain and a bias constant (same RNG stream)
    import torch.nn as nn
    from ppo_and_friends_amd.networks.lstm import LSTMNetwork        # noqa: F401  (same init rule inside)
    lstm = nn.LSTM(5, 8, 1)
    for name, param in lstm.named_parameters():
        if "weight" in name:
            nn.init.orthogonal_(param, 2 ** 0.5)
        elif "bias" in name:
            nn.init.constant_(param, 0.0)
    for n, p in lstm.named_parameters():
        np.testing.assert_array_equal(p.detach().numpy(), g[f"lstm_{n}"], err_msg=n)
    lin = init_layer(nn.Linear(6, 6), gain=0.3, bias_const=0.25)
    np.testing.assert_array_equal(lin.weight.detach().numpy(), g["lin_w"])
    np.testing.assert_array_equal(lin.bias.detach().numpy(), g["lin_b"])
    w = g["lin_w"]
    np.testing.assert_allclose(w @ w.T, 0.09 * np.eye(6), atol=1e-6)           # orthogonal rows, gain 0.3


def test_g7_schedulers_match_the_reference(golden):
    """
    utils/schedulers.py of this package (host scalars of the update: lr, entropy / intrinsic weights, clip bounds,
    and the policy freeze cycle) against the outputs recorded from the unmodified reference over the same
    40-iteration status trajectory: bit-exact, incl. the clamped ends, the first-iteration rule of the step
    schedule, persistent / per-call change detection, group completion and save tags of the freeze cycle.
    """
    import importlib.util
    import os
    import warnings
    from ppo_and_friends_amd.utils import schedulers as mine
    g = golden("g7_schedulers")
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)                       # only its scenario / driver helpers (no reference import)
    sc = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    for k, v in mk.scheduler_scenario().items():
        np.testing.assert_array_equal(sc[k], v)       # the committed inputs are the generator's
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)   # log(0) at iteration 0, clamped -- as in the reference
        got = mk.drive_schedulers(mine, sc)
    for k, v in got.items():
        np.testing.assert_array_equal(v, g[k], err_msg=k)
    assert got["frozen"].any() and len(set(got["active_idx"])) == 3
