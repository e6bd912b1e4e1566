"""
CPU tests of host-side logic that needs no GPU: the closing statistics of a rollout
(utils/rollout_stats.py, torch reductions) against the literal per-step bookkeeping of the
reference restated in oracle/rollout_stats_oracle.py, and the checkpoint interchange formats
(utils/reference_io.py).
"""
import io
import pickle
import sys
import types

import numpy as np
import torch

from oracle.rollout_stats_oracle import rollout_statistics_loop


def test_rollout_statistics_match_the_literal_loop():
    from ppo_and_friends_amd.utils.rollout_stats import rollout_statistics
    rng = np.random.default_rng(3)
    tt = torch.as_tensor
    for trial in range(40):
        T, E = int(rng.integers(2, 30)), int(rng.integers(1, 9))
        p = float(rng.choice([0.0, 0.05, 0.3]))
        term = rng.uniform(0, 1, (T, E)) < p
        max_ts = int(rng.integers(2, 10))
        boot, ep = np.zeros((T, E), bool), np.zeros(E, int)
        for t in range(T):
            ep += 1
            boot[t] = (~term[t]) & ((ep >= max_ts) | (t == T - 1))
            ep[term[t] | boot[t]] = 0
        r, nat = rng.uniform(-1, 1, (T, E)), rng.uniform(-2, 2, (T, E))
        intr, nr = rng.uniform(0, 0.1, (T, E)), rng.uniform(-3, 3, (T, E))
        omin, omax = rng.uniform(-5, -1, T), rng.uniform(1, 5, T)
        want = rollout_statistics_loop(r, nat, intr, omin, omax, term, boot, nr)
        got = rollout_statistics(tt(r), tt(nat), tt(term), tt(boot), tt(nr), 1, (tt(r.min()), tt(r.max())),
                                 (tt(nat.min()), tt(nat.max())), (tt(omin.min()), tt(omax.max())), T, tt(intr),
                                 (tt(intr.min()), tt(intr.max())))
        for k, v in got.items():
            np.testing.assert_allclose(v, want[k], rtol=1e-12, atol=1e-12, err_msg=f"{k} (trial {trial})")


def test_running_stats_pickles_carry_the_reference_class_path():
    from ppo_and_friends_amd.utils import reference_io as rio
    buf = io.BytesIO()
    rio.dump_running_stats({"a0": {"mean": np.arange(3, dtype=np.float32), "variance": np.ones(3, np.float32), "count": 5.5}}, buf)
    raw = buf.getvalue()
    assert b"ppo_and_friends.utils.stats" in raw and b"RunningMeanStd" in raw
    assert "ppo_and_friends.utils.stats" not in sys.modules          # placeholder modules are gone again
    # the reference side: its own class is found by module path and receives mean / variance / count
    mods = {n: types.ModuleType(n) for n in ("ppo_and_friends", "ppo_and_friends.utils", "ppo_and_friends.utils.stats")}

    class RunningMeanStd:            # stands for the reference's class in this test
        pass
    RunningMeanStd.__module__ = "ppo_and_friends.utils.stats"
    mods["ppo_and_friends.utils.stats"].RunningMeanStd = RunningMeanStd
    sys.modules.update(mods)
    try:
        obj = pickle.loads(raw)
    finally:
        for n in mods:
            sys.modules.pop(n, None)
    assert isinstance(obj["a0"], RunningMeanStd)
    np.testing.assert_array_equal(obj["a0"].mean, np.arange(3, dtype=np.float32))
    assert obj["a0"].count == 5.5
    # and back, without the reference installed
    back = rio.load_running_stats(io.BytesIO(raw))
    np.testing.assert_array_equal(back["a0"]["variance"], np.ones(3, np.float32))
    single = io.BytesIO()
    rio.dump_running_stats({"mean": np.float32(1.5), "variance": np.float32(2.0), "count": 7.0}, single)
    assert rio.load_running_stats(io.BytesIO(single.getvalue()))["count"] == 7.0


def test_adam_state_dict_round_trips_through_torch_optim_adam():
    from ppo_and_friends_amd.utils import reference_io as rio
    net = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.Linear(5, 2))      # sizes 15, 5, 10, 2: padded offsets
    total = sum((p.numel() + 3) // 4 * 4 for p in net.parameters())
    g = torch.Generator().manual_seed(0)
    m, v = torch.rand(total, generator=g), torch.rand(total, generator=g)
    sd = rio.adam_state_dict(net, m, v, step=17, lr=1e-3, betas=(0.9, 0.999), eps=1e-5)
    opt = torch.optim.Adam(net.parameters(), lr=3e-4, eps=1e-5)
    opt.load_state_dict(sd)                                  # the reference's _load_optimizers does exactly this
    st = opt.state_dict()
    assert st["param_groups"][0]["lr"] == 1e-3 and float(st["state"][2]["step"]) == 17
    m2, v2 = torch.zeros(total), torch.zeros(total)
    step, lr = rio.load_adam_state_dict(st, net, m2, v2)
    assert (step, lr) == (17, 1e-3)
    off = 0
    for p in net.parameters():
        n = p.numel()
        assert torch.equal(m2[off:off + n], m[off:off + n]) and torch.equal(v2[off:off + n], v[off:off + n])
        off += (n + 3) // 4 * 4
