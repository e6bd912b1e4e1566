"""
-m gpu: K12's fwd_bwd launch with the 256-wide networks' row tiles on workgroup PAIRS (csrc/ppo_update_rowpair.hpp) against
one workgroup per tile, over the shapes the pair body branches on: both networks 256 wide, depth 2 (one exchange) to 4
(five), a first layer too wide to be staged through LDS, ragged batch sizes with a tail mini-batch, tanh, a Gaussian head.
Every output tile is accumulated in the same K order in both forms: parameters, Adam moments, values and statistics must be
BITWISE equal after two iterations of two epochs (the second epoch restarts the record tags).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    "both_256": dict(widths=(256, 256), depth=3, O=11, A=1, cont=False, B=64, act="relu"),
    "depth2_wide_input": dict(widths=(128, 256), depth=2, O=20, A=3, cont=False, B=96, act="relu"),      # critic in_dim 60: not staged
    "depth4": dict(widths=(128, 256), depth=4, O=7, A=2, cont=True, B=48, act="tanh"),
    "ragged": dict(widths=(128, 256), depth=3, O=18, A=3, cont=False, B=200, act="relu"),                # 13 tiles + a tail mini-batch
}


def _run(case, pairs, monkeypatch, graphs):
    from ppo_and_friends_amd import fused_update
    from ppo_and_friends_amd.ppo import PPO
    from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
    from ppo_and_friends_amd.spaces import Box, Discrete
    c = CASES[case]
    monkeypatch.setattr(fused_update.FusedPolicyUpdate, "row_pairs", pairs)
    dev = torch.device("cuda", 0)
    E, T = 9, 37
    space = Box(-1.0, 1.0, (3,), np.float32) if c["cont"] else Discrete(5)
    env_gen = lambda: SyntheticFixedLengthEnv(E, c["O"], space, T, dev, reward="uniform", seed=47, num_agents=c["A"],
                                              critic_view="policy" if c["A"] > 1 else "local")
    sp = Box(-np.inf, np.inf, (c["O"],), np.float32)
    csp = Box(-np.inf, np.inf, (c["O"] * c["A"],), np.float32)
    act = torch.nn.Tanh if c["act"] == "tanh" else torch.nn.ReLU
    kw = lambda h: dict(hidden_size=h, hidden_depth=c["depth"], activation=act())
    pargs = dict(actor_kw_args=kw(c["widths"][0]), critic_kw_args=kw(c["widths"][1]))
    ppo = PPO(env_gen, {"p": (None, sp, csp, space, pargs)}, device=dev, random_seed=11, normalize_obs=False, normalize_rewards=False,
              envs_per_proc=E, ts_per_rollout=T, batch_size=c["B"], epochs_per_iter=2, update_mode="fused", save_state=False,
              use_graphs=graphs)
    before = fused_update.FusedPolicyUpdate.pair_launches
    for _ in range(2):
        ppo.rollout()
        ppo.train_on_rollout()
    pol = ppo.policies["p"]
    fused = ppo._fused_updater("p", c["B"])
    assert fused.split, fused.split_reason
    assert (fused.pairs_reason() == "") == pairs, fused.pairs_reason()
    assert (fused_update.FusedPolicyUpdate.pair_launches > before) == pairs
    stats = {k: float(v) for k, v in ppo.status_dict["p"].items() if isinstance(v, (int, float)) and not isinstance(v, bool)}
    return (pol.policy_params.detach().clone(), pol.policy_exp_avg.detach().clone(), pol.policy_exp_avg_sq.detach().clone(),
            pol.buffer.values.clone(), stats, pol.policy_step_counts.tolist())


@pytest.mark.parametrize("case", list(CASES))
def test_row_pairs_are_bitwise_one_workgroup_per_tile(case, monkeypatch):
    ref = _run(case, False, monkeypatch, graphs=False)
    for graphs in (False, True):
        got = _run(case, True, monkeypatch, graphs=graphs)
        for i, what in enumerate(("parameters", "exp_avg", "exp_avg_sq", "values")):
            assert torch.equal(got[i], ref[i]), f"{case} graphs={graphs}: {what} differ, max |d| {float((got[i] - ref[i]).abs().max()):.3e}"
        assert got[4] == ref[4] and got[5] == ref[5], (got[4], ref[4])
