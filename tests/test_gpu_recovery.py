"""
-m gpu: a launch with in-kernel hand-overs that does not complete (a partner workgroup was not resident in time: another
process on the GPU) must not cost the run.  The epoch's starting state is restored, the form that failed is switched off
for the rest of the run (with the reason) and the epoch is redone without it: bitwise what a run without that form produces.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(monkeypatch):
    from ppo_and_friends_amd.ppo import PPO
    from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
    from ppo_and_friends_amd.spaces import Box, Discrete
    dev = torch.device("cuda", 0)
    E, T, B, O, A = 8, 40, 32, 18, 3
    env_gen = lambda: SyntheticFixedLengthEnv(E, O, Discrete(5), T, dev, reward="uniform", seed=83, num_agents=A, critic_view="policy")
    sp, csp = Box(-np.inf, np.inf, (O,), np.float32), Box(-np.inf, np.inf, (A * O,), np.float32)
    settings = {"p": (None, sp, csp, Discrete(5), dict(actor_kw_args=dict(hidden_size=128), critic_kw_args=dict(hidden_size=256)))}
    ppo = PPO(env_gen, settings, device=dev, random_seed=5, normalize_obs=False, normalize_rewards=False, envs_per_proc=E,
              ts_per_rollout=T, batch_size=B, epochs_per_iter=2, update_mode="fused", save_state=False)
    pol = ppo.policies["p"]
    for _ in range(2):
        ppo.rollout()
        ppo.train_on_rollout()
    stats = {k: float(v) for k, v in ppo.status_dict["p"].items() if isinstance(v, (int, float)) and not isinstance(v, bool)}
    vs = ppo.value_normalizers["p"].running_stats
    return dict(w=pol.policy_params.detach().clone(), m=pol.policy_exp_avg.detach().clone(), stats=stats,
                vn=(float(vs.mean_t), float(vs.var_t), float(vs.count_t)), steps=pol.policy_step_counts.tolist())


def test_failed_row_pair_launch_is_redone_with_one_workgroup_per_tile(monkeypatch):
    """
    Round 4: a fwd_bwd launch in which a workgroup's partner did not answer in time (ppo_update_rowpair.hpp: the error word
    of the record region) costs one epoch's work, not the run: the epoch's starting state comes back, the pairs are
    switched off (with the reason) and the epoch runs again with one workgroup per tile -- bitwise what a run with
    row_pairs = False produces, since the two forms are bitwise equal anyway.  Simulated by setting the error word
    after the first epoch's launches.
    """
    from ppo_and_friends_amd import fused_update
    monkeypatch.setattr(fused_update.FusedPolicyUpdate, "row_pairs", False)
    tiles = _run(monkeypatch)
    monkeypatch.setattr(fused_update.FusedPolicyUpdate, "row_pairs", True)
    before = fused_update.FusedPolicyUpdate.pair_launches
    orig = fused_update.FusedPolicyUpdate.run_epoch
    state = {"failed": False, "updater": None}

    def run_epoch(self):
        orig(self)
        state["updater"] = self
        if not state["failed"]:
            assert self.pairs_reason() == "", self.pairs_reason()
            state["failed"] = True
            self._split_space[self._pair_region:self._pair_region + 4].view(torch.int32).fill_(1)

    monkeypatch.setattr(fused_update.FusedPolicyUpdate, "run_epoch", run_epoch)
    rec = _run(monkeypatch)
    assert state["failed"] and fused_update.FusedPolicyUpdate.pair_launches > before
    assert "did not answer" in state["updater"].pairs_reason()
    assert torch.equal(rec["w"], tiles["w"]) and torch.equal(rec["m"], tiles["m"])
    assert rec["stats"] == tiles["stats"] and rec["vn"] == tiles["vn"] and rec["steps"] == tiles["steps"]
