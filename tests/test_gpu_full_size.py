"""
-m gpu property tests at BASELINE.json's full sizes for the kernels whose oracles only run small cases:
size-independent facts that must hold whatever the data (counts, conservation, monotone progress on a
fixed batch, equality of two execution modes), called through the same C-ABI paths the benchmark uses.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _c_config(name, **kw):
    from ppo_and_friends_amd.ppo import PPO
    from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
    from ppo_and_friends_amd.spaces import Box, Discrete
    dev = torch.device("cuda", 0)
    if name == "C3":
        E, T, O, A, cv = 2048, 128, 17, 1, "local"
        space, pargs, pc = Box(-1.0, 1.0, (6,), np.float32), dict(actor_kw_args=dict(hidden_size=128),
                                                                  critic_kw_args=dict(hidden_size=256), enable_icm=True), None
        extra = dict(normalize_obs=True, normalize_rewards=True, obs_clip=(-10.0, 10.0), reward_clip=(-10.0, 10.0))
    elif name == "C2":
        E, T, O, A, cv = 4096, 128, 4, 1, "local"
        space, pargs, pc = Discrete(2), {}, None
        extra = dict(normalize_obs=False, normalize_rewards=False)
    elif name == "C4":
        E, T, O, A, cv = 1024, 128, 18, 3, "policy"
        space, pargs, pc = Discrete(5), dict(actor_kw_args=dict(hidden_size=128), critic_kw_args=dict(hidden_size=256)), None
        extra = dict(normalize_obs=False, normalize_rewards=False)
    elif name == "C5":
        from ppo_and_friends_amd.policies.mat_policy import MATPolicy
        E, T, O, A, cv = 1024, 128, 18, 3, "local"
        space, pargs, pc = Discrete(5), {}, MATPolicy
        extra = dict(normalize_obs=False, normalize_rewards=False)
    env_gen = lambda: SyntheticFixedLengthEnv(E, O, space, T, dev, reward="uniform", seed=1234, num_agents=A, critic_view=cv)
    sp = Box(-np.inf, np.inf, (O,), np.float32)
    csp = Box(-np.inf, np.inf, (O * A if cv == "policy" else O,), np.float32)
    extra.update(kw)
    return PPO(env_gen, {"p": (pc, sp, csp, space, pargs)}, device=dev, random_seed=1, envs_per_proc=E, ts_per_rollout=T,
               batch_size=256, epochs_per_iter=1, save_state=False, **extra), E, T, A


def test_c3_filters_counts_and_moments_at_full_size():
    """K13 at C3 size: exact sample counts (incl. the E^2 pooled reward updates of quirk Q3), unit-variance output."""
    ppo, E, T, A = _c_config("C3")
    env = ppo.env
    obs, _ = env.reset()
    w = env
    from ppo_and_friends_amd.environments import filter_wrappers as fw
    while not isinstance(w, fw.RewardNormalizer):
        w = w.env
    rn = w
    while not isinstance(w, fw.ObservationNormalizer):
        w = w.env
    on = w
    assert on.actor_running_stats["agent0"]["count"] == pytest.approx(1e-4 + E)
    # first batch from an (almost) empty tracker: the normalised batch is standardised by its own moments
    assert abs(float(obs.mean())) < 1e-3 and abs(float(obs.var(unbiased=False)) - 1.0) < 1e-2
    act = torch.zeros(E, 6, device=obs.device)
    for t in range(3):
        obs, _, rew, *_ = env.step(act)
    assert on.actor_running_stats["agent0"]["count"] == pytest.approx(1e-4 + 4 * E)
    assert rn.running_stats["agent0"]["count"] == pytest.approx(1e-4 + 3.0 * E * E, rel=1e-12)
    assert float(obs.abs().max()) <= 10.0 and float(rew.abs().max()) <= 10.0
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()


def test_c3_icm_update_makes_progress_and_overlap_is_exact(monkeypatch):
    """K14 + K12 at C3 size: a second epoch on the same rollout lowers the ICM loss; two-stream overlap == sequential,
    bitwise (both on the launch chain)."""
    ppo, E, T, A = _c_config("C3")
    ppo.rollout()
    ppo.train_on_rollout()
    first = ppo.status_dict["p"]["icm loss"]          # average over the FIRST epoch on this rollout
    outs = []
    for overlap in (True, False):
        ppo, E, T, A = _c_config("C3")                # same seed: the same rollout, the same first epoch
        ppo.overlap_icm = overlap
        ppo.epochs_per_iter = 2
        ppo.rollout()
        pol = ppo.policies["p"]
        ppo.train_on_rollout()
        second = ppo.status_dict["p"]["icm loss"]     # average over the SECOND epoch
        outs.append((pol.policy_params.clone(), pol.icm_model.flat_params.clone()))
        assert np.isfinite(first) and second < first, (first, second)
        assert int(pol.icm_optim.step_count.item()) == 2 * (E * T // 256)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_c5_mat_update_full_size_properties():
    """K15 / K16 at C5 size: every token written once, finite losses, critic loss falls on repeated epochs,
    fused == torch-path statistics on the first epoch (same rollout, same shuffle)."""
    from ppo_and_friends_amd.ppo import PermutationLoader
    ppo, E, T, A = _c_config("C5")
    ds = ppo.rollout()
    pol, buf = ppo.policies["p"], ppo.policies["p"].buffer
    assert len(ds) == E * T and buf.actions.shape == (T, E, A, 1)
    assert int(buf.actions.min()) >= 0 and int(buf.actions.max()) <= 4
    assert torch.isfinite(buf.log_probs).all() and float(buf.log_probs.max()) <= 0.0
    # sampled actions follow the policy: empirical log-prob mean ~ -entropy of a near-uniform head
    assert abs(float(buf.log_probs.mean()) + np.log(5.0)) < 0.05
    loader = PermutationLoader(pol.dataset, 256, ppo.loader_generator, ppo._perm_cache)
    pol.train()
    crit = []
    for _ in range(3):
        ppo._ppo_batch_train(loader, "p")
        sd = ppo.status_dict["p"]
        assert all(np.isfinite(sd[k]) for k in ("actor loss", "critic loss", "kl avg", "weighted entropy"))
        crit.append(sd["critic loss"])
    assert crit[2] < crit[0]
    assert int(pol.actor_critic_optim.step_count.item()) == 3 * (E * T // 256)


@pytest.mark.parametrize("name", ["C2", "C4"])
def test_k12_full_size_graph_replay_equals_eager_launches(name, monkeypatch):
    """
    (the launch chain)
    K12 at the metric's own size (C2: 2048 mini-batches per epoch; C4: MAPPO shape, 3 agents, 256-wide critic):
    the hipGraph-replayed chain and the eager launches are two execution modes of the same kernels -- bitwise
    equal weights, optimiser state and statistics; every mini-batch counted once; value normaliser saw every row.
    """
    outs = []
    for graphs in (True, False):
        ppo, E, T, A = _c_config(name, use_graphs=graphs)
        ppo.rollout()
        pol = ppo.policies["p"]
        ppo.train_on_rollout()
        n_mb = E * T * A // 256
        assert int(pol.policy_step_counts[0].item()) == int(pol.policy_step_counts[1].item()) == n_mb
        vs = ppo.value_normalizers["p"].running_stats
        assert abs(vs.count - (E * T * A + 1e-4)) < 1e-3
        sd = ppo.status_dict["p"]
        assert all(np.isfinite(sd[k]) for k in ("actor loss", "critic loss", "kl avg", "weighted entropy"))
        outs.append((pol.policy_params.clone(), pol.policy_exp_avg_sq.clone(),
                     [sd[k] for k in ("actor loss", "critic loss", "kl avg", "weighted entropy")], vs.mean.copy()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2] and np.array_equal(outs[0][3], outs[1][3])


@pytest.mark.parametrize("name", ["C2", "C3", "C4"])
def test_full_size_split_wgrad_chain_matches_the_slab_chain(name, monkeypatch):
    """
    The split-wgrad chain (fwd_bwd publishes activation / dz panels, one launch forms the complete weight gradients over
    all 256 rows on MFMA) against the slab chain (16-row partials in 16 slabs, summed in slab order) at the BASELINE
    shapes: ONE mini-batch's gradient bucket within 1e-5 of its largest entry per network and the same loss scalars;
    then a whole epoch of each: every mini-batch counted once, statistics to 2e-4, and the split chain bitwise equal to
    itself run to run (graph replay and eager alike).
    """
    grads, totals, epochs = {}, {}, {}
    for split in ("0", "1", "1"):
        monkeypatch.setenv("PPOAF_SPLIT_WGRAD", split)
        ppo, E, T, A = _c_config(name, use_graphs=len(epochs.get("1", [])) == 0)      # second split run: eager launches
        ppo.rollout()
        pol = ppo.policies["p"]
        pol.train()
        fused = ppo._fused_updater("p", 256)
        assert fused.split == (split == "1"), fused.split_reason
        perm = torch.randperm(len(pol.dataset), device=pol.device, generator=torch.Generator(device=pol.device).manual_seed(3))
        fused.begin_epoch(perm)
        steps = pol.policy_step_counts.clone()
        fused.gradient_only(fused._args_for(256))
        torch.cuda.synchronize()
        pol.policy_step_counts.copy_(steps)
        grads.setdefault(split, pol.policy_grads.clone()); totals.setdefault(split, fused.totals.clone())
        fused.begin_epoch(perm)
        fused.run_epoch()
        t = fused.end_epoch()
        n_mb = E * T * A // 256
        assert t[8] == n_mb and int(pol.policy_step_counts[0].item()) == n_mb
        epochs.setdefault(split, []).append((pol.policy_params.clone(), pol.policy_exp_avg_sq.clone(), t.copy()))
    na = int(ppo._fused_updater("p", 256).actor_desc.size)
    for sl, tag in ((slice(0, na), "actor"), (slice(na, None), "critic")):
        scale = float(grads["0"][sl].abs().max())
        d = float((grads["1"][sl] - grads["0"][sl]).abs().max())
        assert d <= 1e-5 * scale, f"{tag}: max |dg| {d:.3e} against max |g| {scale:.3e}"
    np.testing.assert_allclose(totals["1"].cpu().numpy(), totals["0"].cpu().numpy(), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(epochs["1"][0][2][:5] / n_mb, epochs["0"][0][2][:5] / n_mb, rtol=2e-4, atol=1e-4)   # (means of cancelling O(1) terms)
    a, b = epochs["1"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and np.array_equal(a[2], b[2])


@pytest.mark.parametrize("name", ["C2", "C3", "C4"])
def test_full_size_fused_tail_is_bitwise_the_three_launch_chain(name, monkeypatch):
    """
    Round 4: fwd_bwd -> ppoaf_ppo_update_wgrad_adam (weight gradients, clip norms from tagged records every workgroup
    waits for, clip + Adam on the workgroup's own elements: ONE launch) against fwd_bwd -> wgrad -> Adam at the BASELINE
    shapes, a whole epoch each on the same rollout and shuffle (C2: 2048 mini-batches, 153 workgroups per launch; C3 / C4:
    256-wide critic, 369 workgroups): the same jobs, folds and summation orders, so parameters, both Adam moments, the
    gradient bucket of the last mini-batch, step counters, normaliser state and totals are BITWISE equal -- graph replay
    and eager launches alike -- and no wait ran out of its budget.
    """
    from ppo_and_friends_amd import fused_update
    monkeypatch.setenv("PPOAF_OVERLAP_ICM", "0")
    outs = {}
    for tail, graphs in (("0", True), ("1", True), ("1", False)):
        monkeypatch.setenv("PPOAF_FUSED_TAIL", tail)
        before = fused_update.FusedPolicyUpdate.tail_launches
        ppo, E, T, A = _c_config(name, use_graphs=graphs)
        ppo.rollout()
        pol = ppo.policies["p"]
        pol.train()
        fused = ppo._fused_updater("p", 256)
        assert fused.split and (fused.tail_reason() == "") == (tail == "1"), (fused.split_reason, fused.tail_reason())
        perm = torch.randperm(len(pol.dataset), device=pol.device, generator=torch.Generator(device=pol.device).manual_seed(5))
        fused.begin_epoch(perm)
        fused.run_epoch()
        t = fused.end_epoch()
        n_mb = E * T * A // 256
        assert t[8] == n_mb and int(pol.policy_step_counts[0].item()) == int(pol.policy_step_counts[1].item()) == n_mb
        assert (fused_update.FusedPolicyUpdate.tail_launches > before) == (tail == "1")
        assert fused.tail_reason() == ("" if tail == "1" else "off (PPOAF_FUSED_TAIL=0)")      # no launch failed
        if tail == "1":
            assert int(fused._tail_ctl[2].item()) == 0 and int(fused._tail_ctl[0].item()) == n_mb      # error word, launches completed
        outs[(tail, graphs)] = (pol.policy_params.clone(), pol.policy_exp_avg.clone(), pol.policy_exp_avg_sq.clone(),
                                pol.policy_grads.clone(), t.copy(), fused.vn_mean.clone(), fused.vn_var.clone(), int(fused.cursor.item()))
    ref = outs[("0", True)]
    for key in (("1", True), ("1", False)):
        got = outs[key]
        for i, what in enumerate(("parameters", "exp_avg", "exp_avg_sq", "gradient bucket of the last mini-batch")):
            assert torch.equal(got[i], ref[i]), f"{key}: {what} differ, max |d| {float((got[i] - ref[i]).abs().max()):.3e}"
        assert np.array_equal(got[4], ref[4]), (got[4], ref[4])
        assert torch.equal(got[5], ref[5]) and torch.equal(got[6], ref[6]) and got[7] == ref[7]


@pytest.mark.parametrize("name", ["C3", "C4"])
def test_full_size_row_pairs_are_bitwise_the_one_workgroup_tiles(name, monkeypatch):
    """
    Round 4: the 256-wide critic's 16-row tiles on PAIRS of workgroups (ppo_update_rowpair.hpp: every hidden pass split by
    output columns, the halves exchanged as tagged records, 2 depth - 3 = 3 exchanges per mini-batch) against one
    workgroup per tile, a whole epoch each at the BASELINE shapes on the same rollout and shuffle.  Each output tile is
    accumulated in the same K order in both forms, so parameters, both Adam moments, the last gradient bucket, the critic
    values written back, totals and normaliser state are BITWISE equal -- graph replay and eager launches alike -- and
    no partner ever failed to answer.
    """
    from ppo_and_friends_amd import fused_update
    monkeypatch.setenv("PPOAF_OVERLAP_ICM", "0")
    outs = {}
    for pairs, graphs in ((False, True), (True, True), (True, False)):
        monkeypatch.setattr(fused_update.FusedPolicyUpdate, "row_pairs", pairs)
        before = fused_update.FusedPolicyUpdate.pair_launches
        ppo, E, T, A = _c_config(name, use_graphs=graphs)
        ppo.rollout()
        pol = ppo.policies["p"]
        pol.train()
        fused = ppo._fused_updater("p", 256)
        assert fused.split and (fused.pairs_reason() == "") == pairs, (fused.split_reason, fused.pairs_reason())
        perm = torch.randperm(len(pol.dataset), device=pol.device, generator=torch.Generator(device=pol.device).manual_seed(9))
        for _ in range(2):                                   # the second epoch restarts the record tags
            fused.begin_epoch(perm)
            fused.run_epoch()
            t = fused.end_epoch()
        n_mb = E * T * A // 256
        assert t[8] == n_mb and int(pol.policy_step_counts[0].item()) == int(pol.policy_step_counts[1].item()) == 2 * n_mb
        assert (fused_update.FusedPolicyUpdate.pair_launches > before) == pairs
        assert fused.pairs_reason() == ("" if pairs else "off (row_pairs = False)")           # no launch failed
        outs[(pairs, graphs)] = (pol.policy_params.clone(), pol.policy_exp_avg.clone(), pol.policy_exp_avg_sq.clone(),
                                 pol.policy_grads.clone(), pol.buffer.values.clone(), t.copy(), fused.vn_mean.clone(),
                                 fused.vn_var.clone(), int(fused.cursor.item()))
    ref = outs[(False, True)]
    for key in ((True, True), (True, False)):
        got = outs[key]
        for i, what in enumerate(("parameters", "exp_avg", "exp_avg_sq", "gradient bucket of the last mini-batch", "values")):
            assert torch.equal(got[i], ref[i]), f"{key}: {what} differ, max |d| {float((got[i] - ref[i]).abs().max()):.3e}"
        assert np.array_equal(got[5], ref[5]), (got[5], ref[5])
        assert torch.equal(got[6], ref[6]) and torch.equal(got[7], ref[7]) and got[8] == ref[8]


def test_c5_mat_split_wgrad_chain_matches_the_slab_form_at_full_size(monkeypatch):
    """
    K15 at C5 size (52 token tiles, 832 token rows): ONE mini-batch's gradient bucket of the split-wgrad chain (input / dz
    panels + mat_update_wgrad_kernel over all token rows) within 1e-5 of its largest entry of the slab form's (per-tile
    slabs + slab reduce), the same loss scalars; then an epoch of each: every mini-batch counted once, statistics to
    2e-4, and the split chain bitwise equal to itself run to run (graph replay and eager launches).
    """
    import ctypes as C
    from ppo_and_friends_amd import _lib
    from ppo_and_friends_amd import kernels as K
    grads, totals, epochs = {}, {}, {}
    for split in ("0", "1", "1"):
        monkeypatch.setenv("PPOAF_SPLIT_WGRAD", split)
        ppo, E, T, A = _c_config("C5", use_graphs=len(epochs.get("1", [])) == 0)
        ppo.rollout()
        pol = ppo.policies["p"]
        pol.train()
        fused = ppo._fused_updater("p", 256)
        assert fused.split == (split == "1")
        perm = torch.randperm(len(pol.dataset), device=pol.device, generator=torch.Generator(device=pol.device).manual_seed(3))
        fused.begin_epoch(perm)
        opt = pol.actor_critic_optim
        steps = opt.step_count.clone()
        args, lib, st = fused._args_for(256), _lib.load(), K.stream()
        _lib.check(lib.ppoaf_mat_update_fwd_bwd(C.byref(args), st), "mat fwd_bwd")
        _lib.check(lib.ppoaf_mat_update_reduce(C.byref(args), st), "mat reduce")
        torch.cuda.synchronize()
        opt.step_count.copy_(steps)
        grads.setdefault(split, pol.actor_critic.flat_grads.clone()); totals.setdefault(split, fused.totals.clone())
        fused.begin_epoch(perm)
        fused.run_epoch()
        t = fused.end_epoch()
        n_mb = E * T // 256
        assert t[8] == n_mb and int(opt.step_count.item()) == n_mb
        epochs.setdefault(split, []).append((pol.actor_critic.flat_params.clone(), opt.exp_avg_sq.clone(), t.copy()))
    scale = float(grads["0"].abs().max())
    d = float((grads["1"] - grads["0"]).abs().max())
    assert d <= 1e-5 * scale, f"max |dg| {d:.3e} against max |g| {scale:.3e}"
    np.testing.assert_allclose(totals["1"].cpu().numpy(), totals["0"].cpu().numpy(), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(epochs["1"][0][2][:5] / n_mb, epochs["0"][0][2][:5] / n_mb, rtol=2e-4, atol=1e-4)
    a, b = epochs["1"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_c5_mat_fused_tail_is_bitwise_the_three_launch_chain(monkeypatch):
    """
    K15's fused tail (round 4: mat_update_wgrad_adam_kernel -- weight gradients of the 18 linears and the small tensors, the
    clip norm from tagged records every workgroup waits for, clip + Adam on the workgroup's own elements, ONE launch) against
    ppoaf_mat_update_reduce + ppoaf_adam_step_prenormed at C5 size, a whole epoch each on the same rollout and shuffle
    (512 mini-batches, 146 workgroups per launch): parameters, both Adam moments, the gradient bucket of the last
    mini-batch, step counter, normaliser state and totals BITWISE equal, graph replay and eager launches alike.
    """
    from ppo_and_friends_amd import fused_update
    outs = {}
    for tail, graphs in (("0", True), ("1", True), ("1", False)):
        monkeypatch.setenv("PPOAF_FUSED_TAIL", tail)
        before = fused_update.FusedPolicyUpdate.tail_launches
        ppo, E, T, A = _c_config("C5", use_graphs=graphs)
        ppo.rollout()
        pol = ppo.policies["p"]
        pol.train()
        fused = ppo._fused_updater("p", 256)
        assert fused.split and (fused.tail_reason() == "") == (tail == "1"), (fused.split_reason, fused.tail_reason())
        perm = torch.randperm(len(pol.dataset), device=pol.device, generator=torch.Generator(device=pol.device).manual_seed(7))
        fused.begin_epoch(perm)
        fused.run_epoch()
        t = fused.end_epoch()
        opt = pol.actor_critic_optim
        n_mb = E * T // 256
        assert t[8] == n_mb and int(opt.step_count.item()) == n_mb
        assert (fused_update.FusedPolicyUpdate.tail_launches > before) == (tail == "1")
        if tail == "1":
            assert fused.tail_reason() == "" and int(fused._tail_ctl[2].item()) == 0 and int(fused._tail_ctl[0].item()) == n_mb
        outs[(tail, graphs)] = (pol.actor_critic.flat_params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(),
                                pol.actor_critic.flat_grads.clone(), t.copy(), fused.vn_mean.clone(), fused.vn_var.clone(),
                                float(opt.grad_norm.item()))
    ref = outs[("0", True)]
    for key in (("1", True), ("1", False)):
        got = outs[key]
        for i, what in enumerate(("parameters", "exp_avg", "exp_avg_sq", "gradient bucket of the last mini-batch")):
            assert torch.equal(got[i], ref[i]), f"{key}: {what} differ, max |d| {float((got[i] - ref[i]).abs().max()):.3e}"
        assert np.array_equal(got[4], ref[4]), (got[4], ref[4])
        assert torch.equal(got[5], ref[5]) and torch.equal(got[6], ref[6]) and got[7] == ref[7]


def test_c3_icm_split_wgrad_chain_matches_the_slab_form_at_full_size(monkeypatch):
    """
    K14 at C3 size (B = 256, H = 128, O = 17, Box(6)): ONE mini-batch's gradient bucket of the split-wgrad chain (dz /
    input panels + icm_wgrad_kernel over all rows and both observation streams, Adam on the spot) within 1e-5 of its
    largest entry of the slab form's, the weights after that one Adam step within 1e-6, the same loss; then an epoch of
    each: every mini-batch counted once, the epoch's mean loss to 2e-4, and the split chain bitwise equal to itself run to
    run (graph replay and eager launches).
    """
    from ppo_and_friends_amd.fused_update import FusedIcmUpdate
    grads, weights, losses, epochs = {}, {}, {}, {}
    for split in ("0", "1", "1"):
        monkeypatch.setenv("PPOAF_SPLIT_WGRAD", split)
        ppo, E, T, A = _c_config("C3", use_graphs=len(epochs.get("1", [])) == 0)
        ppo.rollout()
        pol = ppo.policies["p"]
        pol.train()
        fused = FusedIcmUpdate(ppo, "p")
        assert fused.split == (split == "1")
        perm = torch.randperm(len(pol.dataset), device=pol.device, generator=torch.Generator(device=pol.device).manual_seed(3))
        fused.begin_epoch(perm)
        fused._one(fused._args_for(256))
        torch.cuda.synchronize()
        grads.setdefault(split, pol.icm_model.flat_grads.clone()); weights.setdefault(split, pol.icm_model.flat_params.clone())
        losses.setdefault(split, fused.totals.clone())
        fused.begin_epoch(perm)
        fused.run_epoch()
        t = fused.end_epoch()
        n_mb = E * T // 256
        assert t[1] == n_mb and int(pol.icm_optim.step_count.item()) == n_mb + 1
        epochs.setdefault(split, []).append((pol.icm_model.flat_params.clone(), pol.icm_optim.exp_avg_sq.clone(), np.array(t, dtype=np.float64)))
    scale = float(grads["0"].abs().max())
    d = float((grads["1"] - grads["0"]).abs().max())
    assert d <= 1e-5 * scale, f"max |dg| {d:.3e} against max |g| {scale:.3e}"
    assert float((weights["1"] - weights["0"]).abs().max()) <= 1e-6
    np.testing.assert_allclose(losses["1"].cpu().numpy(), losses["0"].cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(epochs["1"][0][2][0] / n_mb, epochs["0"][0][2][0] / n_mb, rtol=2e-4)   # (measured 4.4e-5 after 1024 Adam steps)
    a, b = epochs["1"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_c3_icm_single_launch_is_bitwise_the_three_launches(monkeypatch):
    """
    Round 4: K14's encoder / model / encoder-backward kernels as ONE launch per mini-batch on workgroup pairs (icm_fused_kernel:
    encodings and encoding gradients change hands as tagged records, activations stay in LDS) against the three launches, at
    C3 size: first one mini-batch's gradient bucket and losses, then two epochs (the second restarts the record tags) --
    parameters, both Adam moments and totals BITWISE equal, graph replay and eager launches, unconfined and on an XCD half.
    """
    from ppo_and_friends_amd.fused_update import FusedIcmUpdate
    outs = {}
    for fuse, graphs, half in ((False, True, 0), (True, True, 0), (True, False, 0), (True, True, 2), (False, True, 2)):
        monkeypatch.setattr(FusedIcmUpdate, "fuse_kernels", fuse)
        before = FusedIcmUpdate.fused_launches
        ppo, E, T, A = _c_config("C3", use_graphs=graphs)
        ppo.rollout()
        pol = ppo.policies["p"]
        pol.train()
        fused = FusedIcmUpdate(ppo, "p")
        fused.xcd_half = half
        assert fused.split
        perm = torch.randperm(len(pol.dataset), device=pol.device, generator=torch.Generator(device=pol.device).manual_seed(3))
        fused.begin_epoch(perm)
        assert (fused.fuse_reason() == "") == fuse, fused.fuse_reason()
        steps = pol.icm_optim.step_count.clone()
        keep = [t.clone() for t in fused._epoch_state()]
        fused._one(fused._args_for(256))
        torch.cuda.synchronize()
        g1, l1 = pol.icm_model.flat_grads.clone(), fused.totals.clone()
        for t, k in zip(fused._epoch_state(), keep):
            t.copy_(k)
        for _ in range(2):
            fused.begin_epoch(perm)
            fused.run_epoch()
            t = fused.end_epoch()
        n_mb = E * T // 256
        assert t[1] == n_mb and int(pol.icm_optim.step_count.item()) == int(steps.item()) + 2 * n_mb
        assert (FusedIcmUpdate.fused_launches > before) == fuse and fused.fuse_reason() == ("" if fuse else "off (fuse_kernels = False)")
        outs[(fuse, graphs, half)] = (g1, l1, pol.icm_model.flat_params.clone(), pol.icm_optim.exp_avg.clone(),
                                      pol.icm_optim.exp_avg_sq.clone(), np.array(t, dtype=np.float64))
    for ref_key, keys in (((False, True, 0), [(True, True, 0), (True, False, 0)]), ((False, True, 2), [(True, True, 2)])):
        ref = outs[ref_key]
        for key in keys:
            got = outs[key]
            for i, what in enumerate(("first gradient bucket", "first losses", "parameters", "exp_avg", "exp_avg_sq")):
                assert torch.equal(got[i], ref[i]), f"{key}: {what} differ, max |d| {float((got[i] - ref[i]).abs().max()):.3e}"
            assert np.array_equal(got[5], ref[5]), (got[5], ref[5])


def test_failed_icm_single_launch_is_redone_on_three_launches(monkeypatch):
    """
    An icm_fused_kernel launch in which a workgroup's partner did not answer (word 0 of the record region; simulated after the
    first epoch's launches) costs that epoch's work, not the run: the epoch's starting state comes back, the single launch is
    switched off with the reason, the epoch runs again on three launches -- bitwise what fuse_kernels = False produces.
    """
    from ppo_and_friends_amd.fused_update import FusedIcmUpdate
    outs = {}
    for fuse in (False, True):
        monkeypatch.setattr(FusedIcmUpdate, "fuse_kernels", fuse)
        ppo, E, T, A = _c_config("C3")
        ppo.rollout()
        pol = ppo.policies["p"]
        pol.train()
        fused = FusedIcmUpdate(ppo, "p")
        perm = torch.randperm(len(pol.dataset), device=pol.device, generator=torch.Generator(device=pol.device).manual_seed(4))
        totals = []
        for ep in range(2):
            fused.begin_epoch(perm)
            fused.run_epoch()
            if fuse and ep == 0:
                assert fused.fuse_reason() == ""
                fused._split_space[:4].view(torch.int32).fill_(1)          # "a partner did not answer"
            totals.append(np.array(fused.end_epoch(), dtype=np.float64))
        if fuse:
            assert "ran out of time" in fused.fuse_reason()
        outs[fuse] = (pol.icm_model.flat_params.clone(), pol.icm_optim.exp_avg.clone(), pol.icm_optim.exp_avg_sq.clone(),
                      int(pol.icm_optim.step_count.item()), totals)
    a, b = outs[True], outs[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[3] == b[3]
    assert all(np.array_equal(x, y) for x, y in zip(a[4], b[4]))


def test_c3_xcd_halves_change_placement_only():
    """
    args.xcd_half (K12 / K14 fwd_bwd launches confined to XCDs 0-3 / 4-7, as the overlapped PPO / ICM epochs run them) moves
    workgroups, not arithmetic: one mini-batch's gradient buckets and loss totals are bitwise those of the unconfined
    launches, for both halves.
    """
    from ppo_and_friends_amd.fused_update import FusedIcmUpdate
    out = {}
    for half in (0, 1, 2):
        ppo, E, T, A = _c_config("C3", use_graphs=False)
        ppo.rollout()
        pol = ppo.policies["p"]
        pol.train()
        perm = torch.randperm(len(pol.dataset), device=pol.device, generator=torch.Generator(device=pol.device).manual_seed(3))
        fused = ppo._fused_updater("p", 256)
        fused.xcd_half = half
        fused.begin_epoch(perm)
        args = fused._args_for(256)
        assert args.xcd_half == half
        fused.gradient_only(args)
        icm = FusedIcmUpdate(ppo, "p")
        icm.xcd_half = half
        icm.begin_epoch(perm)
        iargs = icm._args_for(256)
        assert iargs.xcd_half == half
        icm._one(iargs)
        torch.cuda.synchronize()
        out[half] = (pol.policy_grads.clone(), fused.totals.clone(), pol.icm_model.flat_grads.clone(), pol.icm_model.flat_params.clone(),
                     icm.totals.clone())
    for half in (1, 2):
        for a, b in zip(out[0], out[half]):
            assert torch.equal(a, b), half
