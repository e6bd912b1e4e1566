"""
-m gpu: the HIP path against fixtures recorded from the UNMODIFIED reference (tests/golden/g12_*.npz,
make_golden_update.py): the reference's own PPO object was driven for whole iterations over a table-driven
environment; here the product's PPO runs over the same tables from the same initial weights, replays the recorded
raw actions (the one thing a device Philox stream cannot reproduce is the reference's CPU torch generator) and the
recorded shuffles, and must reproduce

  * every rollout: per-step values / log-probs / refined actions, the dataset (order, returns, advantages),
    the rollout statistics block of the status dict,
  * the very first mini-batch before any optimiser step: both losses and the full raw gradient of every parameter
    (1e-5, the north_star tolerance),
  * every epoch's statistics, and the weights / value-normaliser state after all optimiser steps.

Both update paths (fused K12 / K14 kernels; torch-ROCm modules + K2..K11) are checked.
"""
import ctypes as C

import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FF_SCENARIOS = {
    # name: (policy kwargs, PPO kwargs)
    "g12_c2_term": ({}, {}),
    "g12_c2_cut": ({}, {}),
    "g12_c4_mappo": (dict(leaky=True), {}),
    "g12_c3_gauss": (dict(leaky=True, lr=1e-4), {}),
    "g12_gauss_bounds": (dict(hidden=(32, 32)), {}),
    "g12_c2_icm": (dict(enable_icm=True), {}),
    "g12_c3_full": (dict(leaky=True, lr=1e-4, enable_icm=True),
                    dict(normalize_obs=True, normalize_rewards=True, obs_clip=(-2.0, 2.0), reward_clip=(-1.5, 1.5))),
    # the metric's own mini-batch shape, batch_size = 256 (ppo.py:134): 16 row tiles per network, multi-tile layered critic
    "g12_c2_b256": ({}, {}),
    "g12_c4_b256": (dict(leaky=True), {}),
    "g12_c3_b256": (dict(leaky=True, lr=1e-4, enable_icm=True),
                    dict(normalize_obs=True, normalize_rewards=True, obs_clip=(-2.0, 2.0), reward_clip=(-1.5, 1.5))),
}
# KL early stop (ppo.py:2221-2232): run through PPO.train_on_rollout itself, see test_kl_early_stop_matches_the_reference
KLSTOP_SCENARIOS = {
    "g12_c2_klstop": (dict(lr=1e-3, target_kl=0.01), {}),
    "g12_c2_icm_klstop": (dict(lr=3e-3, target_kl=0.005, enable_icm=True), {}),
}
B256 = ["g12_c2_b256", "g12_c3_b256", "g12_c4_b256"]


def _cfg(g):
    return dict(zip([str(x) for x in g["cfg_names"]], [int(x) for x in g["cfg"]]))


# ---- tolerances of quantities that have passed through optimiser steps -----------------------------------------------
# Single-mini-batch quantities (losses, gradients, returns, log-probs) are asserted at the north_star's 1e-5 throughout.
# Per-epoch statistics and final weights sit behind up to a few hundred Adam steps; their bounds are the deviations
# MEASURED on MI355X (tests/golden/measured_deviations.json = tests/golden/merge_deviations.py over a run of the GPU
# suite with PPOAF_RECORD_DEVIATIONS=<directory>;
# the kernels are bitwise reproducible, so a rerun measures the same numbers) times MARGIN -- not round numbers.
#   stat_dev   = max over the epoch's statistics of |got - want| / (0.1 + |want|)   (losses are O(1), KL / entropy O(1e-2))
#   weight_max = max |dw| over a network's weights after all steps; weight_share = share of weights with |dw| > 2e-5
MARGIN = 4.0
_MEASURED_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "measured_deviations.json")
try:
    import json as _json
    with open(_MEASURED_FILE) as _fh:
        MEASURED = _json.load(_fh)
except OSError:
    MEASURED = {}
_RECORDED = {}


def _bound(case, key, value, floor):
    """Assert `value` within MARGIN x the deviation measured for (case, key) -- at least `floor`, the resolution below
    which float32 noise of a different summation order lives -- and record it when asked to."""
    rec = os.environ.get("PPOAF_RECORD_DEVIATIONS")
    if rec:                                    # a directory: one file per process (the two-rank tests record from their ranks)
        slot = _RECORDED.setdefault(case, {})
        slot[key] = max(float(value), slot.get(key, 0.0))
        os.makedirs(rec, exist_ok=True)
        with open(os.path.join(rec, f"{os.getpid()}.json"), "w") as fh:
            _json.dump(_RECORDED, fh)
        return
    assert case in MEASURED and key in MEASURED[case], f"no measured deviation for {case} / {key}: record with PPOAF_RECORD_DEVIATIONS"
    limit = max(MARGIN * MEASURED[case][key], floor)
    assert value <= limit, f"{case} {key}: {value:.3e} > {limit:.3e} (= max({MARGIN} x measured {MEASURED[case][key]:.3e}, {floor:.0e}))"


def case_id(name, update_mode):
    """Fixture + everything that selects a kernel form (each form sums in its own order, so each has its own measured
    deviation): update mode and the path switches present in the environment."""
    keys = ("PPOAF_SPLIT_WGRAD", "PPOAF_OVERLAP_ICM", "PPOAF_GRAD_EXCHANGE",
            "PPOAF_FUSED_TAIL", "WORLD_SIZE")
    env = ",".join(f"{k[6:] if k.startswith('PPOAF_') else k}={os.environ[k]}" for k in keys if k in os.environ)
    return f"{name}/{update_mode}" + (f"[{env}]" if env else "")


def check_epoch_stats(case, got, want, what="stat_dev"):
    got, want = np.atleast_1d(np.asarray(got, dtype=np.float64)), np.atleast_1d(np.asarray(want, dtype=np.float64))
    _bound(case, what, float(np.max(np.abs(got - want) / (0.1 + np.abs(want)))), 2e-6)


def check_final_weights(case, tag, got, want):
    d = np.abs(np.asarray(got, dtype=np.float64) - np.asarray(want, dtype=np.float64))
    _bound(case, f"{tag}_weight_max", float(d.max()), 2e-6)
    _bound(case, f"{tag}_weight_share", float(np.mean(d > 2e-5)), 1e-4)


def agent_major(x):
    """[steps, E, A, ...] (fixture layout) -> [steps, A*E, ...] agent-major columns (the product's rows)."""
    x = np.swapaxes(x, 1, 2)
    return np.ascontiguousarray(x.reshape((x.shape[0], x.shape[1] * x.shape[2]) + x.shape[3:]))


def row_mapping(ref_obs, got_obs):
    """pi with got[pi[i]] == ref[i] (see tests/test_oracle_update_golden.py: the reference orders the episodes of one
    end-of-episode event by a hash-ordered agent list; rows are matched by their unique observation vectors)."""
    index = {row.tobytes(): i for i, row in enumerate(np.ascontiguousarray(got_obs))}
    assert len(index) == len(got_obs), "observation rows are not unique"
    return np.array([index[row.tobytes()] for row in np.ascontiguousarray(ref_obs)], dtype=np.int64)


class FixedPermLoader:
    """The loader surface PPO._ppo_batch_train / _icm_batch_train read, replaying one recorded shuffle."""

    def __init__(self, dataset, batch_size, perm):
        self.dataset, self.batch_size = dataset, int(batch_size)
        self._perm = torch.as_tensor(np.asarray(perm, dtype=np.int64), device=dataset.device)

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def epoch_permutation(self):
        return self._perm

    def prefetch(self):
        pass


def make_product(g, name, update_mode, dev):
    """The product's PPO over the fixture's tables, holding the fixture's initial weights."""
    from ppo_and_friends_amd.ppo import PPO
    from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
    from ppo_and_friends_amd.spaces import Box, Discrete
    c = _cfg(g)
    E, T, A, O = c["E"], c["T"], c["A"], c["O"]
    pk, ppo_kw = {**FF_SCENARIOS, **KLSTOP_SCENARIOS, **RANK_SCENARIOS}[name]
    names = set(g.files)
    continuous = "init_actor.distribution.log_std" in names
    n_out = int(g["init_actor.sequential_net.3.weight"].shape[0])
    c_in = int(g["init_critic.sequential_net.0.weight"].shape[1])
    if continuous:
        lo = {"g12_gauss_bounds": np.array([-1.0, -2.0, 0.0], np.float32)}.get(name, -np.ones(n_out, np.float32))
        hi = {"g12_gauss_bounds": np.array([1.0, 2.0, 5.0], np.float32)}.get(name, np.ones(n_out, np.float32))
        act_space = Box(lo, hi, (n_out,), np.float32)
    else:
        act_space = Discrete(n_out)

    class FixtureEnv(SyntheticFixedLengthEnv):
        def __init__(self):
            super().__init__(E, O, act_space, T, dev, num_agents=A, critic_view="policy" if c_in != O else "local",
                             term_prob=0.5 if g["term_table"].any() else 0.0)
            obs = agent_major(g["obs_table"])
            self.obs_table = torch.from_numpy(obs).to(dev)
            if c_in != O:
                v = self.obs_table.view(T + 1, A, E, O).permute(0, 2, 1, 3).reshape(T + 1, 1, E, A * O)
                self.critic_obs_table = v.expand(T + 1, A, E, A * O).reshape(T + 1, A * E, A * O).contiguous()
            else:
                self.critic_obs_table = self.obs_table
            self.reward_table = torch.from_numpy(agent_major(g["reward_table"])).to(dev)
            if g["term_table"].any():
                self.term_table = torch.from_numpy(np.tile(g["term_table"], (1, A))).to(dev)

    import torch.nn as nn
    h_a, h_c = pk.get("hidden", (128, 256 if int(g["init_critic.sequential_net.0.weight"].shape[0]) == 256 else 128))
    akw, ckw = dict(hidden_size=h_a), dict(hidden_size=h_c)
    if pk.get("leaky"):
        akw["activation"], ckw["activation"] = nn.LeakyReLU(), nn.LeakyReLU()
    pargs = dict(actor_kw_args=akw, critic_kw_args=ckw, lr=pk.get("lr", 3e-4), enable_icm=pk.get("enable_icm", False),
                 target_kl=pk.get("target_kl", 100.))
    sp, csp = Box(-np.inf, np.inf, (O,), np.float32), Box(-np.inf, np.inf, (c_in,), np.float32)
    kw = dict(normalize_obs=False, normalize_rewards=False)
    kw.update(ppo_kw)
    ppo = PPO(FixtureEnv, {"agent": (None, sp, csp, act_space, pargs)}, device=dev, random_seed=c["seed"],
              envs_per_proc=E, ts_per_rollout=T, batch_size=c["batch_size"], epochs_per_iter=c["epochs"],
              max_ts_per_ep=c["max_ts_per_ep"], update_mode=update_mode, save_state=False, **kw)
    pol = ppo.policies["agent"]
    for tag, net in (("actor", pol.actor), ("critic", pol.critic)) + ((("icm", pol.icm_model),) if pol.enable_icm else ()):
        sd = {k[len(f"init_{tag}."):]: torch.from_numpy(g[k]) for k in names if k.startswith(f"init_{tag}.")}
        missing, unexpected = net.load_state_dict(sd, strict=False)
        assert not unexpected and not [m for m in missing if "dist_m" not in m], (tag, missing, unexpected)
    return ppo, pol, c, continuous


def params_in_bucket_order(pol, bucket, net):
    """The slices of a flat bucket (gradients, ...) that belong to `net`'s parameters, in module order."""
    base = pol.policy_params.data_ptr()
    out = []
    for p in net.parameters():
        off = (p.data_ptr() - base) // 4
        out.append(bucket[off:off + p.numel()].reshape(-1))
    return torch.cat(out).cpu().numpy()


def first_minibatch_probe(ppo, pol, perm, B):
    """
    Losses + raw gradient bucket of the first mini-batch WITHOUT an optimiser step.
    fused path: one K12 fwd_bwd + the launch that completes the gradient bucket (complete-K wgrad | slab reduce);
    torch path: one _minibatch_step (value-normaliser state restored).
    """
    from ppo_and_friends_amd import _lib
    from ppo_and_friends_amd import kernels as K
    fused = ppo._fused_updater("agent", B)
    perm_t = torch.as_tensor(np.asarray(perm, dtype=np.int64), device=pol.device)
    if fused is not None:
        fused.begin_epoch(perm_t)
        args = fused._args_for(B)
        steps = pol.policy_step_counts.clone()              # the gradient launch's bookkeeping advances Adam's step counters
        fused.gradient_only(args)                           # fwd_bwd + (complete-K wgrad launch | slab reduce)
        torch.cuda.synchronize()
        pol.policy_step_counts.copy_(steps)
        return fused.totals.cpu().numpy().copy(), pol.policy_grads.clone()
    ds = pol.dataset
    totals = torch.zeros(9, dtype=torch.float64, device=pol.device)
    rs = ppo.value_normalizers["agent"].running_stats
    keep = [x.clone() for x in (rs.mean_t, rs.var_t, rs.count_t)]
    records = ppo._epoch_records("agent", ds, perm_t, B)
    ppo._minibatch_step("agent", ds, perm_t[:B].contiguous(), records[:, 0].reshape(1, 3), totals)
    torch.cuda.synchronize()
    for dst, src in zip((rs.mean_t, rs.var_t, rs.count_t), keep):
        dst.copy_(src)
    return totals.cpu().numpy().copy(), pol.policy_grads.clone()


def replay_rollout_and_check(ppo, pol, g, c, it, continuous, dev):
    """One rollout of the product with the reference's recorded raw actions replayed: per-step quantities, the dataset and
    the rollout statistics block against the fixture.  Returns pi (dataset row of the product for every reference row)."""
    E, T, A = c["E"], c["T"], c["A"]
    tol = dict(rtol=1e-5, atol=1e-5)                                   # north_star: within 1e-5 (fp32)
    keys, rkeys, gkeys = (list(g[k]) for k in ("rollout_status_keys", "rollout_range_keys", "global_status_keys"))
    sl = slice(it * T, (it + 1) * T)
    raw = agent_major(g["step_raw_actions"][sl])
    if not continuous:
        raw = raw.reshape(T, A * E, 1)
    ppo.replay_raw_actions = torch.from_numpy(raw).to(dev)
    ds = ppo.rollout()
    buf = pol.buffer
    # ---- per-step quantities, rows agent-major
    np.testing.assert_allclose(buf.observations.cpu().numpy(), agent_major(g["step_obs"][sl]).astype(np.float32), **tol)
    np.testing.assert_allclose(buf.log_probs.cpu().numpy(), agent_major(g["step_log_probs"][sl])[..., 0], **tol)
    per_step = [i for i, s in enumerate(g["values_calls_step"]) if it * T < s <= (it + 1) * T]
    first = {}
    for i in per_step:
        first.setdefault(int(g["values_calls_step"][i]), i)        # a step's first value call = V(obs_t)
    v_ref = agent_major(np.stack([g["values_calls"][first[s]] for s in range(it * T + 1, (it + 1) * T + 1)]))
    np.testing.assert_allclose(buf.values.cpu().numpy(), v_ref, **tol)
    got_act = buf.actions.cpu().numpy()
    want_act = agent_major(g["step_actions"][sl])
    np.testing.assert_allclose(got_act.reshape(want_act.shape), want_act, **tol)   # tanh + per-dimension rescale
    np.testing.assert_allclose(buf.rewards.cpu().numpy(), agent_major(g["step_rewards"][sl]), **tol)
    # ---- dataset
    pre = f"it{it}_ds_"
    assert len(ds) == len(g[pre + "advantages"])
    got_obs = ds.observations.cpu().numpy()
    if A == 1:                                                      # single agent: the order itself is the contract
        pi = np.arange(len(ds))
        np.testing.assert_allclose(got_obs, g[pre + "observations"], **tol)
        np.testing.assert_array_equal(ds.ep_lens.cpu().numpy(), g[pre + "ep_lens"])
    else:
        pi = row_mapping(g[pre + "observations"], got_obs)
    np.testing.assert_allclose(ds.critic_observations.cpu().numpy()[pi], g[pre + "critic_observations"], **tol)
    np.testing.assert_allclose(ds.values[torch.arange(len(ds), device=dev)].cpu().numpy()[pi], g[pre + "values"], **tol)
    np.testing.assert_allclose(ds.log_probs.cpu().numpy().reshape(-1)[pi], g[pre + "log_probs"], **tol)
    # returns / advantages are sums and differences of value-sized terms: 1e-5 of the value scale as absolute floor (the
    # values themselves were just compared at rtol 1e-5; |V| < 2 in every first iteration, ~10 once the critic has learnt
    # the returns of a second iteration, whose weights already carry the Adam steps' tolerated deviation)
    vscale = max(2.0, float(np.abs(g[pre + "values"]).max()))
    np.testing.assert_allclose(ds.rewards_to_go.cpu().numpy()[pi], g[pre + "rewards_to_go"], rtol=1e-5, atol=1e-5 * vscale)
    np.testing.assert_allclose(ds.advantages.cpu().numpy()[pi], g[pre + "advantages"], rtol=1e-5, atol=1e-5 * vscale)
    # ---- the rollout statistics block of the status dict
    sd, gs = ppo.status_dict["agent"], ppo.status_dict["global status"]
    for k in keys:
        np.testing.assert_allclose(sd[k], g["rollout_status"][it][keys.index(k)], rtol=1e-5, atol=1e-5, err_msg=f"{k} it {it}")
    for k in rkeys:
        np.testing.assert_allclose(sd[k], g["rollout_ranges"][it][rkeys.index(k)], rtol=1e-5, atol=1e-5, err_msg=f"{k} it {it}")
    for k in gkeys:
        np.testing.assert_allclose(gs[k], g["global_status"][it][gkeys.index(k)], rtol=1e-6, err_msg=f"{k} it {it}")
    return pi


@pytest.mark.parametrize("update_mode", ["fused", "torch"])
@pytest.mark.parametrize("name", sorted(FF_SCENARIOS))
def test_product_reproduces_the_reference_ppo_iterations(golden, name, update_mode):
    from ppo_and_friends_amd import kernels as K
    g = golden(name)
    dev = torch.device("cuda", 0)
    ppo, pol, c, continuous = make_product(g, name, update_mode, dev)
    E, T, A, B = c["E"], c["T"], c["A"], c["batch_size"]
    case = case_id(name, update_mode)
    ep = icm_ep = 0
    for it in range(c["iterations"]):
        pi = replay_rollout_and_check(ppo, pol, g, c, it, continuous, dev)
        sd = ppo.status_dict["agent"]
        # ---- first mini-batch of the run: losses + raw gradients before any optimiser step
        pol.train()
        if it == 0:
            sc, grads = first_minibatch_probe(ppo, pol, pi[g["epoch_perms"][0]], B)
            assert sc[8] == 1
            np.testing.assert_allclose([sc[K.SC_ACTOR], sc[K.SC_CRITIC]], g["mb0_losses"], rtol=1e-5, atol=1e-6)
            ga = params_in_bucket_order(pol, grads, pol.actor)
            gc = params_in_bucket_order(pol, grads, pol.critic)
            scale_a, scale_c = np.abs(g["mb0_actor_grad"]).max(), np.abs(g["mb0_critic_grad"]).max()
            np.testing.assert_allclose(ga, g["mb0_actor_grad"], rtol=1e-5, atol=1e-5 * scale_a,
                                       err_msg=f"actor gradient (max |g| {scale_a:.3e})")
            np.testing.assert_allclose(gc, g["mb0_critic_grad"], rtol=1e-5, atol=1e-5 * scale_c,
                                       err_msg=f"critic gradient (max |g| {scale_c:.3e})")
        # ---- epochs with the recorded shuffles
        for e in range(c["epochs"]):
            ppo._ppo_batch_train(FixedPermLoader(pol.dataset, B, pi[g["epoch_perms"][ep]]), "agent")
            got = np.array([sd["actor loss"], sd["critic loss"], sd["kl avg"], sd["weighted entropy"]])
            check_epoch_stats(case, got, g["epoch_stats"][ep])
            ep += 1
            if pol.enable_icm:
                ppo._icm_batch_train(FixedPermLoader(pol.dataset, B, pi[g["icm_epoch_perms"][icm_ep]]), "agent")
                check_epoch_stats(case, sd["icm loss"], g["icm_epoch_stats"][icm_ep][0], "icm_stat_dev")
                icm_ep += 1
        pol.clear_dataset()
    # ---- weights after every optimiser step of the run (Adam's m / sqrt(v) is sign-like where v is tiny: the bulk of the
    # weights stays within 2e-5, none moves by more than 2e-4; the measured maximum is in the message)
    for tag, net in (("actor", pol.actor), ("critic", pol.critic)) + ((("icm", pol.icm_model),) if pol.enable_icm else ()):
        want = np.concatenate([g[f"final_{tag}.{k}"].reshape(-1) for k, _ in net.named_parameters()])
        got = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu().numpy()
        check_final_weights(case, tag, got, want)
    rs = ppo.value_normalizers["agent"].running_stats
    np.testing.assert_allclose([float(rs.mean_t), float(rs.var_t), float(rs.count_t)], g["value_stats"], rtol=1e-5, atol=1e-5)


class RecordedShuffles:
    """Stands in for ppo.PermutationLoader inside PPO.train_on_rollout: hands out the reference's recorded shuffles in the
    order the reference drew them (PPO epoch, then -- with ICM -- the ICM pass of the same epoch) and refuses a draw the
    reference never made, i.e. an epoch beyond the one its KL early stop broke out of."""
    queue = []

    def __init__(self, dataset, batch_size, generator=None, prefetch_cache=None):
        self.dataset, self.batch_size = dataset, int(batch_size)

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def epoch_permutation(self):
        assert RecordedShuffles.queue, "the product asked for a shuffle beyond the epochs the reference ran"
        return torch.as_tensor(np.asarray(RecordedShuffles.queue.pop(0), dtype=np.int64), device=self.dataset.device)

    def prefetch(self):
        pass


# R = 2 ranks of the reference (make_golden_update.py: rank_scenarios; rank r's arrays through RankView)
RANK_SCENARIOS = {
    "g12_c2_r2": (dict(), {}),
    "g12_c4_r2": (dict(leaky=True), {}),
    "g12_c2_icm_r2_klstop": (dict(lr=3e-3, target_kl=0.005, enable_icm=True), {}),
}


class RankView:
    """Rank r's arrays of a fixture holding R ranks of the reference (keys `r<rank>.<key>`) under single-rank key names."""

    def __init__(self, g, rank):
        self._g, self._p = g, f"r{rank}."
        self.files = [k[len(self._p):] for k in g.files if k.startswith(self._p)]

    def __getitem__(self, k):
        return self._g[self._p + k]


def run_kl_stop_scenario(g, name, update_mode, dev, first_minibatch=None):
    """The fixture's iterations through the product's OWN epoch loop (PPO.train_on_rollout: KL early stop, overlapped
    PPO / ICM epochs, whatever the environment selected).  Checks, per iteration, that the loop
    ran exactly the epochs the reference ran, each epoch's statistics (on N > 1 ranks: the all-reduced ones, and the
    value normaliser fed by every rank's data), and the weights at the end.  `first_minibatch(ppo, pol, pi)`: a probe
    run after the first rollout, before any optimiser step.  Returns (ppo, epochs run per iteration)."""
    import ppo_and_friends_amd.ppo as ppo_module
    ppo, pol, c, continuous = make_product(g, name, update_mode, dev)
    B = c["batch_size"]
    case = case_id(name, update_mode) + "/train_on_rollout"
    sd = ppo.status_dict["agent"]
    ep, ran_all = 0, []
    assert float(pol.target_kl) == (float(g["target_kl"][0]) if "target_kl" in g.files else 100.0)
    keep_loader = ppo_module.PermutationLoader
    ppo_module.PermutationLoader = RecordedShuffles
    try:
        for it in range(c["iterations"]):
            pi = replay_rollout_and_check(ppo, pol, g, c, it, continuous, dev)
            pol.train()
            if it == 0 and first_minibatch is not None:
                first_minibatch(ppo, pol, pi)
            want = int(g["epochs_run"][it]) if "epochs_run" in g.files else c["epochs"]
            RecordedShuffles.queue = []
            for e in range(want):
                RecordedShuffles.queue.append(pi[g["epoch_perms"][ep + e]])
                if pol.enable_icm:
                    RecordedShuffles.queue.append(pi[g["icm_epoch_perms"][ep + e]])
            seen = []

            def record():
                rs_ = ppo.value_normalizers["agent"].running_stats
                seen.append((np.array([sd["actor loss"], sd["critic loss"], sd["kl avg"], sd["weighted entropy"]]),
                             sd.get("icm loss"), np.array([float(rs_.mean_t), float(rs_.var_t), float(rs_.count_t)])))

            def wrap(fn, when):
                def inner(loader, policy_id):
                    r = fn(loader, policy_id)
                    if when(r):
                        record()
                    return r
                return inner

            orig = (ppo._ppo_icm_epoch_overlapped, ppo._ppo_batch_train, ppo._icm_batch_train)
            ppo._ppo_icm_epoch_overlapped = wrap(orig[0], lambda r: bool(r))
            ppo._ppo_batch_train = wrap(orig[1], lambda r: not pol.enable_icm)
            ppo._icm_batch_train = wrap(orig[2], lambda r: True)
            try:
                ppo.train_on_rollout()
            finally:
                ppo._ppo_icm_epoch_overlapped, ppo._ppo_batch_train, ppo._icm_batch_train = orig
            assert len(seen) == want and not RecordedShuffles.queue, \
                f"iteration {it}: the product ran {len(seen)} epochs, the reference {want} (target_kl {pol.target_kl})"
            for e, (got, icm_loss, vstats) in enumerate(seen):
                if "epoch_value_stats" in g.files:      # R > 1 fixtures: the normaliser saw every rank's mini-batch (stats.py:47-50)
                    np.testing.assert_allclose(vstats, g["epoch_value_stats"][ep + e], rtol=1e-5, atol=1e-5, err_msg=f"value stats {it}/{e}")
                check_epoch_stats(case, got, g["epoch_stats"][ep + e])
                if pol.enable_icm:
                    check_epoch_stats(case, icm_loss, g["icm_epoch_stats"][ep + e][0], "icm_stat_dev")
                # the stop decision itself: strict `>` on the epoch's average, every epoch but the last one below the target
                assert (got[2] > pol.target_kl) == (e == want - 1 and want < c["epochs"]), (it, e, got[2])
            ep += want
            ran_all.append(want)
    finally:
        ppo_module.PermutationLoader = keep_loader
    for tag, net in (("actor", pol.actor), ("critic", pol.critic)) + ((("icm", pol.icm_model),) if pol.enable_icm else ()):
        want_w = np.concatenate([g[f"final_{tag}.{k}"].reshape(-1) for k, _ in net.named_parameters()])
        got_w = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu().numpy()
        check_final_weights(case, tag, got_w, want_w)
    rs = ppo.value_normalizers["agent"].running_stats
    np.testing.assert_allclose([float(rs.mean_t), float(rs.var_t), float(rs.count_t)], g["value_stats"], rtol=1e-5, atol=1e-5)
    return ppo, ran_all


# how the epoch loop is run: the graph-replayed launch chain, its three-launch and slab forms, the PPO / ICM epochs in turn
# instead of overlapped on two streams, and the torch-ROCm module path
KL_PATHS = {"chain": {}, "three_launches": {"PPOAF_FUSED_TAIL": "0"}, "slabs": {"PPOAF_SPLIT_WGRAD": "0"},
            "sequential": {"PPOAF_OVERLAP_ICM": "0"}, "torch": {}}


@pytest.mark.parametrize("path", sorted(KL_PATHS))
@pytest.mark.parametrize("name", sorted(KLSTOP_SCENARIOS))
def test_kl_early_stop_matches_the_reference(golden, name, path, monkeypatch):
    """
    ppo.py:2201-2232: fixtures recorded with a target_kl the reference reaches -- it left the epoch loop after 2 of 4
    epochs in iteration 0 (after 4 / 3 in iteration 1).  The product's own PPO.train_on_rollout must leave after the same
    epoch on every update path (a wrong or stale per-epoch total would run on, or stop early), with the ICM pass of the
    stopping epoch still run (ppo.py:2213-2214 precede the test), and end with the reference's weights.
    """
    from ppo_and_friends_amd import fused_update
    if path == "sequential" and "icm" not in name:
        pytest.skip("only the ICM scenario has a PPO / ICM epoch pair")
    for k, v in KL_PATHS[path].items():
        monkeypatch.setenv(k, v)
    g = golden(name)
    ppo, ran = run_kl_stop_scenario(g, name, "torch" if path == "torch" else "fused", torch.device("cuda", 0))
    assert ran == [int(x) for x in g["epochs_run"]] and ran[0] < _cfg(g)["epochs"]
    fused = ppo._fused_updater("agent", _cfg(g)["batch_size"])
    assert (fused is None) == (path == "torch")
    if fused is not None:
        assert fused.split == (path != "slabs") and (fused.tail_reason() == "") == (path not in ("slabs", "three_launches"))


@pytest.mark.parametrize("update_mode", ["fused", "fused_slabs", "torch"])
@pytest.mark.parametrize("name", ["g12_c5_mat", "g12_c5_b256"])
def test_product_reproduces_the_reference_mat_iterations(golden, name, update_mode, monkeypatch):
    """
    C5 shapes: the reference's own PPO object with MATPolicy (3 agents, O=18, Discrete(5), embedding 64, 1 block, 1 head;
    fixtures g12_c5_mat: 16-env mini-batches = 4 K15 tiles; g12_c5_b256: batch_size 256 = 52 tiles).  Autoregressive
    rollout (K16 / torch path) with the recorded actions replayed, shared-episode dataset incl. quirk Q14, first
    mini-batch (K15 launch): losses + the full gradient bucket before any optimiser step, epochs, final weights.
    `fused` = the split-wgrad chain (the default), `fused_slabs` = weight-gradient slabs + slab reduce (PPOAF_SPLIT_WGRAD=0).
    """
    if update_mode == "fused_slabs":
        monkeypatch.setenv("PPOAF_SPLIT_WGRAD", "0")
        update_mode = "fused"
    from ppo_and_friends_amd import _lib
    from ppo_and_friends_amd import kernels as K
    from ppo_and_friends_amd.ppo import PPO
    from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
    from ppo_and_friends_amd.policies.mat_policy import MATPolicy
    from ppo_and_friends_amd.spaces import Box, Discrete
    g = golden(name)
    c = _cfg(g)
    case = case_id(name, update_mode)
    E, T, A, O, B = c["E"], c["T"], c["A"], c["O"], c["batch_size"]
    dev = torch.device("cuda", 0)

    class FixtureEnv(SyntheticFixedLengthEnv):
        def __init__(self):
            super().__init__(E, O, Discrete(5), T, dev, num_agents=A)
            self.obs_table = self.critic_obs_table = torch.from_numpy(agent_major(g["obs_table"])).to(dev)
            self.reward_table = torch.from_numpy(agent_major(g["reward_table"])).to(dev)

    sp = Box(-np.inf, np.inf, (O,), np.float32)
    ppo = PPO(FixtureEnv, {"agent": (MATPolicy, sp, sp, Discrete(5), {})}, device=dev, random_seed=c["seed"],
              normalize_obs=False, normalize_rewards=False, envs_per_proc=E, ts_per_rollout=T, batch_size=B,
              epochs_per_iter=c["epochs"], max_ts_per_ep=c["max_ts_per_ep"], update_mode=update_mode, save_state=False)
    pol = ppo.policies["agent"]
    assert (ppo._fused_updater("agent", B) is not None) == (update_mode == "fused")
    if update_mode == "fused":
        assert ppo._fused_updater("agent", B).split == (os.environ.get("PPOAF_SPLIT_WGRAD", "1") == "1")
    sd0 = {"actor." + k[len("init_actor."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("init_actor.")}
    sd0.update({"critic." + k[len("init_critic."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("init_critic.")})
    missing, unexpected = pol.actor_critic.load_state_dict(sd0, strict=False)
    assert not [m for m in missing if "mask" not in m] and not [u for u in unexpected if "mask" not in u], (missing, unexpected)
    # the reference's agent orders: the one it had after finalize, then one recorded shuffle per rollout
    pol.agent_ids = np.array([str(a) for a in g["agent_ids"]])
    orders = iter(g["slot_orders"])

    def recorded_shuffle():
        pol.agent_ids = np.array([f"agent{i}" for i in next(orders)])

    pol.shuffle_agent_ids = recorded_shuffle
    tol = dict(rtol=1e-5, atol=1e-5)
    sd = ppo.status_dict["agent"]
    ep = 0
    for it in range(c["iterations"]):
        order = g["slot_orders"][it]
        sl = slice(it * T, (it + 1) * T)
        ppo.replay_raw_actions = torch.from_numpy(np.ascontiguousarray(g["step_raw_actions"][sl][:, :, order])).to(dev)
        ds = ppo.rollout()
        assert list(pol.agent_slot_order()) == list(order)
        pre = f"it{it}_ds_"
        np.testing.assert_array_equal(ds.observations.cpu().numpy(), g[pre + "observations"])     # [N, A, O], dataset agent order
        np.testing.assert_array_equal(ds.actions.cpu().numpy().reshape(g[pre + "actions"].shape), g[pre + "actions"])
        idx = torch.arange(len(ds), device=dev)
        np.testing.assert_allclose(ds.values[idx].cpu().numpy(), g[pre + "values"], **tol)
        np.testing.assert_allclose(ds.log_probs.cpu().numpy().reshape(g[pre + "log_probs"].shape), g[pre + "log_probs"], **tol)
        np.testing.assert_allclose(ds.rewards_to_go.cpu().numpy(), g[pre + "rewards_to_go"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(ds.advantages.cpu().numpy(), g[pre + "advantages"], rtol=1e-5, atol=2e-5)
        pol.train()
        if it == 0 and update_mode == "fused":
            # one K15 fwd_bwd + reduce launch on the first recorded mini-batch: no optimiser step taken
            fused = ppo._fused_updater("agent", B)
            fused.begin_epoch(torch.as_tensor(g["epoch_perms"][0], device=dev))
            args = fused._args_for(B)
            opt = pol.actor_critic_optim
            keep = opt.step_count.clone()
            lib, st = _lib.load(), K.stream()
            _lib.check(lib.ppoaf_mat_update_fwd_bwd(C.byref(args), st), "mat fwd_bwd")
            _lib.check(lib.ppoaf_mat_update_reduce(C.byref(args), st), "mat reduce")
            torch.cuda.synchronize()
            opt.step_count.copy_(keep)
            sc = fused.totals.cpu().numpy()
            np.testing.assert_allclose([sc[K.SC_ACTOR], sc[K.SC_CRITIC]], g["mb0_losses"], rtol=1e-5, atol=1e-6)
            base = pol.actor_critic.flat_params.data_ptr()
            got = torch.cat([pol.actor_critic.flat_grads[(p.data_ptr() - base) // 4:(p.data_ptr() - base) // 4 + p.numel()]
                             for net in (pol.actor, pol.critic) for p in net.parameters()]).cpu().numpy()
            want = g["mb0_total_grad"]
            scale = np.abs(want).max()
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5 * scale, err_msg=f"gradient bucket (max |g| {scale:.3e})")
        for e in range(c["epochs"]):
            ppo._ppo_batch_train(FixedPermLoader(pol.dataset, B, g["epoch_perms"][ep]), "agent")
            got = np.array([sd["actor loss"], sd["critic loss"], sd["kl avg"], sd["weighted entropy"]])
            check_epoch_stats(case, got, g["epoch_stats"][ep])
            ep += 1
        pol.clear_dataset()
    final = {"actor." + k[len("final_actor."):]: g[k] for k in g.files if k.startswith("final_actor.")}
    final.update({"critic." + k[len("final_critic."):]: g[k] for k in g.files if k.startswith("final_critic.")})
    check_final_weights(case, "actor_critic", np.concatenate([p.detach().cpu().numpy().reshape(-1) for k, p in pol.actor_critic.named_parameters()]),
                        np.concatenate([final[k].reshape(-1) for k, p in pol.actor_critic.named_parameters()]))


@pytest.mark.parametrize("name,S,n_act", [("g12_lstm_term", 4, 2), ("g12_lstm_cut", 3, 3)])
def test_product_reproduces_the_reference_lstm_iterations(golden, name, S, n_act):
    """
    LSTMNetwork actor / critic through the reference's own PPO object (fixtures g12_lstm_*): the product's sequence
    path (torch-ROCm nn.LSTM inside the device rollout / window dataset / update flow) with the recorded actions and
    shuffles: logged hidden states, dataset, epochs with hand-over + write-back, final weights.
    """
    from ppo_and_friends_amd.ppo import PPO
    from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
    from ppo_and_friends_amd.networks.lstm import LSTMNetwork
    from ppo_and_friends_amd.spaces import Box, Discrete
    g = golden(name)
    c = _cfg(g)
    case = case_id(name, "lstm")
    E, T, O, B = c["E"], c["T"], c["O"], c["batch_size"]
    dev = torch.device("cuda", 0)

    class FixtureEnv(SyntheticFixedLengthEnv):
        def __init__(self):
            super().__init__(E, O, Discrete(n_act), T, dev, term_prob=0.5 if g["term_table"].any() else 0.0)
            self.obs_table = self.critic_obs_table = torch.from_numpy(agent_major(g["obs_table"])).to(dev)
            self.reward_table = torch.from_numpy(agent_major(g["reward_table"])).to(dev)
            if g["term_table"].any():
                self.term_table = torch.from_numpy(g["term_table"]).to(dev)

    kw = dict(sequence_length=S, lstm_hidden_size=32, ff_hidden_size=32)
    sp = Box(-np.inf, np.inf, (O,), np.float32)
    ppo = PPO(FixtureEnv, {"agent": (None, sp, sp, Discrete(n_act), dict(ac_network=LSTMNetwork, actor_kw_args=dict(kw),
                                                                       critic_kw_args=dict(kw)))},
              device=dev, random_seed=c["seed"], normalize_obs=False, normalize_rewards=False, envs_per_proc=E,
              ts_per_rollout=T, batch_size=B, epochs_per_iter=c["epochs"], max_ts_per_ep=c["max_ts_per_ep"], save_state=False)
    pol = ppo.policies["agent"]
    for tag, net in (("actor", pol.actor), ("critic", pol.critic)):
        sd0 = {k[len(f"init_{tag}."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"init_{tag}.")}
        missing, unexpected = net.load_state_dict(sd0, strict=False)
        assert not missing and not unexpected, (tag, missing, unexpected)
    tol = dict(rtol=1e-5, atol=1e-5)
    sd = ppo.status_dict["agent"]
    ep = 0
    for it in range(c["iterations"]):
        sl = slice(it * T, (it + 1) * T)
        ppo.replay_raw_actions = torch.from_numpy(agent_major(g["step_raw_actions"][sl]).reshape(T, E, 1)).to(dev)
        ds = ppo.rollout()
        pre = f"it{it}_ds_"
        np.testing.assert_array_equal(ds.observations.cpu().numpy(), g[pre + "observations"])
        np.testing.assert_allclose(ds.log_probs.cpu().numpy().reshape(-1), g[pre + "log_probs"].reshape(-1), **tol)
        np.testing.assert_allclose(ds.rewards_to_go.cpu().numpy(), g[pre + "rewards_to_go"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(ds.advantages.cpu().numpy(), g[pre + "advantages"], rtol=1e-5, atol=2e-5)
        for k in ("actor_hidden", "actor_cell", "critic_hidden", "critic_cell"):
            np.testing.assert_allclose(getattr(ds, k)[torch.arange(E * T, device=dev)].cpu().numpy(), g[pre + k], err_msg=k, **tol)
        pol.train()
        for e in range(c["epochs"]):
            # the recorded shuffles are the 13th tuple entries = sampler index + (S - 1) (episode_info.py:960-962)
            ppo._ppo_batch_train(FixedPermLoader(pol.dataset, B, g["epoch_perms"][ep] - (S - 1)), "agent")
            got = np.array([sd["actor loss"], sd["critic loss"], sd["kl avg"], sd["weighted entropy"]])
            check_epoch_stats(case, got, g["epoch_stats"][ep])
            ep += 1
        pol.clear_dataset()
    for tag, net in (("actor", pol.actor), ("critic", pol.critic)):
        check_final_weights(case, tag, np.concatenate([p.detach().cpu().numpy().reshape(-1) for k, p in net.named_parameters()]),
                            np.concatenate([g[f"final_{tag}.{k}"].reshape(-1) for k, p in net.named_parameters()]))


# ---------------------------------------------------------------- unit fixtures g9 / g10 / g13 through the HIP kernels
def test_value_normalizer_kernels_match_reference_golden_g9(golden):
    """RunningStatNormalizer of the unmodified reference (utils/misc.py:61-128): update + normalise, denormalise,
    normalise without update -- through K5 (batch moments, Chan merge + integrate, normalise / denormalise kernels)."""
    from ppo_and_friends_amd.utils.misc import RunningStatNormalizer
    g = golden("g9_value_normalizer")
    dev = torch.device("cuda", 0)
    vn = RunningStatNormalizer("value_normalizer", dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    np.testing.assert_allclose(vn.denormalize(t(np.array([0.5, -1.0, 3.0], np.float32))).cpu().numpy(), g["denorm_fresh"], rtol=1e-6)
    for i in range(4):
        y = vn.normalize(t(g[f"in{i}"]))
        np.testing.assert_allclose(y.cpu().numpy(), g[f"norm{i}"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(vn.denormalize(t(g[f"probe{i}"])).cpu().numpy(), g[f"denorm{i}"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(vn.normalize(t(g[f"probe{i}"]), update_stats=False).cpu().numpy(), g[f"norm_noupdate{i}"],
                                   rtol=1e-5, atol=1e-5)
        rs = vn.running_stats
        np.testing.assert_allclose([float(rs.mean_t), float(rs.var_t), float(rs.count_t)], g[f"state{i}"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["disc", "cont"])
def test_icm_module_matches_reference_golden_g10(golden, tag):
    """ICM.forward of the unmodified reference (icm.py:22-430): intrinsic reward, inverse and forward losses, and the
    gradient of the training loss w.r.t. every parameter -- through this package's ICM (torch-ROCm MLPs + K8)."""
    from ppo_and_friends_amd.networks.icm import ICM
    from ppo_and_friends_amd.spaces import Box, Discrete
    g = golden("g10_icm")
    dev = torch.device("cuda", 0)
    if tag == "disc":
        icm = ICM(name="icm", obs_space=Box(-np.inf, np.inf, (6,), np.float32), action_space=Discrete(3), encoded_obs_dim=32,
                  encoder_hidden_size=32, inverse_hidden_size=32, forward_hidden_size=32)
    else:
        icm = ICM(name="icm", obs_space=Box(-np.inf, np.inf, (17,), np.float32), action_space=Box(-1.0, 1.0, (6,), np.float32),
                  encoded_obs_dim=32, encoder_hidden_size=64, inverse_hidden_size=32, forward_hidden_size=32,
                  inverse_hidden_depth=3, forward_hidden_depth=1)
    icm.to(dev)
    sd = {k[len(tag) + 3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"{tag}_p_")}
    missing, unexpected = icm.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    act = torch.from_numpy(g[f"{tag}_actions"]).to(dev)
    intr, inv_loss, f_loss = icm(torch.from_numpy(g[f"{tag}_obs1"]).to(dev), torch.from_numpy(g[f"{tag}_obs2"]).to(dev), act)
    loss = (1.0 - 0.8) * f_loss + 0.8 * inv_loss
    np.testing.assert_allclose(intr.detach().cpu().numpy().reshape(-1), g[f"{tag}_intr"].reshape(-1), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose([float(inv_loss), float(f_loss), float(loss)], g[f"{tag}_losses"], rtol=1e-5)
    names = [str(n) for n in g[f"{tag}_names"]]
    params = dict(icm.named_parameters())
    grads = torch.autograd.grad(loss, [params[k] for k in names])
    for k, gr in zip(names, grads):
        want = g[f"{tag}_g_{k}"]
        np.testing.assert_allclose(gr.cpu().numpy(), want, rtol=1e-5, atol=1e-5 * max(np.abs(want).max(), 1e-6), err_msg=k)


def test_env_filter_kernels_match_reference_golden_g13(golden):
    """The wrapper stack of the unmodified reference (ObservationNormalizer -> ObservationClipper -> RewardNormalizer
    with quirk Q3 -> RewardClipper, wrapper_utils.py:81-111) stand-alone: 2 agents ("policy" critic view), terminations,
    two passes -- through K13 (environments/filter_wrappers.py over csrc/env_filters.hip)."""
    from ppo_and_friends_amd.environments import filter_wrappers as fw
    from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
    from ppo_and_friends_amd.spaces import Discrete
    g = golden("g13_filters")
    E, T, A, O = (int(x) for x in g["cfg"])
    dev = torch.device("cuda", 0)

    class FixtureEnv(SyntheticFixedLengthEnv):
        def __init__(self):
            super().__init__(E, O, Discrete(2), T, dev, num_agents=A, critic_view="policy", term_prob=0.5)
            self.obs_table = torch.from_numpy(agent_major(g["obs_table"])).to(dev)
            v = self.obs_table.view(T + 1, A, E, O).permute(0, 2, 1, 3).reshape(T + 1, 1, E, A * O)
            self.critic_obs_table = v.expand(T + 1, A, E, A * O).reshape(T + 1, A * E, A * O).contiguous()
            self.reward_table = torch.from_numpy(agent_major(g["reward_table"])).to(dev)
            self.term_table = torch.from_numpy(np.tile(g["term_table"], (1, A))).to(dev)

    env = fw.wrap_environment(FixtureEnv, normalize_obs=True, normalize_rewards=True, obs_clip=(-1.5, 1.5),
                              reward_clip=(-1.0, 1.0), gamma=0.99)
    tol = dict(rtol=1e-5, atol=1e-5)
    action = torch.zeros(A * E, dtype=torch.int64, device=dev)
    for p in range(2):
        obs, cobs = env.reset()
        np.testing.assert_allclose(obs.cpu().numpy(), agent_major(g[f"p{p}_obs"])[0], **tol)
        np.testing.assert_allclose(cobs.cpu().numpy(), agent_major(g[f"p{p}_critic_obs"])[0], **tol)
        for t in range(T):
            obs, cobs, rew, term, trunc, tobs = env.step(action)
            np.testing.assert_allclose(obs.cpu().numpy(), agent_major(g[f"p{p}_obs"])[t + 1], err_msg=f"obs pass {p} step {t}", **tol)
            np.testing.assert_allclose(cobs.cpu().numpy(), agent_major(g[f"p{p}_critic_obs"])[t + 1], err_msg=f"critic obs {p}/{t}", **tol)
            np.testing.assert_allclose(rew.cpu().numpy(), agent_major(g[f"p{p}_rewards"])[t], err_msg=f"reward pass {p} step {t}", **tol)
            np.testing.assert_allclose(env.natural_reward.cpu().numpy(), agent_major(g[f"p{p}_natural"])[t], rtol=1e-6)
