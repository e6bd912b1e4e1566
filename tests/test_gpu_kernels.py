"""
-m gpu parity tests: every HIP kernel, called through the C ABI
(ppo_and_friends_amd.kernels -> ctypes -> libppoaf_hip.so), against the oracle
on the same seeded inputs, against the committed golden fixtures, and -- at
BASELINE sizes -- through size-independent properties.

Tolerance: north_star asks for 1e-5 (fp32) on returns / advantages / losses.
The scans accumulate in float64 on both sides, so they are held to 2 ulp of
float32 relative (2.4e-7) + 1e-6 absolute instead.
"""
import numpy as np
import pytest
import torch

from oracle import episode_info_oracle as eo
from oracle import ppo_loss_oracle as lo
from oracle import running_stats_oracle as rso

pytestmark = pytest.mark.gpu

RTOL, ATOL = 2.4e-7, 1e-6


@pytest.fixture(scope="module")
def K():
    from ppo_and_friends_amd import kernels
    assert torch.cuda.is_available(), "GPU tests need a device"
    kernels._lib.load()
    return kernels


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


# ---------------------------------------------------------------- K1 traj
def test_gae_traj_golden_g1(K, golden):
    """All 288 reference end_episode cases, batched as one ragged launch per parameter set."""
    g = golden("g1_end_episode")
    n = int(g["n_cases"][0])
    groups = {}
    for c in range(n):
        p = g[f"c{c}_params"]
        key = (p[0], p[1], bool(p[2]), None if np.isnan(p[3]) else (p[3], p[4]))
        groups.setdefault(key, []).append(c)
    for (gamma, lambd, use_gae, clip), cases in groups.items():
        rew = np.concatenate([g[f"c{c}_rewards"] for c in cases]).astype(np.float32)
        val = np.concatenate([g[f"c{c}_values"] for c in cases]).astype(np.float32)
        lens = np.array([len(g[f"c{c}_rewards"]) for c in cases], dtype=np.int32)
        starts = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int64)
        ev = np.array([g[f"c{c}_params"][5] for c in cases], dtype=np.float32)
        er = np.array([g[f"c{c}_params"][6] for c in cases], dtype=np.float32)
        adv, rtg = K.gae_rtg_traj(dev(rew), dev(val), dev(ev), dev(er), dev(starts), dev(lens),
                                  gamma, lambd, clip, use_gae)
        adv, rtg = adv.cpu().numpy(), rtg.cpu().numpy()
        for i, c in enumerate(cases):
            sl = slice(starts[i], starts[i] + lens[i])
            # rewards of the "uniform" cases are float64 in the fixture; the device
            # buffer holds float32, so compare against the oracle on the same float32 data
            a_ref, r_ref = eo.end_episode(rew[sl], val[sl], float(ev[i]), float(er[i]), gamma,
                                          lambd, clip, use_gae, "float64")
            np.testing.assert_allclose(adv[sl], a_ref.astype(np.float32), rtol=RTOL, atol=ATOL,
                                       err_msg=f"case {c} adv")
            np.testing.assert_allclose(rtg[sl], r_ref.astype(np.float32), rtol=RTOL, atol=ATOL,
                                       err_msg=f"case {c} rtg")
            if g[f"c{c}_rewards"].dtype == np.float64 and np.all(g[f"c{c}_rewards"] == 1.0):
                # constant-1 rewards are exact in float32: compare with the reference's own output
                if use_gae:
                    np.testing.assert_allclose(adv[sl], g[f"c{c}_adv"].astype(np.float32),
                                               rtol=RTOL, atol=ATOL)
                np.testing.assert_allclose(rtg[sl], g[f"c{c}_rtg_f64"].astype(np.float32),
                                           rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("lens", [[1], [63, 64, 65], [1, 2, 3, 500, 129, 128, 127, 1000]])
def test_gae_traj_ragged(K, lens):
    rng = np.random.default_rng(3)
    lens = np.array(lens, dtype=np.int32)
    N = int(lens.sum())
    rew = rng.uniform(-1, 1, N).astype(np.float32)
    val = rng.standard_normal(N).astype(np.float32)
    starts = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int64)
    ev = rng.standard_normal(len(lens)).astype(np.float32)
    er = (rng.standard_normal(len(lens)) * 80).astype(np.float32)
    adv, rtg = K.gae_rtg_traj(dev(rew), dev(val), dev(ev), dev(er), dev(starts), dev(lens))
    for i in range(len(lens)):
        sl = slice(starts[i], starts[i] + lens[i])
        a, r = eo.end_episode(rew[sl], val[sl], float(ev[i]), float(er[i]), 0.99, 0.95)
        np.testing.assert_allclose(adv[sl].cpu().numpy(), a.astype(np.float32), rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(rtg[sl].cpu().numpy(), r.astype(np.float32), rtol=RTOL, atol=ATOL)


def test_gae_traj_empty(K):
    z32 = torch.zeros(0, dtype=torch.float32, device="cuda")
    adv, rtg = K.gae_rtg_traj(z32, z32, z32, z32, torch.zeros(0, dtype=torch.int64, device="cuda"),
                              torch.zeros(0, dtype=torch.int32, device="cuda"))
    assert adv.numel() == 0


# ---------------------------------------------------------------- K1 tmajor
def _tmajor_oracle(rew, val, bv, br, ek, **kw):
    T, E = rew.shape
    adv = np.zeros((T, E), dtype=np.float32)
    rtg = np.zeros((T, E), dtype=np.float32)
    d = eo.rollout_to_dataset(rew, val, bv, br, ek, **kw)
    adv[d["flat_t"], d["flat_e"]] = d["adv"]
    rtg[d["flat_t"], d["flat_e"]] = d["rtg"]
    return adv, rtg


@pytest.mark.parametrize("T,E", [(1, 1), (5, 3), (16, 64), (17, 65), (128, 8), (128, 300), (200, 70), (33, 5000)])
@pytest.mark.parametrize("use_gae", [True, False])
def test_gae_tmajor_fixed_length(K, T, E, use_gae):
    rng = np.random.default_rng(T * 1000 + E)
    rew = rng.uniform(-1, 1, (T, E)).astype(np.float32)
    val = rng.standard_normal((T, E)).astype(np.float32)
    boot = (rng.standard_normal(E) * 60).astype(np.float32)
    ek = np.zeros((T, E), dtype=np.int8); ek[-1] = 2
    bv = np.zeros((T, E), dtype=np.float32); bv[-1] = boot
    a_ref, r_ref = _tmajor_oracle(rew, val, bv, bv, ek, use_gae=use_gae)
    adv, rtg = K.gae_rtg_tmajor(dev(rew), dev(val), dev(boot), dev(boot), None, use_gae=use_gae)
    np.testing.assert_allclose(adv.cpu().numpy(), a_ref, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(rtg.cpu().numpy(), r_ref, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("T,E,p", [(12, 3, 0.15), (128, 70, 0.02), (64, 257, 0.3), (130, 9, 0.05)])
@pytest.mark.parametrize("clip", [(-100.0, 100.0), (-0.5, 0.5), None, (0.25, 2.0)])     # the last excludes 0: terminal ends clip too
def test_gae_tmajor_with_episode_ends(K, T, E, p, clip):
    rng = np.random.default_rng(int(p * 100) + T)
    rew = rng.uniform(-1, 1, (T, E)).astype(np.float32)
    val = rng.standard_normal((T, E)).astype(np.float32)
    bv = (rng.standard_normal((T, E)) * 2).astype(np.float32)
    br = (rng.standard_normal((T, E)) * 2).astype(np.float32)
    u = rng.uniform(0, 1, (T, E))
    ek = np.where(u < p, 1, np.where(u < 2 * p, 2, 0)).astype(np.int8)
    ek[-1] = np.where(ek[-1] == 0, 2, ek[-1])
    a_ref, r_ref = _tmajor_oracle(rew, val, bv, br, ek, bootstrap_clip=clip)
    adv, rtg = K.gae_rtg_tmajor(dev(rew), dev(val), dev(bv), dev(br), dev(ek), bootstrap_clip=clip)
    np.testing.assert_allclose(adv.cpu().numpy(), a_ref, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(rtg.cpu().numpy(), r_ref, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_gae_tmajor_golden_g2(K, golden, tag):
    """The reference's own dataset (advantages / rtg in completion order) from the dense buffer."""
    g = golden("g2_dataset")
    pre = tag + "_"
    rew = g[pre + "in_rewards"].astype(np.float32)
    val = g[pre + "in_values"]
    boot = g[pre + "in_boot_v"]
    ek = g[pre + "in_end_kind"]
    adv, rtg = K.gae_rtg_tmajor(dev(rew), dev(val), dev(boot), dev(boot), dev(ek))
    d = eo.rollout_to_dataset(rew, val, boot, boot, ek)
    ft, fe = d["flat_t"], d["flat_e"]
    # float32 rewards on the device vs float64 rewards in the reference run: 1e-5 (north_star)
    np.testing.assert_allclose(adv.cpu().numpy()[ft, fe], g[pre + "adv"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(rtg.cpu().numpy()[ft, fe], g[pre + "rtg"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(adv.cpu().numpy()[ft, fe], d["adv"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("T,E,dense", [(9, (1 << 17) + 3, True), (20, 1 << 17, False),
                                        (6, 1 << 20, True), (11, (1 << 20) + 4, False), (3, (1 << 20) + 2, True)])
def test_gae_tmajor_streaming_kernels(K, T, E, dense):
    """Large-E launch paths (lane-per-column and 4-columns-per-lane streaming) against the C oracle."""
    from oracle import c_oracle
    rng = np.random.default_rng(E % 1000 + T)
    rew = rng.uniform(-1, 1, (T, E)).astype(np.float32)
    val = rng.standard_normal((T, E)).astype(np.float32)
    if dense:
        bv = (rng.standard_normal((T, E)) * 2).astype(np.float32)
        br = (rng.standard_normal((T, E)) * 90).astype(np.float32)
        u = rng.uniform(0, 1, (T, E))
        ek = np.where(u < 0.1, 1, np.where(u < 0.2, 2, 0)).astype(np.int8)
        ek[-1] = np.where(ek[-1] == 0, 2, ek[-1])
        adv, rtg = K.gae_rtg_tmajor(dev(rew), dev(val), dev(bv), dev(br), dev(ek))
        if T == 9:                       # once more with a clip range that excludes zero (terminal ends clip too)
            a2, r2 = K.gae_rtg_tmajor(dev(rew), dev(val), dev(bv), dev(br), dev(ek), bootstrap_clip=(0.5, 3.0))
            a2_ref, r2_ref = c_oracle.gae_rtg_tmajor(rew, val, bv, br, ek, clip=(0.5, 3.0))
            np.testing.assert_allclose(a2.cpu().numpy(), a2_ref, rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(r2.cpu().numpy(), r2_ref, rtol=RTOL, atol=ATOL)
    else:
        boot = (rng.standard_normal(E) * 2).astype(np.float32)
        bv = np.zeros((T, E), dtype=np.float32); bv[-1] = boot
        br = bv
        ek = np.zeros((T, E), dtype=np.int8); ek[-1] = 2
        adv, rtg = K.gae_rtg_tmajor(dev(rew), dev(val), dev(boot), dev(boot), None)
    a_ref, r_ref = c_oracle.gae_rtg_tmajor(rew, val, bv, br, ek)
    np.testing.assert_allclose(adv.cpu().numpy(), a_ref, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(rtg.cpu().numpy(), r_ref, rtol=RTOL, atol=ATOL)


def test_gae_tmajor_full_size_properties(K):
    """C2 size (E=4096, T=128): linearity in (rewards, bootstrap) and the closed form for constant inputs."""
    T, E = 128, 4096
    gen = torch.Generator(device="cuda").manual_seed(5)
    r1 = torch.rand(T, E, device="cuda", generator=gen) * 2 - 1
    r2 = torch.rand(T, E, device="cuda", generator=gen) * 2 - 1
    v1 = torch.randn(T, E, device="cuda", generator=gen)
    v2 = torch.randn(T, E, device="cuda", generator=gen)
    b1 = torch.randn(E, device="cuda", generator=gen)
    b2 = torch.randn(E, device="cuda", generator=gen)
    a1, g1 = K.gae_rtg_tmajor(r1, v1, b1, b1, None, bootstrap_clip=None)
    a2, g2 = K.gae_rtg_tmajor(r2, v2, b2, b2, None, bootstrap_clip=None)
    a3, g3 = K.gae_rtg_tmajor(r1 + r2, v1 + v2, b1 + b2, b1 + b2, None, bootstrap_clip=None)
    torch.testing.assert_close(a3, a1 + a2, rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(g3, g1 + g2, rtol=1e-5, atol=2e-5)
    # constant reward 1, zero values/bootstrap: rtg_t = (1 - gamma^(T-t)) / (1 - gamma)
    ones = torch.ones(T, E, device="cuda")
    zeros = torch.zeros(T, E, device="cuda")
    zb = torch.zeros(E, device="cuda")
    a, r = K.gae_rtg_tmajor(ones, zeros, zb, zb, None)
    k = torch.arange(T, 0, -1, device="cuda", dtype=torch.float64)
    closed = ((1 - 0.99 ** k) / (1 - 0.99)).to(torch.float32)[:, None].expand(T, E)
    torch.testing.assert_close(r, closed, rtol=1e-6, atol=1e-5)
    # one env column against the oracle
    e = 1234
    a_ref, r_ref = eo.end_episode(r1[:, e].cpu().numpy(), v1[:, e].cpu().numpy(),
                                  float(b1[e]), float(b1[e]), 0.99, 0.95, None)
    np.testing.assert_allclose(a1[:, e].cpu().numpy(), a_ref.astype(np.float32), rtol=RTOL, atol=ATOL)


# ---------------------------------------------------------------- K2+K3
@pytest.mark.parametrize("B", [2, 63, 256, 768, 5000])
@pytest.mark.parametrize("cfg", [dict(), dict(normalize_adv=False), dict(use_huber=True),
                                 dict(entropy_weight=0.0), dict(kl_loss_weight=0.5, surr_clip=0.1)])
def test_ppo_loss_matches_torch(K, B, cfg):
    torch.manual_seed(B)
    old_lp = -torch.rand(B) * 2
    cur_lp = (old_lp + torch.randn(B) * 0.3).requires_grad_()
    adv = torch.randn(B) * 3 + 1
    ent = (torch.rand(B) * 0.7).requires_grad_()
    val = (torch.randn(B) * 12).requires_grad_()       # |diff| crosses the Huber delta of 10
    rtg = torch.randn(B)
    kw = dict(normalize_adv=True, surr_clip=0.2, entropy_weight=0.01, kl_loss_weight=0.0, use_huber=False)
    kw.update(cfg)
    ref = lo.ppo_minibatch_losses(cur_lp, old_lp, adv, ent, val, rtg, **kw)
    ref["actor_loss"].backward()
    ref["critic_loss"].backward()
    sc, dlp, dent, dval = K.ppo_loss_fwd_bwd(dev(cur_lp.detach()), dev(old_lp), dev(adv), dev(ent.detach()),
                                             dev(val.detach()), dev(rtg), **kw)
    sc = sc.cpu().numpy()
    tol = dict(rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sc[K.SC_SURR], ref["surr"], **tol)
    np.testing.assert_allclose(sc[K.SC_ACTOR], ref["actor"], **tol)
    np.testing.assert_allclose(sc[K.SC_CRITIC], ref["critic"], **tol)
    if kw["entropy_weight"] != 0.0:      # the reference only tallies entropy when it is weighted (ppo.py:2395-2396)
        np.testing.assert_allclose(sc[K.SC_ENTROPY], ref["entropy"], **tol)
    np.testing.assert_allclose(sc[K.SC_KL], ref["kl"], **tol)
    if kw["normalize_adv"]:
        np.testing.assert_allclose(sc[K.SC_ADV_MEAN], ref["adv_mean"], **tol)
        np.testing.assert_allclose(sc[K.SC_ADV_STD], ref["adv_std"], **tol)
    assert sc[K.SC_BAD] == 0.0
    np.testing.assert_allclose(dlp.cpu().numpy(), cur_lp.grad.numpy(), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(dval.cpu().numpy(), val.grad.numpy(), rtol=1e-5, atol=1e-8)
    exp_dent = ent.grad.numpy() if ent.grad is not None else np.zeros(B, dtype=np.float32)
    np.testing.assert_allclose(dent.cpu().numpy(), exp_dent, rtol=1e-5, atol=1e-9)


def test_ppo_loss_flags_nonfinite_ratio(K):
    B = 64
    old_lp = torch.zeros(B); cur_lp = torch.zeros(B); cur_lp[7] = 200.0      # exp -> inf
    z = torch.zeros(B)
    sc, *_ = K.ppo_loss_fwd_bwd(dev(cur_lp), dev(old_lp), dev(torch.randn(B)), dev(z), dev(z), dev(z))
    assert sc[K.SC_BAD].item() == 1.0


# ---------------------------------------------------------------- K4
def test_minibatch_gather_all_fields(K):
    rng = np.random.default_rng(0)
    N, B = 1000, 256
    obs = rng.standard_normal((N, 4)).astype(np.float32)
    cobs = rng.standard_normal((N, 54)).astype(np.float32)
    act = rng.integers(0, 5, (N, 1)).astype(np.int64)
    adv = rng.standard_normal(N).astype(np.float32)
    perm = rng.permutation(N)[:B].astype(np.int64)
    row_map = rng.permutation(N).astype(np.int32)
    srcs = [dev(obs), dev(cobs), dev(act), dev(adv)]
    for rm in (None, row_map):
        dsts = [torch.zeros(B, 4, device="cuda"), torch.zeros(B, 54, device="cuda"),
                torch.zeros(B, 1, dtype=torch.int64, device="cuda"), torch.zeros(B, device="cuda")]
        K.minibatch_gather(list(zip(srcs, dsts)), dev(perm), None if rm is None else dev(rm))
        rows = perm if rm is None else rm[perm]
        for s, d in zip((obs, cobs, act, adv), dsts):
            np.testing.assert_array_equal(d.cpu().numpy(), s[rows])        # bit-exact
    # scatter is the inverse on the touched rows
    dst = torch.zeros(N, device="cuda")
    vals = torch.arange(B, dtype=torch.float32, device="cuda")
    K.scatter_rows_f32(vals, dev(perm), dst, dev(row_map))
    exp = np.zeros(N, dtype=np.float32); exp[row_map[perm]] = np.arange(B)
    np.testing.assert_array_equal(dst.cpu().numpy(), exp)


# ---------------------------------------------------------------- K5
def test_running_moments_golden_g4(K, golden):
    g = golden("g4_running_stats")
    mean = torch.zeros(1, device="cuda"); var = torch.ones(1, device="cuda")
    count = torch.full((1,), 1e-4, dtype=torch.float64, device="cuda")
    for i in range(4):
        m = K.batch_moments(dev(g[f"s_batch{i}"]), 1)
        K.running_moments_integrate(m, mean, var, count)
        exp = g[f"s_state{i}"]
        np.testing.assert_allclose(mean.item(), exp[0], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(var.item(), exp[1], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(count.item(), exp[2], rtol=1e-15)
    W = 6
    mean = torch.zeros(W, device="cuda"); var = torch.ones(W, device="cuda")
    count = torch.full((1,), 1e-4, dtype=torch.float64, device="cuda")
    for i in range(3):
        m = K.batch_moments(dev(g[f"v_batch{i}"]), W)
        K.running_moments_integrate(m, mean, var, count)
        np.testing.assert_allclose(mean.cpu().numpy(), g[f"v_mean{i}"], rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(var.cpu().numpy(), g[f"v_var{i}"], rtol=2e-6, atol=1e-7)


def test_running_moments_multi_rank_merge_equals_concatenation(K):
    """R per-rank records merged on the device == the reference's allgather + np.mean/np.var."""
    rng = np.random.default_rng(9)
    parts = [rng.standard_normal(256).astype(np.float32) * (r + 1) + r for r in range(4)]
    recs = torch.cat([K.batch_moments(dev(p), 1) for p in parts])
    mean = torch.zeros(1, device="cuda"); var = torch.ones(1, device="cuda")
    count = torch.full((1,), 1e-4, dtype=torch.float64, device="cuda")
    K.running_moments_integrate(recs, mean, var, count)
    rs = rso.RunningMeanStd()
    rs.update(None, gathered=parts)
    np.testing.assert_allclose(mean.item(), float(rs.mean), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(var.item(), float(rs.variance), rtol=2e-6, atol=1e-7)
    assert count.item() == rs.count


def test_normalize_denormalize(K):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((300, 5)).astype(np.float32) * 4
    mean = rng.standard_normal(5).astype(np.float32); var = rng.uniform(0.1, 3, 5).astype(np.float32)
    y = K.normalize(dev(x), dev(mean), dev(var)).cpu().numpy()
    np.testing.assert_allclose(y, rso.normalize(x, mean, var), rtol=1e-6, atol=1e-6)
    z = K.denormalize(dev(y), dev(mean), dev(var)).cpu().numpy()
    np.testing.assert_allclose(z, x, rtol=1e-5, atol=1e-5)
    yc = K.normalize(dev(x), dev(mean), dev(var), clip=(-1.0, 1.0)).cpu().numpy()
    np.testing.assert_allclose(yc, np.clip(rso.normalize(x, mean, var), -1, 1), rtol=1e-6, atol=1e-6)


# ---------------------------------------------------------------- K6
@pytest.mark.parametrize("n,Kc", [(1, 2), (256, 2), (3072, 5), (100, 17)])
def test_categorical_eval_fwd_bwd(K, n, Kc):
    torch.manual_seed(n + Kc)
    logits = (torch.randn(n, Kc) * 3).requires_grad_()
    if n > 4:
        with torch.no_grad():
            logits[3, 0] = 40.0          # saturated row: probs clamp at 1 - eps / eps
    actions = torch.randint(0, Kc, (n,))
    logp_ref, ent_ref, probs_ref = lo.categorical_logp_entropy(logits, actions)
    g_lp = torch.randn(n); g_ent = torch.randn(n)
    (logp_ref * g_lp + ent_ref * g_ent).sum().backward()
    logp, ent, probs = K.categorical_eval_fwd(dev(logits.detach()), dev(actions))
    np.testing.assert_allclose(logp.cpu().numpy(), logp_ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ent.cpu().numpy(), ent_ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(probs.cpu().numpy(), probs_ref.detach().numpy(), rtol=1e-5, atol=1e-7)
    dl = K.categorical_eval_bwd(probs, dev(actions), dev(g_lp), dev(g_ent))
    np.testing.assert_allclose(dl.cpu().numpy(), logits.grad.numpy(), rtol=2e-4, atol=2e-6)


def test_categorical_sample_distribution_and_logp(K):
    n, Kc = 200000, 5
    torch.manual_seed(0)
    row = torch.randn(Kc)
    logits = row.repeat(n, 1).cuda().contiguous()
    a, lp = K.categorical_sample(logits, seed=123, offset=0)
    p = torch.softmax(row, 0).numpy()
    freq = np.bincount(a.cpu().numpy(), minlength=Kc) / n
    np.testing.assert_allclose(freq, p, atol=5e-3)
    np.testing.assert_allclose(lp.cpu().numpy(), np.log(p)[a.cpu().numpy()], rtol=1e-5, atol=1e-6)
    # counter-based: same (seed, offset) -> same draws; different offset -> different draws
    a2, _ = K.categorical_sample(logits, seed=123, offset=0)
    a3, _ = K.categorical_sample(logits, seed=123, offset=n)
    assert torch.equal(a, a2) and not torch.equal(a, a3)


@pytest.mark.parametrize("n,D", [(1, 1), (256, 6), (2048, 6), (77, 3)])
def test_gaussian_tanh_eval_fwd_bwd(K, n, D):
    torch.manual_seed(n * 7 + D)
    mean = torch.randn(n, D).requires_grad_()
    log_std = (torch.randn(D) * 0.5 - 0.5).requires_grad_()
    with torch.no_grad():
        log_std[0] = -8.0                 # softplus < min_std: the max() floor is active
    x = torch.randn(n, D) * 1.5
    if n > 2:
        x[1, 0] = 12.0                     # tanh' underflows -> clamp(1e-6)
        x[2, 0] = 60.0 if D > 0 else 0.0   # log-prob clamp at -100 with tiny std
    lp_ref = lo.gaussian_tanh_logp(mean, log_std, x)
    ent_ref = -lo.gaussian_tanh_logp(mean, log_std, mean)     # entropy := -log_prob of the MEAN (ppo_policy.py:950; pinned by g8)
    g_lp = torch.randn(n); g_ent = torch.randn(n)
    (lp_ref * g_lp + ent_ref * g_ent).sum().backward()
    lp, ent = K.gaussian_tanh_eval_fwd(dev(mean.detach()), dev(log_std.detach()), dev(x))
    np.testing.assert_allclose(lp.cpu().numpy(), lp_ref.detach().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), ent_ref.detach().numpy(), rtol=2e-5, atol=2e-5)
    dm, dls = K.gaussian_tanh_eval_bwd(dev(mean.detach()), dev(log_std.detach()), dev(x), dev(g_lp), dev(g_ent))
    np.testing.assert_allclose(dm.cpu().numpy(), mean.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dls.cpu().numpy(), log_std.grad.numpy(), rtol=1e-3, atol=1e-3)


def test_gaussian_sample_moments_and_logp(K):
    n, D = 100000, 6
    mean = torch.linspace(-1, 1, D).repeat(n, 1).cuda().contiguous()
    log_std = torch.full((D,), -0.5, device="cuda")
    lo_v = torch.tensor([-2.0, -1.0, 0.0, -3.0, -2.0, -0.5], device="cuda")
    hi_v = torch.tensor([4.0, 1.0, 5.0, 3.0, 2.5, 0.5], device="cuda")                      # bounds per action dimension
    raw, act, lp = K.gaussian_tanh_sample(mean, log_std, seed=7, offset=0, act_lo=lo_v, act_hi=hi_v)
    sd = torch.nn.functional.softplus(torch.tensor(-0.5)).item()
    np.testing.assert_allclose(raw.mean(0).cpu().numpy(), np.linspace(-1, 1, D), atol=0.01)
    np.testing.assert_allclose(raw.std(0).cpu().numpy(), sd, atol=0.01)
    ref_act = lo.gaussian_refine(raw.cpu(), lo_v.cpu().numpy(), hi_v.cpu().numpy())
    np.testing.assert_allclose(act.cpu().numpy(), ref_act.numpy(), rtol=1e-5, atol=1e-5)
    ref_lp = lo.gaussian_tanh_logp(mean.cpu(), log_std.cpu(), raw.cpu())
    np.testing.assert_allclose(lp.cpu().numpy(), ref_lp.numpy(), rtol=2e-5, atol=2e-5)


# ---------------------------------------------------------------- K11
@pytest.mark.parametrize("max_norm,grad_scale", [(0.5, 1.0), (None, 1.0), (0.5, 0.125)])
def test_clip_adam_matches_torch(K, max_norm, grad_scale):
    torch.manual_seed(1)
    shapes = [(128, 4), (128,), (128, 128), (128,), (2, 128), (2,)]
    params = [torch.randn(s) * 0.3 for s in shapes]
    steps = 5
    grads = [[torch.randn(s) * (3.0 if k % 2 else 0.05) for s in shapes] for k in range(steps)]
    ref, norms = lo.clip_adam_reference(params, grads, steps, lr=3e-4, eps=1e-5,
                                        max_norm=max_norm, grad_scale=grad_scale)
    flat = torch.cat([p.flatten() for p in params]).cuda()
    m = torch.zeros_like(flat); v = torch.zeros_like(flat)
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    lr = torch.full((1,), 3e-4, device="cuda")
    scratch = torch.zeros(K.NORM_SCRATCH_DOUBLES, dtype=torch.float64, device="cuda")
    gn = torch.zeros(1, device="cuda")
    for k in range(steps):
        g = torch.cat([x.flatten() for x in grads[k]]).cuda()
        K.clip_adam_step(flat, g, m, v, step, lr, scratch, grad_scale=grad_scale,
                         max_norm=max_norm, grad_norm_out=gn)
        if max_norm is not None:
            np.testing.assert_allclose(gn.item(), norms[k], rtol=1e-5)
    assert step.item() == steps
    ref_flat = torch.cat([p.flatten() for p in ref]).numpy()
    np.testing.assert_allclose(flat.cpu().numpy(), ref_flat, rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------- K8
@pytest.mark.parametrize("n,D", [(1, 128), (256, 128), (2048, 128), (77, 40)])
def test_icm_forward_loss_fwd_bwd(K, n, D):
    """icm.py:421-430: intrinsic reward, 0.5 * mean squared error, and its gradient."""
    torch.manual_seed(n + D)
    pred = torch.randn(n, D, requires_grad=True)
    enc2 = torch.randn(n, D, requires_grad=True)
    f = torch.nn.MSELoss(reduction="none")(pred, enc2)
    intr_ref = (0.01 / 2.0) * f.sum(dim=-1)
    loss_ref = 0.5 * f.mean()
    (loss_ref * 3.0).backward()
    intr, loss = K.icm_forward_loss_fwd(dev(pred.detach()), dev(enc2.detach()), 0.01)
    np.testing.assert_allclose(intr.cpu().numpy(), intr_ref.detach().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=1e-5)
    g = torch.full((1,), 3.0, device="cuda")
    dp, de = K.icm_forward_loss_bwd(dev(pred.detach()), dev(enc2.detach()), g)
    np.testing.assert_allclose(dp.cpu().numpy(), pred.grad.numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(de.cpu().numpy(), enc2.grad.numpy(), rtol=1e-5, atol=1e-9)


# ---------------------------------------------------------------- K9
def _attention_ref(q, k, v, masked):
    """attention.py:94-103 on torch-CPU."""
    L, D = q.shape[-2], q.shape[-1]
    att = (q @ k.transpose(-2, -1)) * (1.0 / np.sqrt(D))
    if masked:
        att = att.masked_fill(torch.tril(torch.ones(L, L)) == 0, float("-inf"))
    return torch.softmax(att, dim=-1) @ v


@pytest.mark.parametrize("n_seq,L,D,masked", [(1, 3, 64, False), (4, 3, 64, True), (1024, 3, 64, True),
                                              (7, 5, 32, False), (33, 16, 64, True), (10, 1, 16, False),
                                              (257, 4, 128, True)])
def test_mat_attention_core_fwd_bwd(K, n_seq, L, D, masked):
    torch.manual_seed(n_seq + L + D)
    q = torch.randn(n_seq, L, D, requires_grad=True)
    k = torch.randn(n_seq, L, D, requires_grad=True)
    v = torch.randn(n_seq, L, D, requires_grad=True)
    y_ref = _attention_ref(q, k, v, masked)
    g = torch.randn_like(y_ref)
    (y_ref * g).sum().backward()
    y, probs = K.mat_attention_fwd(dev(q.detach()), dev(k.detach()), dev(v.detach()), masked)
    np.testing.assert_allclose(y.cpu().numpy(), y_ref.detach().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(probs.sum(-1).cpu().numpy(), 1.0, rtol=1e-5)
    dq, dk, dv = K.mat_attention_bwd(dev(q.detach()), dev(k.detach()), dev(v.detach()), probs, dev(g))
    np.testing.assert_allclose(dq.cpu().numpy(), q.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dk.cpu().numpy(), k.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dv.cpu().numpy(), v.grad.numpy(), rtol=2e-4, atol=2e-5)


def test_attention_blocks_match_reference_golden_g5(golden):
    """SelfAttention / encoder / decoder blocks with the reference's seeded weights -> its recorded outputs."""
    from ppo_and_friends_amd.networks import attention as at
    g = golden("g5_attention")
    x = dev(g["x"]); rep = dev(g["rep"])

    def load(mod, prefix):
        sd = {k: torch.tensor(g[prefix + k]) for k in mod.state_dict() if k != "mask" and not k.endswith(".mask")}
        mod.load_state_dict(sd, strict=False)
        return mod.cuda()

    for tag, masked in (("u", False), ("m", True)):
        sa = load(at.SelfAttention(64, 1, 3, masked=masked), f"sa_{tag}_")
        np.testing.assert_allclose(sa(x, x, x).detach().cpu().numpy(), g[f"sa_{tag}_y"], rtol=1e-5, atol=1e-5)
    enc = load(at.SelfAttentionEncodingBlock(64, 1, 3), "enc_")
    np.testing.assert_allclose(enc(x).detach().cpu().numpy(), g["enc_y"], rtol=1e-5, atol=2e-5)
    dec = load(at.SelfAttentionDecodingBlock(64, 1, 3), "dec_")
    np.testing.assert_allclose(dec(x, rep).detach().cpu().numpy(), g["dec_y"], rtol=1e-5, atol=2e-5)


def test_c_abi_collectives_single_rank(K):
    """
    ppoaf_comm_* / allreduce_avg / bcast / allgather_moments (utils/mpi_utils.py:50-111,
    utils/stats.py:47-50) with a one-rank communicator -- what a one-GPU box can host: RCCL is bound at run
    time, the calls run on the caller's stream and are the identity for world = 1, as the reference's
    collectives are for num_procs == 1.
    """
    import ctypes as C
    from ppo_and_friends_amd import _lib
    lib = _lib.load()
    uid = C.create_string_buffer(128)
    _lib.check(lib.ppoaf_comm_unique_id(uid), "comm_unique_id")
    comm = C.c_void_p()
    _lib.check(lib.ppoaf_comm_init(0, 1, uid, C.byref(comm)), "comm_init")
    st = K.stream()
    x = torch.randn(1000, device="cuda")
    want = x.clone()
    _lib.check(lib.ppoaf_allreduce_avg_f32(comm, x.data_ptr(), x.numel(), st), "allreduce_avg_f32")
    _lib.check(lib.ppoaf_bcast_f32(comm, x.data_ptr(), x.numel(), 0, st), "bcast_f32")
    rec = torch.tensor([256.0, 0.25, 17.5], dtype=torch.float64, device="cuda")
    out = torch.zeros(1, 3, dtype=torch.float64, device="cuda")
    _lib.check(lib.ppoaf_allgather_moments(comm, rec.data_ptr(), 3, out.data_ptr(), st), "allgather_moments")
    torch.cuda.synchronize()
    assert torch.equal(x, want) and torch.equal(out[0], rec)
    assert lib.ppoaf_bcast_f32(comm, x.data_ptr(), x.numel(), 3, st) != 0          # root outside the communicator
    assert b"root" in lib.ppoaf_last_error()
    _lib.check(lib.ppoaf_comm_destroy(comm), "comm_destroy")


# ---------------------------------------------------------------- K1 fuzz
def test_gae_kernels_fuzz_against_the_oracle(K):
    """
    Randomised shapes and episode structures (hypothesis, derandomised): the dense time-major form -- any T, E,
    density of terminal / bootstrapped ends, clip range, (gamma, lambda), GAE or rtg - V -- and the ragged
    trajectory form incl. zero-length trajectories, both against oracle/episode_info_oracle.py.
    """
    from hypothesis import given, settings, strategies as st, HealthCheck

    params = st.sampled_from([(0.99, 0.95), (1.0, 1.0), (0.9, 0.0), (0.5, 0.99)])
    clips = st.sampled_from([(-100.0, 100.0), (-0.25, 0.75), None, (0.1, 1.5)])

    @settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(T=st.integers(1, 300), E=st.integers(1, 150), p_term=st.floats(0.0, 0.4), p_boot=st.floats(0.0, 0.4),
           gl=params, clip=clips, use_gae=st.booleans(), seed=st.integers(0, 10_000))
    def dense(T, E, p_term, p_boot, gl, clip, use_gae, seed):
        rng = np.random.default_rng(seed)
        rew = rng.uniform(-1, 1, (T, E)).astype(np.float32)
        val = rng.standard_normal((T, E)).astype(np.float32)
        bv = (rng.standard_normal((T, E)) * 2).astype(np.float32)
        br = (rng.standard_normal((T, E)) * 3).astype(np.float32)
        u = rng.uniform(0, 1, (T, E))
        ek = np.where(u < p_term, 1, np.where(u < p_term + p_boot, 2, 0)).astype(np.int8)
        ek[-1] = np.where(ek[-1] == 0, 2, ek[-1])
        kw = dict(gamma=gl[0], lambd=gl[1], bootstrap_clip=clip, use_gae=use_gae)
        a_ref, r_ref = _tmajor_oracle(rew, val, bv, br, ek, **kw)
        adv, rtg = K.gae_rtg_tmajor(dev(rew), dev(val), dev(bv), dev(br), dev(ek), **kw)
        np.testing.assert_allclose(adv.cpu().numpy(), a_ref, rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(rtg.cpu().numpy(), r_ref, rtol=RTOL, atol=ATOL)

    @settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(lens=st.lists(st.integers(0, 260), min_size=1, max_size=12), gl=params, clip=clips, use_gae=st.booleans(),
           seed=st.integers(0, 10_000))
    def ragged(lens, gl, clip, use_gae, seed):
        rng = np.random.default_rng(seed)
        lens = np.array(lens, dtype=np.int32)
        N = int(lens.sum())
        rew = rng.uniform(-1, 1, N).astype(np.float32)
        val = rng.standard_normal(N).astype(np.float32)
        starts = np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.int64)
        ev = rng.standard_normal(len(lens)).astype(np.float32)
        er = (rng.standard_normal(len(lens)) * 3).astype(np.float32)
        kw = dict(gamma=gl[0], lambd=gl[1], bootstrap_clip=clip, use_gae=use_gae)
        adv, rtg = K.gae_rtg_traj(dev(rew), dev(val), dev(ev), dev(er), dev(starts), dev(lens), **kw)
        adv, rtg = adv.cpu().numpy(), rtg.cpu().numpy()
        for i in range(len(lens)):
            if lens[i] == 0:
                continue
            sl = slice(starts[i], starts[i] + lens[i])
            a, r = eo.end_episode(rew[sl], val[sl], float(ev[i]), float(er[i]), gl[0], gl[1], clip, use_gae)
            np.testing.assert_allclose(adv[sl], a.astype(np.float32), rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(rtg[sl], r.astype(np.float32), rtol=RTOL, atol=ATOL)

    dense()
    ragged()


def test_episode_info_drop_in_matches_golden_g1(K, golden):
    """
    utils/episode_info.py:EpisodeInfo -- the reference's per-trajectory class (episode_info.py:169-482: same
    constructor, add_info / end_episode / compute_advantages), host-staged lists with the scans on the GPU -- driven
    as policies/ppo_policy.py:638-651,684-691 drive it, against the reference's own recorded outputs (golden g1),
    incl. staged LSTM states and the recalculation after a value update.
    """
    from ppo_and_friends_amd.utils.episode_info import EpisodeInfo
    g = golden("g1_end_episode")
    n = int(g["n_cases"][0])
    checked = 0
    for c in range(n):
        p = g[f"c{c}_params"]
        clip = None if np.isnan(p[3]) else (float(p[3]), float(p[4]))
        rew, val = g[f"c{c}_rewards"], g[f"c{c}_values"]
        ep = EpisodeInfo(starting_ts=3, use_gae=bool(p[2]), gamma=float(p[0]), lambd=float(p[1]), bootstrap_clip=clip)
        L = len(rew)
        for t in range(L):
            h = torch.full((1, 4), float(t))
            ep.add_info(observation=np.zeros(2), next_observation=np.ones(2), raw_action=np.zeros(1), action=np.zeros(1),
                        value=float(val[t]), log_prob=-0.5, reward=float(rew[t]), critic_observation=np.zeros(2),
                        actor_hidden=h, actor_cell=h + 1, critic_hidden=h + 2, critic_cell=h + 3)
        ep.end_episode(ending_ts=3 + L, terminal=False, ending_value=float(p[5]), ending_reward=float(p[6]))
        assert ep.is_finished and ep.length == L and ep.has_hidden_states and len(ep.critic_cell) == L
        assert ep.values.dtype == np.float32
        # float32 rewards on the device vs the reference's float64 lists: north_star tolerance
        np.testing.assert_allclose(ep.rewards_to_go, g[f"c{c}_rtg_f64"], rtol=1e-5, atol=1e-5, err_msg=f"case {c}")
        if bool(p[2]):
            np.testing.assert_allclose(ep.advantages, g[f"c{c}_adv"], rtol=1e-5, atol=1e-5, err_msg=f"case {c}")
        before = ep.advantages.copy()
        ep.compute_advantages()                                    # recalculation keeps the clipped ending reward
        np.testing.assert_array_equal(ep.advantages, before)
        checked += 1
    assert checked == n and n > 10
    with pytest.raises(ValueError):
        EpisodeInfo().add_info(np.zeros(1), np.zeros(1), 0, 0, 0.0, 0.0, 0.0, actor_hidden=torch.zeros(1, 2))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_dataset_from_episode_list_matches_golden_g2(K, golden, tag):
    """
    The reference's own way of filling a PPODataset (episode_info.py:689-719, 745-914): finished EpisodeInfo objects
    added one by one in completion order, build(), 13-tuple items, values write-back + recalculate_advantages --
    driven from the raw tables of golden g2 and compared with the tensors the unmodified reference built from them.
    """
    from ppo_and_friends_amd.utils.episode_info import EpisodeInfo, PPODataset
    g = golden("g2_dataset")
    pre = tag + "_"
    rew, val, boot, ek = g[pre + "in_rewards"], g[pre + "in_values"], g[pre + "in_boot_v"], g[pre + "in_end_kind"]
    obs, logp, act = g[pre + "in_obs"], g[pre + "in_logp"], g[pre + "in_actions"]
    ds = PPODataset(device="cuda", action_dtype="discrete")
    for (e, t0, t1, kind) in eo.segments_from_end_kind(ek):
        ep = EpisodeInfo(starting_ts=t0, use_gae=True, gamma=0.99, lambd=0.95, bootstrap_clip=(-100.0, 100.0))
        for t in range(t0, t1 + 1):
            ep.add_info(observation=obs[t, e], next_observation=obs[t + 1, e], raw_action=act[t, e], action=act[t, e],
                        value=float(val[t, e]), log_prob=float(logp[t, e]), reward=float(rew[t, e]),
                        critic_observation=obs[t, e])
        ev = 0.0 if kind == 1 else float(boot[t1, e])
        ep.end_episode(ending_ts=t1 + 1, terminal=kind == 1, ending_value=ev, ending_reward=ev)
        ds.add_episode(ep)
    ds.build()
    assert len(ds) == int(g[pre + "len"][0])
    np.testing.assert_array_equal(ds.ep_lens.cpu().numpy(), g[pre + "ep_lens"])
    np.testing.assert_array_equal(ds.observations.cpu().numpy(), g[pre + "obs"])
    np.testing.assert_array_equal(ds.next_observations.cpu().numpy(), g[pre + "next_obs"])
    np.testing.assert_array_equal(ds.actions.cpu().numpy(), g[pre + "actions"])
    np.testing.assert_allclose(ds.log_probs.cpu().numpy(), g[pre + "logp"], rtol=1e-6)
    np.testing.assert_array_equal(ds.values[torch.arange(len(ds))].cpu().numpy(), g[pre + "values"])
    tol = dict(rtol=1e-5, atol=1e-5)                     # float32 rewards on the device, float64 lists in the reference
    np.testing.assert_allclose(ds.advantages.cpu().numpy(), g[pre + "adv"], **tol)
    np.testing.assert_allclose(ds.rewards_to_go.cpu().numpy(), g[pre + "rtg"], **tol)
    idx = int(g[pre + "item_idx"][0])
    item = ds[idx]
    assert len(item) == 13 and item[12] == idx
    np.testing.assert_array_equal(item[1].cpu().numpy(), g[pre + "item_obs"])
    np.testing.assert_allclose(float(item[5]), g[pre + "item_adv"][0], **tol)
    ds.values[torch.arange(len(ds), device="cuda")] = dev(g[pre + "new_values"])
    ds.recalculate_advantages()
    np.testing.assert_allclose(ds.advantages.cpu().numpy(), g[pre + "adv_recalc"], **tol)
    with pytest.raises(RuntimeError):
        ds.build()


def test_shared_episode_dataset_matches_golden_g3(K, golden):
    """
    AgentSharedEpisode / PPOSharedEpisodeDataset (episode_info.py:485-644, 990-1084) filled the reference's way --
    every agent's finished EpisodeInfo handed over with its env index -- against the tensors the unmodified
    reference built (golden g3): rows of [A, .], env-major, agents in `agent_ids` order.
    """
    from ppo_and_friends_amd.utils.episode_info import EpisodeInfo, PPOSharedEpisodeDataset
    g = golden("g3_shared")
    obs, rew, val = g["in_obs"], g["in_rewards"], g["in_values"]
    boot, logp, act = g["in_boot"], g["in_logp"], g["in_actions"]
    T, E, A = rew.shape
    agent_ids = np.array(["a0", "a1", "a2"])
    ds = PPOSharedEpisodeDataset(E, agent_ids, device="cuda", action_dtype="discrete")
    for e in range(E):
        for a in (2, 0, 1):                                   # delivery order within an env does not matter
            ep = EpisodeInfo(starting_ts=0, use_gae=True, gamma=0.99, lambd=0.95, bootstrap_clip=(-100.0, 100.0))
            for t in range(T):
                ep.add_info(observation=obs[t, e, a], next_observation=obs[t + 1, e, a], raw_action=act[t, e, a],
                            action=act[t, e, a], value=float(val[t, e, a]), log_prob=float(logp[t, e, a]),
                            reward=float(rew[t, e, a]), critic_observation=obs[t, e, a])
            ep.end_episode(ending_ts=T, terminal=False, ending_value=float(boot[e, a]), ending_reward=float(boot[e, a]))
            ds.add_shared_episode(ep, agent_ids[a], e)
    ds.build()
    assert len(ds) == int(g["len"][0]) == E * T
    np.testing.assert_array_equal(ds.observations.cpu().numpy(), g["obs"])
    np.testing.assert_array_equal(ds.actions.cpu().numpy(), g["actions"])
    np.testing.assert_array_equal(ds.values[torch.arange(E * T)].cpu().numpy(), g["values"])
    np.testing.assert_allclose(ds.log_probs.cpu().numpy(), g["logp"], rtol=1e-6)
    np.testing.assert_allclose(ds.advantages.cpu().numpy(), g["adv"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ds.rewards_to_go.cpu().numpy(), g["rtg"], rtol=1e-5, atol=1e-5)
    item = ds[5]
    assert len(item) == 13 and item[1].shape == (A, obs.shape[-1]) and item[12] == 5


def test_multi_categorical_and_bernoulli_distributions_match_torch(K):
    """
    networks/distributions.py:134-196 (Bernoulli, MultiBinary spaces) and :272-438 (MultiCategorical, MultiDiscrete
    spaces; per-slice softmax of the actor output :1046-1056): log-probs, entropies, their gradients and the
    deterministic refinement against torch.distributions on the CPU; samples are in range and their logged
    log-probs equal a re-evaluation.
    """
    from ppo_and_friends_amd.networks.distributions import MultiCategoricalDistribution, BernoulliDistribution
    from torch.distributions import Bernoulli, Categorical
    torch.manual_seed(3)
    n, nvec = 37, [3, 5, 2]
    logits = torch.randn(n, sum(nvec))
    acts = torch.stack([torch.randint(0, k, (n,)) for k in nvec], dim=1)
    d = MultiCategoricalDistribution(nvec, seed=5)
    lg = dev(logits.numpy()).requires_grad_(True)
    lp, ent = d.get_log_probs_and_entropy(lg, dev(acts.numpy()))
    (lp.sum() + 0.3 * ent.sum()).backward()
    ref_l = logits.clone().requires_grad_(True)
    start, lps, ents = 0, [], []
    for i, k in enumerate(nvec):
        c = Categorical(torch.softmax(ref_l[:, start:start + k], dim=-1))
        lps.append(c.log_prob(acts[:, i])); ents.append(c.entropy()); start += k
    rlp, rent = torch.stack(lps, -1).sum(-1), torch.stack(ents, -1).sum(-1)
    (rlp.sum() + 0.3 * rent.sum()).backward()
    np.testing.assert_allclose(lp.detach().cpu().numpy()[:, 0], rlp.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ent.detach().cpu().numpy(), rent.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(lg.grad.cpu().numpy(), ref_l.grad.numpy(), rtol=1e-4, atol=1e-6)
    a, raw, slp = d.sample_distribution(lg.detach())
    assert a.shape == (n, 3) and torch.equal(a, raw) and all(int(a[:, i].max()) < k and int(a[:, i].min()) >= 0 for i, k in enumerate(nvec))
    lp2, _ = d.get_log_probs_and_entropy(lg.detach(), a)
    torch.testing.assert_close(slp, lp2, rtol=1e-5, atol=1e-6)
    want = torch.stack([torch.argmax(logits[:, s:s + k], -1) for s, k in zip((0, 3, 8), nvec)], -1)
    assert torch.equal(d.refine_prediction(lg.detach()).cpu(), want)

    bits = 6
    blog = torch.randn(n, bits)
    bact = (torch.rand(n, bits) < 0.5).float()
    b = BernoulliDistribution(seed=9)
    bl = dev(blog.numpy()).requires_grad_(True)
    lp, ent = b.get_log_probs_and_entropy(bl, dev(bact.numpy()))
    (lp.sum() + 0.3 * ent.sum()).backward()
    rb = blog.clone().requires_grad_(True)
    rd = Bernoulli(probs=torch.sigmoid(rb))
    (rd.log_prob(bact).sum() + 0.3 * rd.entropy().sum()).backward()
    np.testing.assert_allclose(lp.detach().cpu().numpy()[:, 0], rd.log_prob(bact).sum(-1).detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(bl.grad.cpu().numpy(), rb.grad.numpy(), rtol=1e-4, atol=1e-6)
    a, raw, slp = b.sample_distribution(bl.detach())
    assert set(a.unique().tolist()) <= {0.0, 1.0} and a.shape == (n, bits)
    torch.testing.assert_close(slp, b.get_log_probs_and_entropy(bl.detach(), a)[0], rtol=1e-5, atol=1e-6)
    assert torch.equal(b.refine_prediction(bl.detach()).cpu(), (torch.sigmoid(blog) >= 0.5).float())


# ---------------------------------------------------------------- K6 against the reference's own distribution classes
@pytest.mark.parametrize("tag", ["c2", "c5"])
def test_categorical_kernels_match_reference_golden_g8(K, golden, tag):
    """CategoricalDistribution of the unmodified reference (fixture g8): log-probs, entropy and both gradients."""
    g = golden("g8_distributions")
    logits, actions = dev(torch.tensor(g[f"{tag}_logits"])), dev(torch.tensor(g[f"{tag}_actions"]).reshape(-1))
    lp, ent, probs = K.categorical_eval_fwd(logits, actions)
    np.testing.assert_allclose(probs.cpu().numpy(), g[f"{tag}_probs"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(lp.cpu().numpy(), g[f"{tag}_log_probs"].reshape(-1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), g[f"{tag}_entropy"], rtol=1e-5, atol=1e-5)
    n = logits.shape[0]
    ones, zeros = torch.ones(n, device="cuda"), torch.zeros(n, device="cuda")
    np.testing.assert_allclose(K.categorical_eval_bwd(probs, actions, ones, zeros).cpu().numpy(), g[f"{tag}_dlogp_dlogits"],
                               rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(K.categorical_eval_bwd(probs, actions, zeros, ones).cpu().numpy(), g[f"{tag}_dent_dlogits"],
                               rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("tag", ["unit", "bounds"])
def test_gaussian_kernels_match_reference_golden_g8(K, golden, tag):
    """GaussianDistribution of the unmodified reference (fixture g8): min_std floor, +-100 and 1e-6 clamps, entropy at
    the mean, gradients w.r.t. mean and log_std, refine with per-dimension bounds."""
    g = golden("g8_distributions")
    mean, log_std, raw = (dev(torch.tensor(g[f"g_{tag}_{k}"])) for k in ("mean", "log_std", "raw"))
    lp, ent = K.gaussian_tanh_eval_fwd(mean, log_std, raw)
    np.testing.assert_allclose(lp.cpu().numpy(), g[f"g_{tag}_log_probs"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), g[f"g_{tag}_entropy"], rtol=1e-5, atol=1e-5)
    n = mean.shape[0]
    ones, zeros = torch.ones(n, device="cuda"), torch.zeros(n, device="cuda")
    dm, dls = K.gaussian_tanh_eval_bwd(mean, log_std, raw, ones, zeros)
    np.testing.assert_allclose(dm.cpu().numpy(), g[f"g_{tag}_dlogp_dmean"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(dls.cpu().numpy(), g[f"g_{tag}_dlogp_dlogstd"], rtol=1e-4, atol=1e-3)
    dm, dls = K.gaussian_tanh_eval_bwd(mean, log_std, raw, zeros, ones)
    # the kernel uses the closed form -2 tanh(m); torch's float32 autograd chain (1 / (1 - t^2)) * 2 t * (1 - t^2) loses
    # up to ~1e-4 relative where tanh saturates (1 - t^2 ~ 1e-4 carries 8e-5 of rounding)
    np.testing.assert_allclose(dm.cpu().numpy(), g[f"g_{tag}_dent_dmean"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dls.cpu().numpy(), g[f"g_{tag}_dent_dlogstd"], rtol=1e-4, atol=1e-4)
    from ppo_and_friends_amd.networks.distributions import GaussianDistribution
    dist = GaussianDistribution(mean.shape[1], distribution_min=g[f"g_{tag}_low"], distribution_max=g[f"g_{tag}_high"]).cuda()
    np.testing.assert_allclose(dist.refine_prediction(mean).cpu().numpy(), g[f"g_{tag}_refined_prediction"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dist.refine_prediction(raw).cpu().numpy(), g[f"g_{tag}_refined_sample"], rtol=1e-5, atol=1e-6)
