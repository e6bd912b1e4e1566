"""
-m gpu parity tests of K13 (environment filters) through the C ABI, against the numpy restatement
of the reference's wrapper stack (oracle/filter_oracle.py) on the same raw env stream.

Tolerance: float32 statistics follow numpy's float32 row-sequential sums in the oracle and
float64 tree sums on the device -- held to north_star's 1e-5 (relative + absolute); the reward
stream is float64 on both sides and is held tighter.
"""
import copy

import numpy as np
import pytest
import torch

from oracle import filter_oracle as fo

pytestmark = pytest.mark.gpu


def _envs(A, E, O, T, critic_view="local", term_prob=0.15, seed=5):
    from ppo_and_friends_amd.environments.synthetic import SyntheticFixedLengthEnv
    from ppo_and_friends_amd.spaces import Discrete
    mk = lambda: SyntheticFixedLengthEnv(E, O, Discrete(3), T, "cuda", reward="uniform", seed=seed,
                                         term_prob=term_prob, num_agents=A, critic_view=critic_view)
    return mk(), mk()


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("A,E,O,critic_view", [(1, 64, 5, "local"), (3, 48, 7, "policy"), (1, 1000, 17, "local"),
                                               (2, 1, 4, "local")])
@pytest.mark.parametrize("obs_clip,reward_clip", [(None, None), ((-1.5, 1.5), (-0.8, 0.8))])
def test_filter_stack_matches_oracle(A, E, O, critic_view, obs_clip, reward_clip):
    from ppo_and_friends_amd.environments import filter_wrappers as fw
    T = 12
    raw, twin = _envs(A, E, O, T, critic_view)
    env = fw.wrap_environment(lambda: raw, normalize_obs=True, normalize_rewards=True, obs_clip=obs_clip,
                              reward_clip=reward_clip, gamma=0.97)
    orc = fo.FilteredEnvOracle(A, E, O, raw.critic_obs_dim, True, True, obs_clip, reward_clip, gamma=0.97)
    obs, cobs = env.reset()
    r_obs, r_cobs = twin.reset()
    o_obs, o_cobs = orc.filter_obs(_np(r_obs), _np(r_cobs))
    np.testing.assert_allclose(_np(obs), o_obs, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(_np(cobs), o_cobs, rtol=1e-5, atol=1e-5)
    action = torch.zeros(A * E, dtype=torch.int64, device="cuda")
    for t in range(T):
        obs, cobs, rew, term, trunc, tobs = env.step(action)
        r_obs, r_cobs, r_rew, r_term, r_trunc, r_tobs = twin.step(action)
        o_obs, o_cobs, o_rew = orc.filter_step(_np(r_obs), _np(r_cobs), _np(r_rew), _np(r_term), _np(r_trunc))
        np.testing.assert_allclose(_np(obs), o_obs, rtol=1e-5, atol=1e-5, err_msg=f"obs t={t}")
        np.testing.assert_allclose(_np(cobs), o_cobs, rtol=1e-5, atol=1e-5, err_msg=f"critic obs t={t}")
        np.testing.assert_allclose(_np(rew), o_rew, rtol=1e-6, atol=1e-6, err_msg=f"reward t={t}")
        # logged next observation: filtered, except the raw terminal observation of terminated envs
        want_t = np.where(_np(r_term)[:, None], _np(r_tobs), o_obs)
        np.testing.assert_allclose(_np(tobs), want_t, rtol=1e-5, atol=1e-5)
        assert torch.equal(env.natural_reward, r_rew)
    # running state after T steps
    w = env
    while not isinstance(w, fw.RewardNormalizer):
        w = w.env
    for a, aid in enumerate(raw.agent_ids):
        st = w.running_stats[aid]
        np.testing.assert_allclose(st["mean"], orc.rew_norm.stats[a].mean, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(st["variance"], orc.rew_norm.stats[a].variance, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(st["count"], orc.rew_norm.stats[a].count, rtol=1e-12)
        np.testing.assert_allclose(_np(w.running_reward[a]), orc.rew_norm.running_reward[a], rtol=1e-12, atol=1e-12)
    while not isinstance(w, fw.ObservationNormalizer):
        w = w.env
    for a, aid in enumerate(raw.agent_ids):
        for got, want in ((w.actor_running_stats[aid], orc.obs_norm.stats[a]),
                          (w.critic_running_stats[aid], orc.cobs_norm.stats[a])):
            np.testing.assert_allclose(got["mean"], want.mean, rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(got["variance"], want.variance, rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(got["count"], want.count, rtol=1e-12)


def test_frozen_stats_and_clip_only():
    """update_stats=False (test mode) normalises with frozen stats; clip-only stacks skip the moments launch."""
    from ppo_and_friends_amd.environments import filter_wrappers as fw
    A, E, O, T = 2, 32, 6, 6
    raw, twin = _envs(A, E, O, T)
    env = fw.RewardClipper(fw.ObservationClipper(raw, clip_range=(-0.5, 0.7)), clip_range=(-0.2, 0.3))
    obs, cobs = env.reset()
    r_obs, _ = twin.reset()
    np.testing.assert_array_equal(_np(obs), np.clip(_np(r_obs), -0.5, 0.7))
    action = torch.zeros(A * E, dtype=torch.int64, device="cuda")
    obs, cobs, rew, *_ = env.step(action)
    r_obs, r_cobs, r_rew, *_ = twin.step(action)
    np.testing.assert_array_equal(_np(obs), np.clip(_np(r_obs), -0.5, 0.7))
    np.testing.assert_array_equal(_np(rew), np.clip(_np(r_rew), -0.2, 0.3))

    raw, twin = _envs(A, E, O, T)
    env = fw.wrap_environment(lambda: raw, test_mode=True, gamma=0.9)
    orc = fo.FilteredEnvOracle(A, E, O, O, gamma=0.9, update_stats=False)
    env.reset(); twin.reset()
    for t in range(3):
        obs, cobs, rew, *_ = env.step(action)
        r = twin.step(action)
        o_obs, o_cobs, o_rew = orc.filter_step(*[_np(x) for x in r[:5]])
        np.testing.assert_allclose(_np(obs), o_obs, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(_np(rew), o_rew, rtol=1e-6, atol=1e-6)


def test_wrapper_order_is_enforced():
    from ppo_and_friends_amd.environments import filter_wrappers as fw
    raw, _ = _envs(1, 8, 3, 2)
    with pytest.raises(NotImplementedError):
        fw.ObservationNormalizer(fw.ObservationClipper(raw))


def test_two_rank_records_merge():
    """
    The N > 1 arithmetic without processes: two rank states on one device, their records
    concatenated as the all-gather would, applied on both -- against the oracle's gathered updates.
    """
    from ppo_and_friends_amd import kernels as K
    rng = np.random.default_rng(11)
    G, n, W, R, gamma = 2, 40, 5, 2, 0.95
    dev = lambda a, dt: torch.as_tensor(a, dtype=dt).cuda().contiguous()
    st = []
    for r in range(R):
        st.append(dict(
            obs=(torch.zeros(G * W, device="cuda"), torch.ones(G * W, device="cuda"),
                 torch.full((G * W,), 1e-4, dtype=torch.float64, device="cuda")),
            rew=(torch.zeros(G * n, dtype=torch.float64, device="cuda"), torch.zeros(G, dtype=torch.float64, device="cuda"),
                 torch.ones(G, dtype=torch.float64, device="cuda"), torch.full((G,), 1e-4, dtype=torch.float64, device="cuda"))))
    o_obs = [fo.ObservationNormalizerOracle(G, W) for _ in range(R)]
    o_rew = [fo.RewardNormalizerOracle(G, n, gamma=gamma) for _ in range(R)]
    L = K.env_filter_record_len(G, W, 0, True)
    for t in range(4):
        xs = [rng.standard_normal((G * n, W)).astype(np.float32) * (1 + r) + r for r in range(R)]
        rs = [rng.uniform(-1, 1, G * n).astype(np.float32) for _ in range(R)]
        ds = [rng.uniform(0, 1, G * n) < 0.2 for _ in range(R)]
        fl, recs = [], torch.zeros(R, L, dtype=torch.float64, device="cuda")
        for r in range(R):
            x, out = dev(xs[r], torch.float32), torch.empty(G * n, W, device="cuda")
            rw, ro = dev(rs[r], torch.float32), torch.empty(G * n, device="cuda")
            dn = dev(ds[r], torch.bool)
            of = K.obs_filter(x, out, G, n, st[r]["obs"], True, None)
            rf = K.reward_filter(rw, dn, None, ro, G, n, st[r]["rew"], True, None, gamma)
            K.env_filter_moments(of, None, rf, G, n, recs[r])
            fl.append((of, rf, out, ro, (x, rw, dn)))
        want_r = fo.reward_filter_ranks(o_rew, rs, ds)
        for r in range(R):
            of, rf, out, ro, _keep = fl[r]
            K.env_filter_apply(of, None, rf, G, n, recs.reshape(-1))
            want_o = o_obs[r].filter(xs[r], gathered=xs)
            np.testing.assert_allclose(_np(out), want_o, rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(_np(ro), want_r[r], rtol=1e-6, atol=1e-6)
    for r in range(R):
        for g in range(G):
            np.testing.assert_allclose(_np(st[r]["rew"][1])[g], o_rew[r].stats[g].mean, rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(_np(st[r]["rew"][2])[g], o_rew[r].stats[g].variance, rtol=1e-9)
            np.testing.assert_allclose(_np(st[r]["rew"][3])[g], o_rew[r].stats[g].count, rtol=1e-12)
        np.testing.assert_allclose(_np(st[r]["obs"][0]).reshape(G, W), np.stack([s.mean for s in o_obs[r].stats]),
                                   rtol=1e-5, atol=1e-6)


def test_filter_stack_fuzz_against_the_oracle():
    """
    Randomised stacks and shapes (hypothesis, derandomised): any subset of {obs normaliser, obs clipper, reward
    normaliser, reward clipper}, 1-4 agents, 1-300 envs (E = 1: zero batch variance), critic views, termination
    densities up to "every step", against oracle/filter_oracle.py on the same raw stream.
    """
    from hypothesis import given, settings, strategies as st, HealthCheck
    from ppo_and_friends_amd.environments import filter_wrappers as fw

    @settings(max_examples=30, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(A=st.integers(1, 4), E=st.integers(1, 300), O=st.integers(1, 40), view=st.sampled_from(["local", "policy"]),
           norm_obs=st.booleans(), norm_rew=st.booleans(), obs_clip=st.sampled_from([None, (-1.0, 1.0), (-0.1, 3.0)]),
           rew_clip=st.sampled_from([None, (-0.5, 0.5)]), term=st.sampled_from([0.0, 0.2, 1.0]), T=st.integers(1, 10),
           gamma=st.sampled_from([0.99, 0.9, 1.0]))
    def run(A, E, O, view, norm_obs, norm_rew, obs_clip, rew_clip, term, T, gamma):
        if not (norm_obs or norm_rew or obs_clip or rew_clip):
            return
        raw, twin = _envs(A, E, O, T, view, term_prob=term, seed=11)
        env = fw.wrap_environment(lambda: raw, normalize_obs=norm_obs, normalize_rewards=norm_rew, obs_clip=obs_clip,
                                  reward_clip=rew_clip, gamma=gamma)
        orc = fo.FilteredEnvOracle(A, E, O, raw.critic_obs_dim, norm_obs, norm_rew, obs_clip, rew_clip, gamma=gamma)
        obs, cobs = env.reset()
        r_obs, r_cobs = twin.reset()
        o_obs, o_cobs = orc.filter_obs(_np(r_obs), _np(r_cobs))
        tol = dict(rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(_np(obs), o_obs, **tol)
        np.testing.assert_allclose(_np(cobs), o_cobs, **tol)
        action = torch.zeros(A * E, dtype=torch.int64, device="cuda")
        for t in range(T):
            obs, cobs, rew, term_t, trunc, tobs = env.step(action)
            r_obs, r_cobs, r_rew, r_term, r_trunc, r_tobs = twin.step(action)
            o_obs, o_cobs, o_rew = orc.filter_step(_np(r_obs), _np(r_cobs), _np(r_rew), _np(r_term), _np(r_trunc))
            np.testing.assert_allclose(_np(obs), o_obs, err_msg=f"obs t={t}", **tol)
            np.testing.assert_allclose(_np(cobs), o_cobs, err_msg=f"critic obs t={t}", **tol)
            np.testing.assert_allclose(_np(rew), o_rew, rtol=1e-5, atol=1e-6, err_msg=f"reward t={t}")

    run()
