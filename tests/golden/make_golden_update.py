#!/usr/bin/env python3
"""
Golden vectors for the UPDATE half of the hot path (fixtures g8 ...), recorded from the *unmodified*
reference imported from /root/reference (build container only; see tests/golden/ref_import.py for the
recipe: mpi4py single-rank stand-in + metadata-only gymnasium stand-in).  Fixtures are data: inputs
(tables, initial weights, recorded actions, shuffles) and the reference's outputs.

  g8_distributions      CategoricalDistribution / GaussianDistribution       networks/distributions.py:199-269,441-694
  g9_value_normalizer   RunningStatNormalizer normalize / denormalize        utils/misc.py:61-128
  g10_icm               ICM.forward + backward (discrete, continuous)        networks/ppo_networks/icm.py:22-430
  g11_networks          FeedForwardNetwork / LSTMNetwork / MATActorCritic    networks/ppo_networks/*.py, actor_critic/*
  g12_<scenario>        the reference's own PPO object (ppo.py:126-2567) driven for whole iterations over a
                        table-driven environment written against the reference's public env-wrapper API
                        (PPOEnvironmentWrapper): rollout -> EpisodeInfo / PPODataset -> _ppo_batch_train /
                        _icm_batch_train epochs -> status_dict, datasets, shuffles, first-mini-batch losses and
                        raw gradients, final weights, normaliser states.
  g13_filters           ObservationNormalizer / RewardNormalizer / clippers  environments/filter_wrappers.py:113-719

Usage:  python tests/golden/make_golden_update.py [fixture names ...]
"""
import copy
import os
import sys
import tempfile

sys.dont_write_bytecode = True
if os.environ.get("PYTHONHASHSEED") != "0":
    # PPOPolicy.register_agent orders a policy's agents through a set of strings (policies/ppo_policy.py:375-376):
    # pin the hash seed so that regenerating a multi-agent fixture reproduces it bit for bit
    os.environ["PYTHONHASHSEED"] = "0"
    os.execv(sys.executable, [sys.executable] + sys.argv)

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402


# ------------------------------------------------------------------------------------------------
# the table-driven environment (test-side code written against the reference's env-wrapper API)
# ------------------------------------------------------------------------------------------------
def make_tables(seed, T, E, A, O, reward="uniform", term_prob=0.0, obs_scale=1.0, obs_shift=0.0):
    """obs [T+1,E,A,O] f32, rewards [T,E,A] f32, terminations [T,E] bool (never on the last row)."""
    rng = np.random.default_rng(seed)
    obs = (rng.standard_normal((T + 1, E, A, O)) * obs_scale + obs_shift).astype(np.float32)
    rew = np.ones((T, E, A), np.float32) if reward == "ones" else rng.uniform(-1, 1, (T, E, A)).astype(np.float32)
    term = rng.uniform(0, 1, (T, E)) < term_prob
    term[-1] = False
    return obs, rew, term


def table_env_class():
    from ppo_and_friends.environments.ppo_env_wrappers import PPOEnvironmentWrapper
    from gymnasium.spaces import Box

    class TableEnv(PPOEnvironmentWrapper):
        """
        One environment instance = column `e` of the tables.  step k returns obs[k+1]; a finished env
        "resets" onto the observation it stopped on (the contract of the product's synthetic env);
        a reset after a full pass (T steps) rewinds to row 0.  All agents finish together.
        """

        def __init__(self, tables, e, action_space, n_agents, **kw):
            self._tables, self._e, self._aspace, self._A = tables, e, action_space, n_agents
            self.t = 0
            self.actions_seen = []
            super().__init__(env=None, **kw)

        def _define_agent_ids(self):
            self.agent_ids = tuple(f"agent{a}" for a in range(self._A))

        def _define_multi_agent_spaces(self):
            O = self._tables[0].shape[-1]
            for a in self.agent_ids:
                self.observation_space[a] = Box(-np.inf, np.inf, (O,), np.float32)
                self.action_space[a] = self._aspace

        def seed(self, s):
            pass

        def _obs(self, t):
            row = self._tables[0][t, self._e]
            return {a: row[i].copy() for i, a in enumerate(self.agent_ids)}

        def reset(self):
            T = self._tables[1].shape[0]
            if self.t >= T:
                self.t = 0
            self.all_done = False
            self._reset_done_agents()
            obs = self._obs(self.t)
            return obs, self._construct_critic_observation(obs, self.agents_done)

        def step(self, action):
            obs_t, rew_t, term_t = self._tables
            t = self.t
            self.actions_seen.append({a: np.array(action[a]).copy() for a in action})
            obs = self._obs(t + 1)
            term = bool(term_t[t, self._e])
            rew = {a: float(rew_t[t, self._e, i]) for i, a in enumerate(self.agent_ids)}
            terminated = {a: term for a in self.agent_ids}
            truncated = {a: False for a in self.agent_ids}
            info = {a: {} for a in self.agent_ids}
            self.all_done = term
            self._update_done_agents(terminated)
            self.t += 1
            return obs, self._construct_critic_observation(obs, self.agents_done), rew, terminated, truncated, info

    return TableEnv


# ------------------------------------------------------------------------------------------------
# g12: whole iterations of the reference's PPO
# ------------------------------------------------------------------------------------------------
def _flat_state(module):
    return {k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def run_scenario(name, *, seed, E, T, A, O, action_space, reward="uniform", term_prob=0.0, max_ts_per_ep=200,
                 batch_size=64, epochs=2, iterations=1, critic_view="local", policy_class=None, policy_args=None,
                 ppo_args=None, obs_scale=1.0, obs_shift=0.0, ac_network=None, rank=0):
    """`rank`: this process's rank under the R-process mpi4py stand-in (ref_import.py): its own environment tables and
    seed + rank for everything random, as `ppoaf train` seeds its ranks (ppoaf_cli.py:419)."""
    from ppo_and_friends.ppo import PPO
    from ppo_and_friends.networks.ppo_networks.feed_forward import FeedForwardNetwork
    from torch.utils.data import DataLoader

    TableEnv = table_env_class()
    tables = make_tables(seed + 7919 * rank, T, E, A, O, reward, term_prob, obs_scale, obs_shift)
    counter = {"n": 0}
    instances = []
    pmap = lambda agent_id: "agent"

    def env_generator():
        e = max(counter["n"] - 1, 0) if E > 1 else 0      # VectorizedEnv: call 0 is the template when E > 1
        counter["n"] += 1
        env = TableEnv(tables, e, action_space, A, critic_view=critic_view, policy_mapping_fn=pmap)
        instances.append(env)
        return env

    template = TableEnv(tables, 0, action_space, A, critic_view=critic_view, policy_mapping_fn=pmap)
    obs_space = template.observation_space["agent0"]
    cobs_space = template.critic_observation_space["agent0"]
    pargs = dict(policy_args or {})
    pargs.setdefault("ac_network", ac_network or FeedForwardNetwork)
    settings = {"agent": (policy_class, obs_space, cobs_space, action_space, pargs)}

    state_dir = tempfile.mkdtemp(prefix="ppoaf_golden_state_")
    torch.manual_seed(seed + rank)
    np.random.seed(seed + rank)
    kw = dict(device="cpu", random_seed=seed + rank, envs_per_proc=E, max_ts_per_ep=max_ts_per_ep, batch_size=batch_size,
              ts_per_rollout=T, epochs_per_iter=epochs, normalize_obs=False, normalize_rewards=False,
              state_path=state_dir, save_train_scores=False, save_ep_scores=False, save_avg_ep_len=False,
              save_running_time=False, save_bs_info=False, checkpoint_every=10 ** 9)
    kw.update(ppo_args or {})
    ppo = PPO(env_generator=env_generator, policy_settings=settings, policy_mapping_fn=pmap, **kw)
    pol = ppo.policies["agent"]

    out = {}
    out["cfg_names"] = np.array(["seed", "E", "T", "A", "O", "max_ts_per_ep", "batch_size", "epochs", "iterations"])
    out["cfg"] = np.array([seed, E, T, A, O, max_ts_per_ep, batch_size, epochs, iterations], dtype=np.int64)
    out["obs_table"], out["reward_table"], out["term_table"] = tables
    out["agent_ids"] = np.array(list(pol.agent_ids))
    for tag, net in (("actor", pol.actor), ("critic", pol.critic)) + ((("icm", pol.icm_model),) if pol.enable_icm else ()):
        for k, v in _flat_state(net).items():
            out[f"init_{tag}.{k}"] = v

    # ---- observation hooks (record, then call the reference's own method unchanged)
    rec = {"steps": [], "epochs": [], "mb": [], "datasets": []}
    orig_actions = ppo.get_rollout_actions
    orig_values = ppo.get_policy_values
    orig_denorm = ppo.get_denormalized_values if ppo.normalize_values else None

    def rec_actions(obs):
        raw, act, lp = orig_actions(obs)
        rec["steps"].append(dict(
            obs={a: np.array(obs[a]).copy() for a in obs},
            raw={a: np.array(raw[a]).copy() for a in raw}, act={a: np.array(act[a]).copy() for a in act},
            logp={a: lp[a].detach().numpy().copy() for a in lp}))
        return raw, act, lp

    ppo.get_rollout_actions = rec_actions
    if orig_denorm is not None:
        def rec_denorm(values):
            v = orig_denorm(values)
            rec.setdefault("values", []).append({a: v[a].detach().numpy().copy() for a in v})
            rec.setdefault("values_step", []).append(len(rec["steps"]))
            return v
        ppo.get_denormalized_values = rec_denorm
    else:
        def rec_values(cobs):
            v = orig_values(cobs)
            rec.setdefault("values", []).append({a: v[a].detach().numpy().copy() for a in v})
            rec.setdefault("values_step", []).append(len(rec["steps"]))
            return v
        ppo.get_policy_values = rec_values

    orig_intr = ppo.apply_intrinsic_rewards
    orig_nat = ppo.get_natural_reward

    def rec_intr(ext_rewards, prev_obs, obs, actions):
        ext = {a: np.array(ext_rewards[a], dtype=np.float64).copy() for a in ext_rewards}
        rewards, intr = orig_intr(ext_rewards, prev_obs, obs, actions)
        rec.setdefault("rewards", []).append(dict(
            ext=ext, reward={a: np.array(rewards[a], dtype=np.float64).copy() for a in rewards},
            intr={a: np.broadcast_to(np.array(intr[a], dtype=np.float64), ext[a].shape).copy() for a in intr},
            obs_minmax=np.array([min(float(np.min(obs[a])) for a in obs), max(float(np.max(obs[a])) for a in obs)])))
        return rewards, intr

    def rec_nat(info):
        have, nat = orig_nat(info)
        rec.setdefault("nat", []).append(None if not have else {a: np.array(nat[a], dtype=np.float64).copy() for a in nat})
        return have, nat

    ppo.apply_intrinsic_rewards = rec_intr
    ppo.get_natural_reward = rec_nat

    if getattr(pol, "agent_grouping", False):              # MAT: the agents' slot order is reshuffled every rollout (ppo.py:1643-1644)
        orig_shuffle = pol.shuffle_agent_ids

        def rec_shuffle():
            orig_shuffle()
            rec.setdefault("slot_orders", []).append(np.array([int(str(a)[len("agent"):]) for a in pol.agent_ids]))

        pol.shuffle_agent_ids = rec_shuffle

    orig_finalize = pol.finalize_dataset

    def rec_finalize():
        orig_finalize()
        ds = pol.dataset
        d = dict(observations=ds.observations, next_observations=ds.next_observations,
                 critic_observations=ds.critic_observations, actions=ds.actions, raw_actions=ds.raw_actions,
                 advantages=ds.advantages, log_probs=ds.log_probs, rewards_to_go=ds.rewards_to_go, values=ds.values)
        if getattr(pol, "using_lstm", False):
            d.update(actor_hidden=ds.actor_hidden, actor_cell=ds.actor_cell, critic_hidden=ds.critic_hidden,
                     critic_cell=ds.critic_cell)
        rec["datasets"].append({k: v.detach().cpu().numpy().copy() for k, v in d.items()} | {"ep_lens": np.array(ds.ep_lens)})

    pol.finalize_dataset = rec_finalize

    orig_update = pol.update_weights

    def rec_update(actor_loss, critic_loss):
        if len(rec["mb"]) < 1:                                  # first mini-batch: losses + raw (unclipped) gradients
            ga = torch.autograd.grad(actor_loss, [p for p in pol.actor.parameters() if p.requires_grad],
                                     retain_graph=True, allow_unused=True)
            gc = torch.autograd.grad(critic_loss, [p for p in pol.critic.parameters() if p.requires_grad],
                                     retain_graph=True, allow_unused=True)
            total = None
            if getattr(pol, "agent_grouping", False):           # MAT: one optimiser over actor + critic, summed loss (mat_policy.py:677-699)
                both = [p for p in pol.actor.parameters()] + [p for p in pol.critic.parameters()]
                gt = torch.autograd.grad(actor_loss + critic_loss, both, retain_graph=True, allow_unused=True)
                total = np.concatenate([np.zeros(p.numel(), np.float32) if g is None else g.numpy().reshape(-1)
                                        for p, g in zip(both, gt)])
            rec["mb"].append(dict(
                total_grad=total,
                actor_loss=float(actor_loss.item()), critic_loss=float(critic_loss.item()),
                actor_grads=[np.zeros(0, np.float32) if g is None else g.numpy().copy() for g in ga],
                critic_grads=[np.zeros(0, np.float32) if g is None else g.numpy().copy() for g in gc]))
        return orig_update(actor_loss, critic_loss)

    pol.update_weights = rec_update

    # what mpi_avg_gradients (utils/mpi_utils.py:89-111) left in .grad on the first mini-batch: the rank-averaged gradient
    import ppo_and_friends.policies.ppo_policy as ref_policy_module
    import ppo_and_friends.ppo as ref_ppo_module
    orig_avg = ref_policy_module.mpi_avg_gradients

    def rec_avg(model):
        orig_avg(model)
        tag = "actor" if model is pol.actor else "critic" if model is pol.critic else "icm"
        if tag not in rec.setdefault("avg_grads", {}):
            rec["avg_grads"][tag] = np.concatenate([np.zeros(p.numel(), np.float32) if p.grad is None else
                                                    p.grad.detach().numpy().reshape(-1).copy() for p in model.parameters()])

    ref_policy_module.mpi_avg_gradients = rec_avg
    ref_ppo_module.mpi_avg_gradients = rec_avg

    orig_train = ppo._ppo_batch_train

    def harvest_perm(data_loader):
        state = torch.get_rng_state()
        probe = DataLoader(data_loader.dataset, batch_size=data_loader.batch_size, shuffle=True)
        perm = torch.cat([b[12] for b in probe]).numpy().copy()
        torch.set_rng_state(state)
        return perm

    def rec_train(data_loader, policy_id):
        perm = harvest_perm(data_loader)
        orig_train(data_loader, policy_id)
        sd = ppo.status_dict[policy_id]
        vs = None
        if ppo.normalize_values:
            rs_ = ppo.value_normalizers[policy_id].running_stats
            vs = np.array([rs_.mean, rs_.variance, rs_.count], dtype=np.float64)
        rec["epochs"].append(dict(kind="ppo", perm=perm, value_stats=vs, stats=np.array(
            [sd["actor loss"], sd["critic loss"], sd["kl avg"], sd["weighted entropy"]], dtype=np.float64)))

    ppo._ppo_batch_train = rec_train
    if pol.enable_icm:
        orig_icm = ppo._icm_batch_train

        def rec_icm(data_loader, policy_id):
            perm = harvest_perm(data_loader)
            orig_icm(data_loader, policy_id)
            rec["epochs"].append(dict(kind="icm", perm=perm,
                                      stats=np.array([ppo.status_dict[policy_id]["icm loss"]], dtype=np.float64)))

        ppo._icm_batch_train = rec_icm

    status_after_rollout = []
    orig_rollout = ppo.rollout

    def rec_rollout():
        orig_rollout()
        status_after_rollout.append(copy.deepcopy(ppo.status_dict))
        rec.setdefault("ppo_epochs_before_rollout", []).append(sum(e["kind"] == "ppo" for e in rec["epochs"]))

    ppo.rollout = rec_rollout

    try:
        ppo.learn(E * T * iterations * max(int(os.environ.get("PPOAF_FAKE_MPI_SIZE", "1")), 1))   # (timesteps count every rank's)
    finally:
        ref_policy_module.mpi_avg_gradients = orig_avg
        ref_ppo_module.mpi_avg_gradients = orig_avg

    # ---- pack
    agents = [f"agent{a}" for a in range(A)]
    n_steps = len(rec["steps"])
    stack = lambda key: np.stack([np.stack([s[key][a] for a in agents], 1) for s in rec["steps"]])     # [steps, E, A, ...]
    out["step_obs"], out["step_raw_actions"], out["step_actions"], out["step_log_probs"] = \
        stack("obs"), stack("raw"), stack("act"), stack("logp")
    # values: the first n_steps-per-rollout calls of each rollout are the per-step values; extra calls are bootstraps
    vals = rec["values"]
    out["values_calls"] = np.stack([np.stack([v[a] for a in agents], 1) for v in vals])
    out["values_calls_step"] = np.array(rec["values_step"], dtype=np.int64)   # 1-based step the call belongs to
    sr = lambda key: np.stack([np.stack([r[key][a] for a in agents], 1) for r in rec["rewards"]])[..., 0]   # [steps, E, A]
    out["step_ext_rewards"], out["step_rewards"], out["step_intr_rewards"] = sr("ext"), sr("reward"), sr("intr")
    out["step_natural_rewards"] = np.stack([
        np.stack([(rec["rewards"][i]["ext"] if n is None else n)[a] for a in agents], 1) for i, n in enumerate(rec["nat"])])[..., 0]
    out["step_next_obs_minmax"] = np.stack([r["obs_minmax"] for r in rec["rewards"]])
    for i, d in enumerate(rec["datasets"]):
        for k, v in d.items():
            out[f"it{i}_ds_{k}"] = v
    ppo_ep = [e for e in rec["epochs"] if e["kind"] == "ppo"]
    icm_ep = [e for e in rec["epochs"] if e["kind"] == "icm"]
    # PPO epochs the reference actually ran in each iteration: fewer than `epochs` when the KL early stop of
    # ppo.py:2221-2232 broke out of the epoch loop
    if int(os.environ.get("PPOAF_FAKE_MPI_SIZE", "1")) > 1:   # (R-rank fixtures only: older fixtures regenerate unchanged)
        for tag, v in rec.get("avg_grads", {}).items():
            out[f"mb0_{tag}_avg_grad"] = v          # identical on every rank: the all-reduced mean of the ranks' gradients
        out["epoch_value_stats"] = np.stack([e["value_stats"] for e in ppo_ep])       # after every PPO epoch
    if "target_kl" in pargs:                  # (only the KL-stop scenarios carry these: older fixtures regenerate unchanged)
        marks = rec["ppo_epochs_before_rollout"] + [len(ppo_ep)]
        out["epochs_run"] = np.diff(np.array(marks, dtype=np.int64))
        out["target_kl"] = np.array([float(pol.target_kl)], dtype=np.float64)
    out["epoch_perms"] = np.stack([e["perm"] for e in ppo_ep])
    out["epoch_stats"] = np.stack([e["stats"] for e in ppo_ep])         # actor loss, critic loss, kl avg, weighted entropy
    if icm_ep:
        out["icm_epoch_perms"] = np.stack([e["perm"] for e in icm_ep])
        out["icm_epoch_stats"] = np.stack([e["stats"] for e in icm_ep])
    if "slot_orders" in rec:
        out["slot_orders"] = np.stack(rec["slot_orders"])               # [rollouts, A]: original agent index in slot j
    mb = rec["mb"][0]
    out["mb0_losses"] = np.array([mb["actor_loss"], mb["critic_loss"]], dtype=np.float64)
    out["mb0_actor_grad"] = np.concatenate([g.reshape(-1) for g in mb["actor_grads"]])
    out["mb0_critic_grad"] = np.concatenate([g.reshape(-1) for g in mb["critic_grads"]])
    if mb["total_grad"] is not None:
        out["mb0_total_grad"] = mb["total_grad"]                        # d(actor_loss + critic_loss) / d(actor params, critic params)
    for tag, net in (("actor", pol.actor), ("critic", pol.critic)) + ((("icm", pol.icm_model),) if pol.enable_icm else ()):
        for k, v in _flat_state(net).items():
            out[f"final_{tag}.{k}"] = v
    if ppo.normalize_values:
        rs = ppo.value_normalizers["agent"].running_stats
        out["value_stats"] = np.array([rs.mean, rs.variance, rs.count], dtype=np.float64)
    keys = ["score avg", "natural score avg", "top score", "top natural reward", "bootstrap avg"]
    if pol.enable_icm:
        keys.append("intrinsic score avg")
    rng_keys = ["natural reward range", "reward range", "bootstrap range", "obs range"] + \
        (["intr reward range"] if pol.enable_icm else [])
    gkeys = ["total episodes", "longest episode", "shortest episode", "average episode", "timesteps"]
    out["rollout_status_keys"] = np.array(keys)
    out["rollout_status"] = np.array([[float(s["agent"][k]) for k in keys] for s in status_after_rollout])
    out["rollout_range_keys"] = np.array(rng_keys)
    out["rollout_ranges"] = np.array([[[float(x) for x in s["agent"][k]] for k in rng_keys] for s in status_after_rollout])
    out["global_status_keys"] = np.array(gkeys)
    out["global_status"] = np.array([[float(s["global status"][k]) for k in gkeys] for s in status_after_rollout])
    vec = ppo.env
    while not hasattr(vec, "envs"):
        vec = vec.env
    out["env_actions_env0"] = np.stack([np.stack([np.asarray(s[a]) for a in agents]) for s in vec.envs[0].actions_seen])
    # filter stacks (obs / reward normalisers) keep running statistics worth pinning
    env = ppo.env
    while env is not None and hasattr(env, "env"):
        cls = type(env).__name__
        for attr in ("actor_running_stats", "critic_running_stats", "running_stats"):
            val = getattr(env, attr, None)
            if isinstance(val, dict):
                for a, rs in val.items():
                    out[f"filter_{cls}_{attr}_{a}_mean"] = np.asarray(rs.mean, dtype=np.float64)
                    out[f"filter_{cls}_{attr}_{a}_var"] = np.asarray(rs.variance, dtype=np.float64)
                    out[f"filter_{cls}_{attr}_{a}_count"] = np.array([rs.count], dtype=np.float64)
        rr = getattr(env, "running_reward", None)
        if isinstance(rr, dict):
            for a, v in rr.items():
                out[f"filter_{cls}_running_reward_{a}"] = np.asarray(v, dtype=np.float64)
        env = env.env
    import shutil
    shutil.rmtree(state_dir, ignore_errors=True)
    return out



# ------------------------------------------------------------------------------------------------
# small unit fixtures: g8 distributions, g9 value normaliser, g10 ICM forward / backward, g13 filter wrappers
# ------------------------------------------------------------------------------------------------
def gen_g8_distributions():
    """networks/distributions.py:199-269 (Categorical), :441-694 (tanh-Gaussian) on fixed inputs."""
    import ppo_and_friends.networks.distributions as D
    from gymnasium.spaces import Box, Discrete
    out = {}
    g = torch.Generator().manual_seed(8)
    # Categorical: softmax output_func (:1043-1045) -> Categorical(probs): log_prob, entropy, refine_prediction
    for tag, (n, k) in {"c2": (64, 2), "c5": (96, 5), "c17": (10, 17)}.items():
        dist, output_func = D.get_actor_distribution(Discrete(k))
        logits = torch.randn(n, k, generator=g) * 3.0
        logits[0] = 40.0 * torch.nn.functional.one_hot(torch.tensor(0), k) - 20.0          # a saturated row
        actions = torch.randint(0, k, (n, 1), generator=g)
        probs = output_func(logits.clone()).requires_grad_(False)
        lg = logits.clone().requires_grad_(True)
        td = dist.get_distribution(output_func(lg))
        lp = dist.get_log_probs(td, actions)
        ent = dist.get_entropy(td, output_func(lg))
        glp, = torch.autograd.grad(lp.sum(), lg, retain_graph=True)
        gent, = torch.autograd.grad(ent.sum(), lg)
        out.update({f"{tag}_logits": logits.numpy(), f"{tag}_actions": actions.numpy(), f"{tag}_probs": probs.numpy(),
                    f"{tag}_log_probs": lp.detach().numpy(), f"{tag}_entropy": ent.detach().numpy(),
                    f"{tag}_dlogp_dlogits": glp.numpy(), f"{tag}_dent_dlogits": gent.numpy(),
                    f"{tag}_refined": dist.refine_prediction(probs).numpy()})
    # Gaussian: per-dimension bounds, min_std clamp, +-100 clamp, tanh correction, entropy := -log_prob(mean)
    for tag, (lo, hi) in {"unit": ([-1.0] * 6, [1.0] * 6), "bounds": ([-1.0, -2.0, 0.0], [1.0, 2.0, 5.0])}.items():
        space = Box(np.array(lo, np.float32), np.array(hi, np.float32))
        dist, _ = D.get_actor_distribution(space)
        n, d = 48, len(lo)
        mean = (torch.randn(n, d, generator=g) * 1.5).requires_grad_(True)
        raw = torch.randn(n, d, generator=g) * 2.0
        raw[0] = 9.0                                                                      # tanh' below the 1e-6 clamp
        raw[1, 0] = 60.0                                                                  # normal log-prob below the -100 clamp
        with torch.no_grad():
            dist.log_std.copy_(torch.linspace(-6.0, 0.5, d))                              # first dims hit min_std = 0.01
        td = dist.get_distribution(mean)
        lp = dist.get_log_probs(td, raw)
        ent = dist.get_entropy(td, mean)
        gm, gs = torch.autograd.grad(lp.sum(), [mean, dist.log_std], retain_graph=True)
        em, es = torch.autograd.grad(ent.sum(), [mean, dist.log_std])
        out.update({f"g_{tag}_low": np.array(lo, np.float32), f"g_{tag}_high": np.array(hi, np.float32),
                    f"g_{tag}_mean": mean.detach().numpy(), f"g_{tag}_raw": raw.numpy(),
                    f"g_{tag}_log_std": dist.log_std.detach().numpy().copy(), f"g_{tag}_std": td.stddev[0].detach().numpy(),
                    f"g_{tag}_log_probs": lp.detach().numpy(), f"g_{tag}_entropy": ent.detach().numpy(),
                    f"g_{tag}_dlogp_dmean": gm.numpy(), f"g_{tag}_dlogp_dlogstd": gs.numpy(),
                    f"g_{tag}_dent_dmean": em.numpy(), f"g_{tag}_dent_dlogstd": es.numpy(),
                    f"g_{tag}_refined_sample": dist.refine_sample(raw).numpy(),
                    f"g_{tag}_refined_prediction": dist.refine_prediction(mean.detach()).numpy()})
    return out


def gen_g9_value_normalizer():
    """utils/misc.py:61-128: RunningStatNormalizer.normalize (update + normalise) / denormalize sequences."""
    from ppo_and_friends.utils.misc import RunningStatNormalizer
    out = {}
    rng = np.random.default_rng(9)
    vn = RunningStatNormalizer("value_normalizer", torch.device("cpu"))
    out["denorm_fresh"] = vn.denormalize(torch.tensor([0.5, -1.0, 3.0])).numpy()
    for i, (n, s, m) in enumerate(((256, 1.0, 0.0), (256, 4.0, 12.0), (31, 0.05, -3.0), (1, 1.0, 7.0))):
        x = torch.tensor((rng.standard_normal(n) * s + m).astype(np.float32))
        y = vn.normalize(x)
        probe = torch.tensor(rng.standard_normal(8).astype(np.float32))
        out[f"in{i}"], out[f"norm{i}"], out[f"probe{i}"] = x.numpy(), y.numpy(), probe.numpy()
        out[f"denorm{i}"] = vn.denormalize(probe).numpy()
        out[f"norm_noupdate{i}"] = vn.normalize(probe, update_stats=False).numpy()
        rs = vn.running_stats
        out[f"state{i}"] = np.array([rs.mean, rs.variance, rs.count], dtype=np.float64)
    return out


def gen_g10_icm():
    """networks/ppo_networks/icm.py:22-430: ICM.forward (intrinsic reward, inverse loss, forward loss) and the
    gradients of the training loss ppo.py:2547-2548 w.r.t. every parameter; small widths keep the fixture small."""
    from ppo_and_friends.networks.ppo_networks.icm import ICM
    from gymnasium.spaces import Box, Discrete
    out = {}
    for tag, (O, space, kw) in {
            "disc": (6, Discrete(3), dict(encoded_obs_dim=32, encoder_hidden_size=32, inverse_hidden_size=32, forward_hidden_size=32)),
            "cont": (17, Box(-1.0, 1.0, (6,), np.float32), dict(encoded_obs_dim=32, encoder_hidden_size=64, inverse_hidden_size=32,
                                                                forward_hidden_size=32, inverse_hidden_depth=3, forward_hidden_depth=1))}.items():
        torch.manual_seed(10)
        icm = ICM(obs_space=Box(-np.inf, np.inf, (O,), np.float32), action_space=space, name="icm", **kw)
        n = 40
        o1, o2 = torch.randn(n, O), torch.randn(n, O)
        act = torch.randint(0, 3, (n, 1)) if tag == "disc" else torch.tanh(torch.randn(n, 6))
        intr, inv_loss, f_loss = icm(o1, o2, act)
        loss = (1.0 - 0.8) * f_loss + 0.8 * inv_loss
        grads = torch.autograd.grad(loss, list(icm.parameters()))
        for k, v in icm.state_dict().items():
            out[f"{tag}_p_{k}"] = v.detach().numpy().copy()
        out[f"{tag}_names"] = np.array([k for k, _ in icm.named_parameters()])
        for (k, _), gr in zip(icm.named_parameters(), grads):
            out[f"{tag}_g_{k}"] = gr.numpy()
        out.update({f"{tag}_obs1": o1.numpy(), f"{tag}_obs2": o2.numpy(), f"{tag}_actions": act.numpy(),
                    f"{tag}_intr": intr.detach().numpy(), f"{tag}_losses": np.array([inv_loss.item(), f_loss.item(), loss.item()])})
    return out


def gen_g13_filters():
    """environments/filter_wrappers.py:113-719 stand-alone: 2 agents x 5 envs through ObservationNormalizer ->
    ObservationClipper -> RewardNormalizer (quirk Q3) -> RewardClipper (wrapper_utils.py:81-111) for 12 steps with
    terminations, then a second pass after a hard reset."""
    from ppo_and_friends.environments.wrapper_utils import wrap_environment
    from gymnasium.spaces import Discrete
    TableEnv = table_env_class()
    E, T, A, O = 5, 12, 2, 3
    tables = make_tables(13, T, E, A, O, "uniform", 0.15, obs_scale=2.5, obs_shift=-1.0)
    counter = {"n": 0}
    pmap = lambda a: "agent"

    def env_generator():
        e = max(counter["n"] - 1, 0)
        counter["n"] += 1
        return TableEnv(tables, e, Discrete(2), A, critic_view="policy", policy_mapping_fn=pmap)

    env = wrap_environment(env_generator, pmap, {}, envs_per_proc=E, random_seed=1, normalize_obs=True,
                           normalize_rewards=True, obs_clip=(-1.5, 1.5), reward_clip=(-1.0, 1.0), gamma=0.99)
    out = {"obs_table": tables[0], "reward_table": tables[1], "term_table": tables[2], "cfg": np.array([E, T, A, O])}
    agents = [f"agent{a}" for a in range(A)]
    stack = lambda d: np.stack([np.asarray(d[a]) for a in agents], 1)
    for p in range(2):
        obs, cobs = env.reset()
        f_obs, f_cobs, f_rew, nat = [stack(obs)], [stack(cobs)], [], []
        for t in range(T):
            acts = {a: np.zeros((E, 1), dtype=np.int64) for a in agents}
            obs, cobs, rew, term, trunc, info = env.step(acts)
            f_obs.append(stack(obs)); f_cobs.append(stack(cobs)); f_rew.append(stack(rew)[..., 0])
            nat.append(np.stack([[float(info[a][e]["natural reward"]) for a in agents] for e in range(E)]))
        out[f"p{p}_obs"], out[f"p{p}_critic_obs"] = np.stack(f_obs), np.stack(f_cobs)      # [T+1, E, A, .]
        out[f"p{p}_rewards"], out[f"p{p}_natural"] = np.stack(f_rew), np.stack(nat)        # [T, E, A]
    w = env
    while w is not None and hasattr(w, "env"):
        cls = type(w).__name__
        for attr in ("actor_running_stats", "critic_running_stats", "running_stats"):
            val = getattr(w, attr, None)
            if isinstance(val, dict):
                for a, rs in val.items():
                    out[f"{cls}_{attr}_{a}_mean"] = np.asarray(rs.mean, dtype=np.float64)
                    out[f"{cls}_{attr}_{a}_var"] = np.asarray(rs.variance, dtype=np.float64)
                    out[f"{cls}_{attr}_{a}_count"] = np.array([rs.count], dtype=np.float64)
        w = w.env
    return out


UNIT_FIXTURES = {"g8_distributions": gen_g8_distributions, "g9_value_normalizer": gen_g9_value_normalizer,
                 "g10_icm": gen_g10_icm, "g13_filters": gen_g13_filters}


def scenarios():
    from gymnasium.spaces import Box, Discrete
    import torch.nn as nn
    leaky = lambda: {"activation": nn.LeakyReLU(), "hidden_size": 128}
    big = lambda: {"activation": nn.LeakyReLU(), "hidden_size": 256}
    sc = {}
    # C1/C2 layer shapes: CartPole dims, Discrete(2), 128^3 ReLU; terminations, bootstraps at the rollout end
    sc["g12_c2_term"] = dict(seed=101, E=8, T=32, A=1, O=4, action_space=Discrete(2), reward="ones", term_prob=0.06,
                             batch_size=64, epochs=2, iterations=2)
    # same, no terminations, episodes cut every 8 steps (all envs aligned), tail mini-batch (N=264 % 64 != 0 -> E=11,T=24)
    sc["g12_c2_cut"] = dict(seed=102, E=11, T=24, A=1, O=4, action_space=Discrete(2), reward="uniform", term_prob=0.0,
                            max_ts_per_ep=8, batch_size=64, epochs=3, iterations=1)
    # C4 layer shapes (MAPPO): 3 agents share one policy, actor 128^3 on O=18 -> Discrete(5), critic 256^3 on the
    # concatenated "policy" view (O_c = 54), LeakyReLU (baselines/pettingzoo/mpe_simple_spread.py:40-75)
    sc["g12_c4_mappo"] = dict(seed=104, E=4, T=16, A=3, O=18, action_space=Discrete(5), reward="uniform", term_prob=0.05,
                              batch_size=48, epochs=2, iterations=1, critic_view="policy",
                              policy_args=dict(actor_kw_args=leaky(), critic_kw_args=big()))
    # C3 layer shapes (HalfCheetah-v4: O=17, Box(6) in [-1,1], LeakyReLU actor 128^3 / critic 256^3, lr 1e-4;
    # baselines/gymnasium/half_cheetah.py:20-49): the tanh-Gaussian head alone, with terminations
    cheetah = lambda: Box(-1.0, 1.0, (6,), np.float32)
    sc["g12_c3_gauss"] = dict(seed=105, E=6, T=16, A=1, O=17, action_space=cheetah(), reward="uniform", term_prob=0.05,
                              batch_size=32, epochs=2, iterations=1,
                              policy_args=dict(actor_kw_args=leaky(), critic_kw_args=big(), lr=1e-4))
    # ... and the full C3 stack: ICM + observation / reward normalisers + clippers (tight clip ranges so that they
    # bite), episodes cut every 8 steps with all envs aligned (the shapes quirks Q1 / Q2 are well defined for)
    sc["g12_c3_full"] = dict(seed=106, E=6, T=16, A=1, O=17, action_space=cheetah(), reward="uniform", term_prob=0.0,
                             max_ts_per_ep=8, batch_size=32, epochs=2, iterations=2, obs_scale=3.0, obs_shift=1.0,
                             policy_args=dict(actor_kw_args=leaky(), critic_kw_args=big(), lr=1e-4, enable_icm=True),
                             ppo_args=dict(normalize_obs=True, normalize_rewards=True, obs_clip=(-2.0, 2.0),
                                           reward_clip=(-1.5, 1.5)))
    # per-dimension action bounds (networks/distributions.py:476-483, 580-609): small networks, unequal Box sides
    sc["g12_gauss_bounds"] = dict(seed=107, E=5, T=12, A=1, O=7, reward="uniform", term_prob=0.0, batch_size=20,
                                  epochs=1, iterations=1,
                                  action_space=Box(np.array([-1.0, -2.0, 0.0], np.float32), np.array([1.0, 2.0, 5.0], np.float32)),
                                  policy_args=dict(actor_kw_args={"hidden_size": 32}, critic_kw_args={"hidden_size": 32}))
    # discrete ICM at CartPole dims (ICM defaults: encoder O->128^3->128, inverse 256->128^2->n, forward 128+n->128^2->128)
    sc["g12_c2_icm"] = dict(seed=108, E=6, T=16, A=1, O=4, action_space=Discrete(2), reward="ones", term_prob=0.0,
                            max_ts_per_ep=8, batch_size=32, epochs=2, iterations=2,
                            policy_args=dict(enable_icm=True))
    # C5: MATPolicy (embedding 64, 1 block, 1 head), 3 agents, critic view "local"
    # (baselines/pettingzoo/mpe_simple_spread.py:40-92); fixed-length shared episodes
    from ppo_and_friends.policies.mat_policy import MATPolicy
    import ppo_and_friends.networks.actor_critic.multi_agent_transformer as mat
    # envs_per_proc = 1: the reference's autoregressive rollout assigns [E,1,1] tensors into [E,1] slices
    # (policies/mat_policy.py:476-479), which only broadcasts for E = 1 (quirk Q13)
    sc["g12_c5_mat"] = dict(seed=109, E=1, T=40, A=3, O=18, action_space=Discrete(5), reward="uniform", term_prob=0.0,
                            batch_size=16, epochs=2, iterations=2, critic_view="local", policy_class=MATPolicy,
                            ac_network=mat.MATActorCritic, policy_args=dict(mat_kw_args={"embedding size": 64}))
    # LSTM actor / critic (networks/ppo_networks/lstm.py:13-127): sequence windows with terminal masks, hidden-state
    # hand-over and write-back (episode_info.py:775-809,954-987; ppo.py:2312-2319,2450-2466)
    from ppo_and_friends.networks.ppo_networks.lstm import LSTMNetwork
    lstm_kw = lambda S: dict(sequence_length=S, lstm_hidden_size=32, ff_hidden_size=32)
    sc["g12_lstm_term"] = dict(seed=110, E=5, T=20, A=1, O=4, action_space=Discrete(2), reward="uniform", term_prob=0.08,
                               batch_size=24, epochs=2, iterations=2, ac_network=LSTMNetwork,
                               policy_args=dict(actor_kw_args=lstm_kw(4), critic_kw_args=lstm_kw(4)))
    # ---- the metric's own mini-batch shape: batch_size = 256 (the reference's default, ppo.py:134) -- 16 row tiles per
    # network in K12, the multi-tile layered critic, 86 K15 tiles; two full mini-batches per epoch (+ a tail at C4)
    sc["g12_c2_b256"] = dict(seed=121, E=16, T=32, A=1, O=4, action_space=Discrete(2), reward="ones", term_prob=0.06,
                             batch_size=256, epochs=2, iterations=1)
    sc["g12_c4_b256"] = dict(seed=124, E=6, T=32, A=3, O=18, action_space=Discrete(5), reward="uniform", term_prob=0.05,
                             batch_size=256, epochs=2, iterations=1, critic_view="policy",
                             policy_args=dict(actor_kw_args=leaky(), critic_kw_args=big()))
    sc["g12_c3_b256"] = dict(seed=126, E=16, T=32, A=1, O=17, action_space=cheetah(), reward="uniform", term_prob=0.0,
                             max_ts_per_ep=8, batch_size=256, epochs=2, iterations=1, obs_scale=3.0, obs_shift=1.0,
                             policy_args=dict(actor_kw_args=leaky(), critic_kw_args=big(), lr=1e-4, enable_icm=True),
                             ppo_args=dict(normalize_obs=True, normalize_rewards=True, obs_clip=(-2.0, 2.0),
                                           reward_clip=(-1.5, 1.5)))
    # MAT: envs_per_proc must be 1 in the reference (quirk Q13), so the 256-row mini-batch comes from a long rollout
    sc["g12_c5_b256"] = dict(seed=129, E=1, T=512, A=3, O=18, action_space=Discrete(5), reward="uniform", term_prob=0.0,
                             max_ts_per_ep=512, batch_size=256, epochs=2, iterations=1, critic_view="local", policy_class=MATPolicy,
                             ac_network=mat.MATActorCritic, policy_args=dict(mat_kw_args={"embedding size": 64}))
    # ---- KL early stop (ppo.py:2221-2232): 4 epochs allowed, a learning rate / target_kl pair under which the reference
    # leaves the epoch loop early (`epochs_run` records after how many epochs, per iteration); with ICM the ICM pass of
    # the stopping epoch still runs before the break (ppo.py:2213-2214 precede the test)
    sc["g12_c2_klstop"] = dict(seed=131, E=16, T=32, A=1, O=4, action_space=Discrete(2), reward="ones", term_prob=0.06,
                               batch_size=64, epochs=4, iterations=2, policy_args=dict(target_kl=0.01, lr=1e-3))
    sc["g12_c2_icm_klstop"] = dict(seed=131, E=16, T=32, A=1, O=4, action_space=Discrete(2), reward="ones", term_prob=0.06,
                                   batch_size=64, epochs=4, iterations=2,
                                   policy_args=dict(target_kl=0.005, lr=3e-3, enable_icm=True))
    sc["g12_lstm_cut"] = dict(seed=111, E=4, T=18, A=1, O=4, action_space=Discrete(3), reward="uniform", term_prob=0.0,
                              max_ts_per_ep=6, batch_size=16, epochs=2, iterations=1, ac_network=LSTMNetwork,
                              policy_args=dict(actor_kw_args=lstm_kw(3), critic_kw_args=lstm_kw(3)))
    return sc


def rank_scenarios():
    """R = 2 ranks of the unmodified reference under the two-process mpi4py stand-in (ref_import.py): what pins the DD-PPO
    arithmetic -- mpi_avg_gradients (utils/mpi_utils.py:50-111), the value normaliser's per-mini-batch allgather of raw
    data (utils/stats.py:47-50), the per-epoch all-reduced statistics that drive the KL early stop (ppo.py:2468-2475,
    2221-2232), the rank-0 broadcast of the initial weights (ppo_policy.py:457-471).  One fixture holds both ranks
    (keys r0.* / r1.*)."""
    from gymnasium.spaces import Discrete
    import torch.nn as nn
    leaky = lambda: {"activation": nn.LeakyReLU(), "hidden_size": 128}
    big = lambda: {"activation": nn.LeakyReLU(), "hidden_size": 256}
    sc = {}
    sc["g12_c2_r2"] = dict(seed=141, E=16, T=32, A=1, O=4, action_space=Discrete(2), reward="ones", term_prob=0.06,
                           batch_size=256, epochs=2, iterations=2)
    sc["g12_c4_r2"] = dict(seed=144, E=6, T=32, A=3, O=18, action_space=Discrete(5), reward="uniform", term_prob=0.05,
                           batch_size=256, epochs=2, iterations=1, critic_view="policy",
                           policy_args=dict(actor_kw_args=leaky(), critic_kw_args=big()))
    # ICM (mpi_avg_gradients(icm_model), ppo.py:2559) + KL early stop decided on all-reduced totals, on two ranks
    sc["g12_c2_icm_r2_klstop"] = dict(seed=148, E=16, T=32, A=1, O=4, action_space=Discrete(2), reward="ones", term_prob=0.06,
                                      batch_size=64, epochs=4, iterations=2,
                                      policy_args=dict(target_kl=RANK_KL_TARGET, lr=3e-3, enable_icm=True))
    return sc


RANK_KL_TARGET = float(os.environ.get("PPOAF_GOLDEN_KL_TARGET", "0.005"))
RANKS = 2


def run_rank_scenario(name):
    """Parent: start RANKS children of this script (one reference process per rank), merge their arrays."""
    import subprocess
    tmp = tempfile.mkdtemp(prefix="ppoaf_golden_ranks_")
    addr = os.path.join(tmp, "hub.sock")
    procs = []
    for r in range(RANKS):
        env = dict(os.environ, PYTHONHASHSEED="0", PPOAF_FAKE_MPI_RANK=str(r), PPOAF_FAKE_MPI_SIZE=str(RANKS), PPOAF_FAKE_MPI_ADDR=addr)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--rank-child", name, os.path.join(tmp, f"rank{r}.npz")],
                                      env=env, stdout=subprocess.DEVNULL))
    for p_ in procs:
        if p_.wait() != 0:
            raise RuntimeError(f"{name}: a rank failed")
    out = {"ranks": np.array([RANKS], dtype=np.int64)}
    for r in range(RANKS):
        with np.load(os.path.join(tmp, f"rank{r}.npz")) as g:
            out.update({f"r{r}.{k}": g[k] for k in g.files})
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return out


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--rank-child":
        scratch = ref_import.make_scratch()
        try:
            out = run_scenario(sys.argv[2], rank=int(os.environ["PPOAF_FAKE_MPI_RANK"]), **rank_scenarios()[sys.argv[2]])
            np.savez_compressed(sys.argv[3], **out)
        finally:
            ref_import.drop_scratch(scratch)
        return
    only = set(sys.argv[1:])
    scratch = ref_import.make_scratch()
    try:
        for name, fn in UNIT_FIXTURES.items():
            if only and name not in only:
                continue
            out = fn()
            path = os.path.join(HERE, name + ".npz")
            np.savez_compressed(path, **out)
            print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} B")
        for name, cfg in scenarios().items():
            if only and name not in only:
                continue
            out = run_scenario(name, **cfg)
            path = os.path.join(HERE, name + ".npz")
            np.savez_compressed(path, **out)
            print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} B")
        for name in rank_scenarios():
            if only and name not in only:
                continue
            out = run_rank_scenario(name)
            path = os.path.join(HERE, name + ".npz")
            np.savez_compressed(path, **out)
            print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} B")
    finally:
        ref_import.drop_scratch(scratch)


if __name__ == "__main__":
    main()
