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
                 ppo_args=None, obs_scale=1.0, obs_shift=0.0, ac_network=None):
    from ppo_and_friends.ppo import PPO
    from ppo_and_friends.networks.ppo_networks.feed_forward import FeedForwardNetwork
    from torch.utils.data import DataLoader

    TableEnv = table_env_class()
    tables = make_tables(seed, T, E, A, O, reward, term_prob, obs_scale, obs_shift)
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
    torch.manual_seed(seed)
    np.random.seed(seed)
    kw = dict(device="cpu", random_seed=seed, envs_per_proc=E, max_ts_per_ep=max_ts_per_ep, batch_size=batch_size,
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
        rec["datasets"].append({k: v.detach().cpu().numpy().copy() for k, v in d.items()} | {"ep_lens": np.array(ds.ep_lens)})

    pol.finalize_dataset = rec_finalize

    orig_update = pol.update_weights

    def rec_update(actor_loss, critic_loss):
        if len(rec["mb"]) < 1:                                  # first mini-batch: losses + raw (unclipped) gradients
            ga = torch.autograd.grad(actor_loss, [p for p in pol.actor.parameters() if p.requires_grad],
                                     retain_graph=True, allow_unused=True)
            gc = torch.autograd.grad(critic_loss, [p for p in pol.critic.parameters() if p.requires_grad],
                                     retain_graph=True, allow_unused=True)
            rec["mb"].append(dict(
                actor_loss=float(actor_loss.item()), critic_loss=float(critic_loss.item()),
                actor_grads=[np.zeros(0, np.float32) if g is None else g.numpy().copy() for g in ga],
                critic_grads=[np.zeros(0, np.float32) if g is None else g.numpy().copy() for g in gc]))
        return orig_update(actor_loss, critic_loss)

    pol.update_weights = rec_update

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
        rec["epochs"].append(dict(kind="ppo", perm=perm, stats=np.array(
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

    ppo.rollout = rec_rollout

    ppo.learn(E * T * iterations)

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
    return sc


def main():
    only = set(sys.argv[1:])
    scratch = ref_import.make_scratch()
    try:
        for name, cfg in scenarios().items():
            if only and name not in only:
                continue
            out = run_scenario(name, **cfg)
            path = os.path.join(HERE, name + ".npz")
            np.savez_compressed(path, **out)
            print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} B")
    finally:
        ref_import.drop_scratch(scratch)


if __name__ == "__main__":
    main()
