"""
Observation / reward filter wrappers for batched device environments -- the stand-ins for
environments/filter_wrappers.py:22-719 of the reference (ObservationNormalizer, ObservationClipper,
RewardNormalizer, RewardClipper) and the IdentityWrapper they derive from
(environments/ppo_env_wrappers.py:24-199).

The classes keep the reference's names, constructor arguments and save/load file names, and they
are stacked the same way (`wrap_environment`, environments/wrapper_utils.py:81-111).  What differs
is the execution: the stack shares ONE filter plan, and a step of the outermost wrapper runs the
raw env step followed by the two K13 launches (moments -> [all-gather of a small float64 record
across ranks] -> apply) that update every running statistic, normalise and clip observations,
critic observations and rewards of all agents at once.  The reference walks the same stack with a
Python call per wrapper per agent, an allgather of the raw data per statistic, and -- for rewards --
one statistics update per env inside a loop (quirk Q3; reproduced in closed form by the kernel).

Env contract (environments/synthetic.py):
    obs, critic_obs, reward, terminated, truncated, terminal_obs = env.step(action)
with rows agent-major ([A*E, .]); `terminal_obs` stays unfiltered as the reference's
info["terminal observation"] does (it is written below the filters, ppo_env_wrappers.py:1128-1137).
"""
import os
import pickle

import numpy as np
import torch

from .. import kernels as K
from ..utils import mpi_utils
from ..utils.reference_io import dump_running_stats, load_running_stats


class IdentityWrapper:
    """ppo_env_wrappers.py:24-199: forwards everything to the wrapped env."""

    def __init__(self, env, test_mode=False, **kw_args):
        self.env = env
        self.test_mode = test_mode
        self.finalized = False

    def __getattr__(self, name):          # observation_space, agent_ids, get_batch_size, device, ...
        if name == "env":
            raise AttributeError(name)
        return getattr(self.env, name)

    def step(self, action):
        return self.env.step(action)

    def reset(self):
        return self.env.reset()

    def soft_reset(self):
        return self.env.soft_reset()

    def finalize(self, status_dict):
        self.finalized = True
        inner = getattr(self.env, "finalize", None)
        if callable(inner):
            inner(status_dict)

    def save_info(self, path):
        inner = getattr(self.env, "save_info", None)
        if callable(inner):
            inner(path)

    def load_info(self, path):
        inner = getattr(self.env, "load_info", None)
        if callable(inner):
            inner(path)


class _FilterPlan:
    """The fused state of a stack of filter wrappers over one raw env."""

    def __init__(self, raw_env):
        self.raw = raw_env
        self.obs_norm = None       # dict(update, eps)
        self.obs_clip = None       # (lo callable, hi callable)
        self.rew_norm = None       # dict(update, eps, gamma)
        self.rew_clip = None
        self._ready = False
        self._raw_obs = None
        self.natural_reward = None

    # ---- configuration (called by the wrapper constructors, innermost first) ----
    def add(self, kind, cfg):
        # wrapper_utils.py:81-111 is the only order the fused pass implements
        if kind == "obs_norm" and self.obs_clip is not None:
            raise NotImplementedError("ObservationNormalizer must wrap the env before ObservationClipper")
        if kind == "rew_norm" and self.rew_clip is not None:
            raise NotImplementedError("RewardNormalizer must wrap the env before RewardClipper")
        if getattr(self, kind) is not None:
            raise ValueError(f"the env already has a {kind} filter")
        setattr(self, kind, cfg)
        self._ready = False

    def _setup(self):
        raw = self.raw
        self.device = torch.device(raw.device)
        self.G = int(getattr(raw, "num_agents", 1))
        self.n = int(raw.get_batch_size())
        self.W_o = int(np.prod(raw.observation_space.shape))
        self.W_c = int(np.prod(raw.critic_observation_space.shape))
        G, n, dev = self.G, self.n, self.device
        if self.obs_norm is not None and "stats" not in self.obs_norm:
            mk = lambda W: (torch.zeros(G * W, dtype=torch.float32, device=dev),
                            torch.ones(G * W, dtype=torch.float32, device=dev),
                            torch.full((G * W,), 1e-4, dtype=torch.float64, device=dev))   # stats.py:25-27
            self.obs_norm["stats"] = mk(self.W_o)
            self.obs_norm["critic_stats"] = mk(self.W_c)
        if self.rew_norm is not None and "state" not in self.rew_norm:
            self.rew_norm["state"] = (torch.zeros(G * n, dtype=torch.float64, device=dev),   # :385-387
                                      torch.zeros(G, dtype=torch.float64, device=dev),
                                      torch.ones(G, dtype=torch.float64, device=dev),
                                      torch.full((G,), 1e-4, dtype=torch.float64, device=dev))
        self.filters_obs = self.obs_norm is not None or self.obs_clip is not None
        self.filters_rew = self.rew_norm is not None or self.rew_clip is not None
        self._record = torch.zeros(K.env_filter_record_len(G, self.W_o if self.filters_obs else 0,
                                                           self.W_c if self.filters_obs else 0, self.filters_rew),
                                   dtype=torch.float64, device=dev)
        self._ready = True

    @staticmethod
    def _range(clip):
        return None if clip is None else (clip[0](), clip[1]())

    # ---- the fused pass ----
    def _run(self, obs, critic_obs, reward=None, terminated=None, truncated=None):
        if not self._ready:
            self._setup()
        G, n = self.G, self.n
        of = cf = rf = None
        out_o, out_c, out_r = obs, critic_obs, reward
        updating = False
        if self.filters_obs:
            obs = obs.reshape(G * n, self.W_o).contiguous()
            critic_obs = critic_obs.reshape(G * n, self.W_c).contiguous()
            out_o, out_c = torch.empty_like(obs), torch.empty_like(critic_obs)
            cfg = self.obs_norm
            clip = self._range(self.obs_clip)
            upd = bool(cfg and cfg["update"])
            eps = cfg["eps"] if cfg else 1e-8
            of = K.obs_filter(obs, out_o, G, n, cfg["stats"] if cfg else None, upd, clip, eps)
            cf = K.obs_filter(critic_obs, out_c, G, n, cfg["critic_stats"] if cfg else None, upd, clip, eps)
            updating |= upd
        if reward is not None and self.filters_rew:
            reward = reward.reshape(G * n).contiguous()
            out_r = torch.empty_like(reward)
            cfg = self.rew_norm
            upd = bool(cfg and cfg["update"])
            rf = K.reward_filter(reward, terminated.contiguous() if cfg else None,
                                 truncated.contiguous() if (cfg and truncated is not None) else None,
                                 out_r, G, n, cfg["state"] if cfg else None, upd, self._range(self.rew_clip),
                                 cfg["gamma"] if cfg else 0.99, cfg["eps"] if cfg else 1e-8)
            updating |= upd
        if of is None and rf is None:
            return out_o, out_c, out_r
        records = None
        if updating:
            # records are laid out for the streams of THIS call (reset has no reward stream)
            K.env_filter_moments(of, cf, rf, G, n, self._record)
            need = K.env_filter_record_len(G, self.W_o if of is not None else 0, self.W_c if cf is not None else 0,
                                           rf is not None)
            records = self._record[:need]
            if mpi_utils.distributed_path():
                records = mpi_utils.allgather_records(records).reshape(-1)      # stats.py:47-50
        K.env_filter_apply(of, cf, rf, G, n, records)
        return out_o, out_c, out_r

    def step(self, action):
        obs, cobs, reward, terminated, truncated, term_obs = self.raw.step(action)
        self._raw_obs = (obs, cobs)
        self.natural_reward = reward            # info["natural reward"], filter_wrappers.py:426-437
        obs, cobs, reward = self._run(obs, cobs, reward, terminated, truncated)
        if self.filters_obs and term_obs is not None:
            # the "next observation" the reference logs (ppo.py:1725-1745): the FILTERED next observation,
            # except for terminated envs, whose info["terminal observation"] was written below the filters
            # (ppo_env_wrappers.py:1128-1137) and stays raw
            term_obs = torch.where(terminated.reshape(-1, 1), term_obs.reshape(obs.shape), obs)
        return obs, cobs, reward, terminated, truncated, term_obs

    def reset(self):
        obs, cobs = self.raw.reset()
        self._raw_obs = (obs, cobs)
        obs, cobs, _ = self._run(obs, cobs)
        return obs, cobs

    def soft_reset(self):
        """ppo_env_wrappers.py:149-199: the raw cached observation is filtered (and counted) again."""
        if self._raw_obs is None:
            return self.reset()
        soft = getattr(self.raw, "soft_reset", None)
        obs, cobs = soft() if callable(soft) else self._raw_obs
        self._raw_obs = (obs, cobs)
        obs, cobs, _ = self._run(obs, cobs)
        return obs, cobs


class _DeviceFilter(IdentityWrapper):
    """Base of the four filters: joins (or starts) the fused plan of the env it wraps."""

    def __init__(self, env, **kw_args):
        super().__init__(env, **kw_args)
        self._plan = env._plan if isinstance(env, _DeviceFilter) else _FilterPlan(env)

    def step(self, action):
        return self._plan.step(action)

    def reset(self):
        return self._plan.reset()

    def soft_reset(self):
        return self._plan.soft_reset()

    @property
    def natural_reward(self):
        """The unfiltered reward of the last step (info["natural reward"] of the reference)."""
        return self._plan.natural_reward


def _stats_to_host(G, W, mean, var, count, agent_ids):
    m, v, c = mean.cpu().numpy().reshape(G, W), var.cpu().numpy().reshape(G, W), count.cpu().numpy().reshape(G, W)
    return {a: {"mean": m[i].copy(), "variance": v[i].copy(), "count": float(c[i, 0])}
            for i, a in enumerate(agent_ids)}


def _stats_from_host(d, G, W, mean, var, count, agent_ids):
    for i, a in enumerate(agent_ids):
        if a not in d:
            continue
        mean.view(G, W)[i].copy_(torch.as_tensor(np.asarray(d[a]["mean"], dtype=np.float32).reshape(W)))
        var.view(G, W)[i].copy_(torch.as_tensor(np.asarray(d[a]["variance"], dtype=np.float32).reshape(W)))
        count.view(G, W)[i].fill_(float(d[a]["count"]))


def _rank_file(path, stem, test_mode):
    """<stem>_<rank>.pickle with the rank-0 fallback of filter_wrappers.py:313-340."""
    r = 0 if test_mode else mpi_utils.get_rank()
    f = os.path.join(path, f"{stem}_{r}.pickle")
    return f if os.path.exists(f) else os.path.join(path, f"{stem}_0.pickle")


class ObservationNormalizer(_DeviceFilter):
    """filter_wrappers.py:113-340."""

    def __init__(self, env, update_stats=True, epsilon=1e-8, **kw_args):
        super().__init__(env, **kw_args)
        self.update_stats = update_stats
        self.epsilon = epsilon
        self._cfg = {"update": bool(update_stats), "eps": float(epsilon)}
        self._plan.add("obs_norm", self._cfg)

    def _agent_ids(self):
        return list(getattr(self._plan.raw, "agent_ids", ["agent0"]))

    @property
    def actor_running_stats(self):
        p = self._plan
        if not p._ready:
            p._setup()
        return _stats_to_host(p.G, p.W_o, *self._cfg["stats"], self._agent_ids())

    @property
    def critic_running_stats(self):
        p = self._plan
        if not p._ready:
            p._setup()
        return _stats_to_host(p.G, p.W_c, *self._cfg["critic_stats"], self._agent_ids())

    def save_info(self, path):
        """filter_wrappers.py:296-311 (file names kept; the payload is plain numpy state per agent)."""
        if not self.test_mode:
            r = mpi_utils.get_rank()
            for stem, st in (("ActorRunningObsStats", self.actor_running_stats),
                             ("CriticRunningObsStats", self.critic_running_stats)):
                with open(os.path.join(path, f"{stem}_{r}.pickle"), "wb") as fh:
                    dump_running_stats(st, fh)
        super().save_info(path)

    def load_info(self, path):
        p = self._plan
        if not p._ready:
            p._setup()
        for stem, key, W in (("ActorRunningObsStats", "stats", p.W_o), ("CriticRunningObsStats", "critic_stats", p.W_c)):
            with open(_rank_file(path, stem, self.test_mode), "rb") as fh:
                _stats_from_host(load_running_stats(fh), p.G, W, *self._cfg[key], self._agent_ids())
        super().load_info(path)


class RewardNormalizer(_DeviceFilter):
    """filter_wrappers.py:342-521."""

    def __init__(self, env, update_stats=True, epsilon=1e-8, gamma=0.99, **kw_args):
        super().__init__(env, **kw_args)
        self.update_stats = update_stats
        self.epsilon = epsilon
        self.gamma = gamma
        self._cfg = {"update": bool(update_stats), "eps": float(epsilon), "gamma": float(gamma)}
        self._plan.add("rew_norm", self._cfg)

    @property
    def running_stats(self):
        p = self._plan
        if not p._ready:
            p._setup()
        _, mean, var, count = self._cfg["state"]
        ids = list(getattr(p.raw, "agent_ids", ["agent0"]))
        m, v, c = mean.cpu().numpy(), var.cpu().numpy(), count.cpu().numpy()
        return {a: {"mean": float(m[i]), "variance": float(v[i]), "count": float(c[i])} for i, a in enumerate(ids)}

    @property
    def running_reward(self):
        p = self._plan
        if not p._ready:
            p._setup()
        return self._cfg["state"][0].view(p.G, p.n)

    def save_info(self, path):
        if not self.test_mode:
            with open(os.path.join(path, f"RunningRewardsStats_{mpi_utils.get_rank()}.pickle"), "wb") as fh:
                dump_running_stats(self.running_stats, fh)
        super().save_info(path)

    def load_info(self, path):
        p = self._plan
        if not p._ready:
            p._setup()
        with open(_rank_file(path, "RunningRewardsStats", self.test_mode), "rb") as fh:
            d = load_running_stats(fh)
        _, mean, var, count = self._cfg["state"]
        for i, a in enumerate(getattr(p.raw, "agent_ids", ["agent0"])):
            if a in d:
                mean[i], var[i], count[i] = float(d[a]["mean"]), float(d[a]["variance"]), float(d[a]["count"])
        super().load_info(path)


class GenericClipper(_DeviceFilter):
    """filter_wrappers.py:523-597: bounds may be numbers or status-driven callables."""

    def __init__(self, env, clip_range=(-10., 10.), **kw_args):
        from ..policies.ppo_policy import CallableValue
        super().__init__(env, **kw_args)
        self.clip_range = tuple(c if callable(c) else CallableValue(c) for c in clip_range)

    def finalize(self, status_dict):
        self.clip_range[0].finalize(status_dict)
        self.clip_range[1].finalize(status_dict)
        super().finalize(status_dict)

    def get_clip_range(self):
        return self.clip_range[0](), self.clip_range[1]()


class ObservationClipper(GenericClipper):
    """filter_wrappers.py:617-661."""

    def __init__(self, env, clip_range=(-10., 10.), **kw_args):
        super().__init__(env, clip_range=clip_range, **kw_args)
        self._plan.add("obs_clip", self.clip_range)


class RewardClipper(GenericClipper):
    """filter_wrappers.py:663-719."""

    def __init__(self, env, clip_range=(-10., 10.), **kw_args):
        super().__init__(env, clip_range=clip_range, **kw_args)
        self._plan.add("rew_clip", self.clip_range)


def wrap_environment(env_generator, normalize_obs=True, normalize_rewards=True, obs_clip=None,
                     reward_clip=None, gamma=0.99, test_mode=False, **_unused):
    """
    environments/wrapper_utils.py:8-113 for a batched device env: `env_generator()` already returns
    the vectorised env (envs_per_proc is its batch size), so only the filter stack is applied, in the
    reference's order.  Arguments the reference uses for its Python vectorisation are accepted and ignored.
    """
    env = env_generator()
    if normalize_obs:
        env = ObservationNormalizer(env=env, test_mode=test_mode, update_stats=not test_mode)
    if obs_clip is not None and type(obs_clip) == tuple:
        env = ObservationClipper(env=env, test_mode=test_mode, clip_range=obs_clip)
    if normalize_rewards:
        env = RewardNormalizer(env=env, test_mode=test_mode, update_stats=not test_mode, gamma=gamma)
    if reward_clip is not None and type(reward_clip) == tuple:
        env = RewardClipper(env=env, test_mode=test_mode, clip_range=reward_clip)
    return env
