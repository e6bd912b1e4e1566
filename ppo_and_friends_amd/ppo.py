"""
PPO trainer for the hot path: rollout() -> dataset build (GAE) -> epochs of
mini-batch updates, the stand-in for ppo.py:124-2567 of the reference
(PPO.rollout :1534-2110, PPO.learn :2112-2272, PPO._ppo_batch_train :2274-2485).

Device-first differences
  * the environment is batched and device-resident (environments/synthetic.py);
    a rollout step never leaves the GPU: actor/critic inference, sampling,
    buffer writes and episode bookkeeping are enqueued on one stream;
  * every trajectory scan of the rollout is one HIP launch at finalize;
  * one mini-batch update = gather (K4) + value-normaliser update (K5) +
    evaluate + loss fwd/bwd (K2+K3) + backward + [RCCL all-reduce] + fused
    clip/Adam (K11) on static buffers, captured once in a hipGraph and
    replayed; statistics stay on the device until the epoch ends (the KL early
    stop is the only host read per epoch);
  * the ranks meet in ONE gradient all-reduce per mini-batch (actor and critic
    buckets are adjacent) and one all-gather of the value-normaliser moment
    records per epoch -- the reference issues 16 pickled all-reduces, an
    all-gather of raw data and a barrier per mini-batch.
"""
import os
import pickle
import sys
import time
from collections import OrderedDict

import numpy as np
import torch

from . import kernels as K
from .environments.filter_wrappers import wrap_environment
from .policies.ppo_policy import CallableValue, PPOPolicy
from .policies.utils import generate_policy
from .utils import mpi_utils
from .utils.misc import RunningStatNormalizer
from .utils.mpi_utils import rank_print


class PPOLossFunction(torch.autograd.Function):
    """K2+K3 as an autograd node: (logp, entropy, values) -> (actor_loss, critic_loss, scalars[8])."""

    @staticmethod
    def forward(ctx, cur_logp, entropy, values, old_logp, adv, rtg, normalize_adv, surr_clip,
                entropy_weight, kl_loss_weight, use_huber):
        sc, dlp, dent, dval = K.ppo_loss_fwd_bwd(
            cur_logp.reshape(-1).contiguous(), old_logp.reshape(-1).contiguous(),
            adv.reshape(-1).contiguous(), entropy.reshape(-1).contiguous(),
            values.reshape(-1).contiguous(), rtg.reshape(-1).contiguous(),
            normalize_adv, surr_clip, entropy_weight, kl_loss_weight, use_huber, 10.0)
        ctx.save_for_backward(dlp, dent, dval)
        ctx.shapes = (cur_logp.shape, entropy.shape, values.shape)
        return sc[K.SC_ACTOR], sc[K.SC_CRITIC], sc

    @staticmethod
    def backward(ctx, g_actor, g_critic, g_sc):
        dlp, dent, dval = ctx.saved_tensors
        s0, s1, s2 = ctx.shapes
        return ((dlp * g_actor).reshape(s0), (dent * g_actor).reshape(s1), (dval * g_critic).reshape(s2),
                None, None, None, None, None, None, None, None)


class PermutationLoader:
    """
    DataLoader(dataset, batch_size, shuffle=True) (ppo.py:2181-2184) reduced to
    what it does here: a fresh torch.randperm per epoch (host generator, so a
    seeded reference run can be replayed) cut into batch_size slices.
    """

    def __init__(self, dataset, batch_size, generator=None, prefetch_cache=None):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.generator = generator
        # {"n": N, "perm": tensor}: the NEXT draw of `generator`, made early (see prefetch())
        self.prefetch_cache = prefetch_cache if prefetch_cache is not None else {}

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def epoch_permutation(self):
        """
        The index order one `for batch in DataLoader(..., shuffle=True)` pass uses,
        drawing from the host RNG exactly as torch's DataLoader does so a seeded
        run can be replayed: the iterator first draws its base seed, then
        RandomSampler draws the permutation (from a fresh generator seeded off the
        global RNG when no generator was given).
        """
        n = len(self.dataset)
        c = self.prefetch_cache
        if c.get("n") == n and c.get("perm") is not None:
            perm, c["perm"] = c["perm"], None
        else:
            self._discard_prefetch()
            perm = self._draw(n)
        # the staging buffer is reused by the next draw: the device copy is made (and completed) here
        return perm.to(self.dataset.device, non_blocking=False)

    def _discard_prefetch(self):
        """A permutation drawn ahead for another dataset size (the next policy's dataset differs): put the
        generator back where the reference's would be, so the stream of draws stays the reference's."""
        c = self.prefetch_cache
        if c.get("perm") is not None and c.get("state_before") is not None and self.generator is not None:
            self.generator.set_state(c["state_before"])
        c["perm"], c["state_before"] = None, None

    def _draw(self, n):
        g = self.generator
        torch.empty((), dtype=torch.int64).random_(generator=g)          # _BaseDataLoaderIter base seed
        if g is None:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            g = torch.Generator().manual_seed(seed)
        # drawn in ordinary (cached) host memory, then copied with one sequential pass into a pinned staging buffer that
        # is allocated once (allocating pinned memory synchronises with the device and would serialise the prefetch
        # behind the running epoch).
        c = self.prefetch_cache
        buf, work = c.get("pinned"), c.get("work")
        if buf is None or buf.numel() != n:
            buf = torch.empty(n, dtype=torch.int64, pin_memory=torch.cuda.is_available())
            work = torch.empty(n, dtype=torch.int64)
            c["pinned"], c["work"] = buf, work
        # ONE intra-op thread for the draw: the shuffle is sequential anyway (1.8 ms for 524 288 indices with one thread,
        # 5 ms with torch's default pool), and with the default pool on a host whose CPU quota is smaller than its core
        # count (128 OpenMP threads on a 16-CPU share of the GPU boxes) single statements of this function stalled
        # for 30-90 ms at random -- longer than the epoch the prefetch hides behind, which made whole training steps
        # prefetch-bound (tools/probes/prefetch_probe.py).  The reference pins its thread count too (mpi_utils.py:37-48).
        nt = torch.get_num_threads()
        if nt > 1:
            torch.set_num_threads(1)
        try:
            torch.randperm(n, generator=g, out=work)
            buf.copy_(work)
            torch.randperm(n, generator=g, out=work)   # RandomSampler's trailing `randperm(n)[:num_samples % n]` draw (discarded)
        finally:
            if nt > 1:
                torch.set_num_threads(nt)
        return buf

    def prefetch(self):
        """
        Draw the generator's NEXT permutation now (while the GPU is busy with the current epoch).  The
        draw order of the generator is unchanged -- the next epoch_permutation() call, whoever makes it,
        receives exactly this draw -- so seeded runs replay as before.  Only for a dedicated generator.
        """
        if self.generator is None:
            return
        c = self.prefetch_cache
        n = len(self.dataset)
        if c.get("perm") is None or c.get("n") != n:
            self._discard_prefetch()
            c["state_before"] = self.generator.get_state()
            c["n"], c["perm"] = n, self._draw(n)

    def __iter__(self):
        perm = self.epoch_permutation()
        for o in range(0, perm.numel(), self.batch_size):
            yield perm[o:o + self.batch_size]


class PPO:

    def __init__(self, env_generator, policy_settings, policy_mapping_fn=None, device="cuda",
                 random_seed=None, envs_per_proc=1, max_ts_per_ep=200, batch_size=256,
                 ts_per_rollout=1024, gamma=0.99, epochs_per_iter=10, ext_reward_weight=1.0,
                 normalize_adv=True, normalize_obs=True, normalize_rewards=True,
                 normalize_values=True, obs_clip=None, reward_clip=None, recalc_advantages=False,
                 soft_resets=False, state_path="./saved_state", load_state=False, checkpoint_every=100,
                 save_train_scores=False, save_avg_ep_len=False, save_running_time=False, save_bs_info=False,
                 save_state=True, use_graphs=True, update_mode="auto", verbose=False, freeze_scheduler=None,
                 reference_device="cpu", **kw_args):
        """
        ppo.py:126-167.  `ts_per_rollout` is per environment (ppo.py:317-318
        multiplies by envs_per_proc).  normalize_obs / normalize_rewards / obs_clip / reward_clip
        put the device filter stack of environments/filter_wrappers.py around `env_generator()`
        (ppo.py:358-371 -> wrapper_utils.py:81-111); soft_resets as ppo.py:1580-1586.
        """
        mpi_utils.set_torch_threads()
        self.device = torch.device(device)
        # Which of the reference's two behaviours to reproduce where they DIFFER by the device the reference itself runs
        # on: "cpu" (its default device, ppo.py:130, the path north_star compares against and the one every fixture was
        # recorded on) or "cuda".  One place so far, quirk Q12: on a CPU tensor `.detach().cpu().numpy()` is a VIEW
        # (ppo.py:1115-1141), on a CUDA tensor `.cpu()` copies.
        if reference_device not in ("cpu", "cuda"):
            raise ValueError(f"reference_device={reference_device!r}: 'cpu' or 'cuda'")
        self.reference_device = reference_device
        self.envs_per_proc = int(envs_per_proc)
        self.ts_per_rollout = int(ts_per_rollout) * self.envs_per_proc
        self.max_ts_per_ep = int(max_ts_per_ep)
        self.batch_size = int(batch_size)
        self.epochs_per_iter = int(epochs_per_iter)
        self.normalize_adv = normalize_adv
        self.normalize_values = normalize_values
        self.recalc_advantages = recalc_advantages
        self.ext_reward_weight = ext_reward_weight
        self.use_graphs = use_graphs and self.device.type == "cuda" and os.environ.get("PPOAF_GRAPHS", "1") != "0"
        # "fused": K12 kernels (MLP policies); "torch": torch-ROCm MLPs + K2..K11; "auto": fused when covered
        self.update_mode = update_mode
        self._fused = {}
        self._perm_cache = {}
        self.verbose = verbose
        self.random_seed = 0 if random_seed is None else int(random_seed)
        rank = mpi_utils.get_rank()
        # seed + rank, as ppoaf_cli.py:419 does
        torch.manual_seed(self.random_seed + rank)
        np.random.seed(self.random_seed + rank)
        self.loader_generator = torch.Generator().manual_seed(self.random_seed + rank)

        self.env = wrap_environment(env_generator, normalize_obs=normalize_obs,
                                    normalize_rewards=normalize_rewards, obs_clip=obs_clip,
                                    reward_clip=reward_clip, gamma=gamma, test_mode=False)
        self.soft_resets = soft_resets if callable(soft_resets) else CallableValue(bool(soft_resets))
        self.policy_mapping_fn = policy_mapping_fn or (lambda agent_id: next(iter(policy_settings)))
        max_int = int(np.iinfo(np.int32).max)
        self.status_dict = OrderedDict()                     # keys and initial values of ppo.py:455-518
        self.status_dict["global status"] = OrderedDict([
            ("iteration", 0), ("rollout time", 0.0), ("train time", 0.0), ("running time", 0.0), ("timesteps", 0),
            ("total episodes", 0), ("longest episode", 0), ("shortest episode", max_int), ("average episode", 0)])
        self.state_path = state_path
        self.env_info_path = os.path.join(state_path, "env_info")
        self.curve_path = os.path.join(state_path, "curves")
        self.checkpoint_every = int(checkpoint_every)
        self.save_state = bool(save_state)                   # False: learn() writes nothing (tests, benchmarks)
        self.save_train_scores, self.save_avg_ep_len = save_train_scores, save_avg_ep_len
        self.save_running_time, self.save_bs_info = save_running_time, save_bs_info
        self.normalize_obs = bool(normalize_obs)
        self.policies = {}
        self.value_normalizers = {}
        for policy_id, settings in policy_settings.items():
            # the reference's 5-tuple (policy_class, obs, critic_obs, action, args), ppo.py:329-345
            policy_class, obs_space, critic_obs_space, act_space, policy_args = \
                settings if len(settings) == 5 else (None,) + tuple(settings)
            policy_args = dict(policy_args)
            policy_args.setdefault("gamma", gamma)
            pol = generate_policy(policy_name=policy_id, policy_class=policy_class,
                                  actor_observation_space=obs_space,
                                  critic_observation_space=critic_obs_space, action_space=act_space,
                                  test_mode=False, envs_per_proc=self.envs_per_proc,
                                  random_seed=self.random_seed, **policy_args)
            for agent_id in getattr(self.env, "agent_ids", ["agent0"]):
                if self.policy_mapping_fn(agent_id) == policy_id:
                    pol.register_agent(agent_id)
            self.policies[policy_id] = pol
            self.status_dict[policy_id] = OrderedDict([
                ("score avg", 0), ("natural score avg", 0), ("top score", -max_int), ("weighted entropy", 0),
                ("actor loss", 0), ("critic loss", 0), ("kl avg", 0), ("natural reward range", (max_int, -max_int)),
                ("top natural reward", -max_int), ("reward range", (max_int, -max_int)),
                ("bootstrap range", (max_int, -max_int)), ("bootstrap avg", "N/A"), ("obs range", (max_int, -max_int)),
                ("frozen", False)])
            if dict(policy_args).get("enable_icm", False):
                self.status_dict[policy_id].update({"icm loss": 0, "intrinsic score avg": 0,
                                                    "intr reward range": (max_int, -max_int)})
            if normalize_values:
                self.value_normalizers[policy_id] = RunningStatNormalizer(
                    name=f"{policy_id}-value_normalizer", device=self.device)
        for pol in self.policies.values():
            pol.finalize(self.status_dict, self.device)
            pol.fused_icm_reward = update_mode != "torch"       # K14's kernels for the rollout-time intrinsic reward
        # ppo.py:663-678: freeze cycling over policy groups (utils/schedulers.py:FreezeCyclingScheduler)
        from .utils.schedulers import FreezeCyclingScheduler
        if freeze_scheduler is None:
            freeze_scheduler = CallableValue(None)
        elif not isinstance(freeze_scheduler, FreezeCyclingScheduler):
            raise TypeError(f"freeze_scheduler must be a FreezeCyclingScheduler, got {type(freeze_scheduler)}")
        self.freeze_scheduler = freeze_scheduler
        self.freeze_scheduler.finalize(self.state_path, self.status_dict, self.policies)
        if load_state:
            self.freeze_scheduler.load_info()
        # ppo.py:325-354: policies may post-process what the environment returns
        self.have_policy_step_constraints = any(p.have_step_constraints for p in self.policies.values())
        self.have_policy_reset_constraints = any(p.have_reset_constraints for p in self.policies.values())
        self.soft_resets.finalize(self.status_dict)
        if callable(getattr(self.env, "finalize", None)):
            self.env.finalize(self.status_dict)          # status-driven clip ranges (filter_wrappers.py:560-566)
        self._graphs = {}
        self._obs = None
        if load_state and os.path.exists(os.path.join(state_path, "state_0.pickle")):      # ppo.py:521-545
            saved = self.load(state_path, "latest")
            for k in self.status_dict:
                if k in saved:
                    self.status_dict[k].update(saved[k])

    # ------------------------------------------------------------------ rollout
    def get_policy_values(self, policy_id, critic_obs):
        """ppo.py:1030-1075 + get_denormalized_values :1143-1167."""
        with torch.no_grad():
            v = self.policies[policy_id].get_critic_values(critic_obs)
            v = v.reshape(v.shape[:-1]).contiguous()          # [rows] or [rows, A]
            if self.normalize_values:
                v = self.value_normalizers[policy_id].denormalize(v)
        return v

    def rollout(self):
        """
        ppo.py:1534-2110 for a batched device environment and one policy.
        Per step: actions / log-probs (actor + sampling kernel), values (critic,
        denormalised), env.step, buffer row write, episode-end flags -- all
        enqueued without a host read.  Episode-end cases (ppo.py:1795-1983):
        terminated -> terminal end; ep_ts == max_ts_per_ep, truncated, or the
        last step of the rollout -> bootstrapped end with the critic's value of
        the next observation.
        """
        if len(self.policies) > 1:
            return self._rollout_multi_policy()
        start = time.time()
        policy_id = next(iter(self.policies))
        pol = self.policies[policy_id]
        pol.initialize_dataset()
        pol.eval()
        env = self.env
        n_envs = env.get_batch_size()
        T = self.ts_per_rollout // n_envs
        pol.initialize_episodes(n_envs, self.status_dict, ts_per_rollout=self.ts_per_rollout)
        buf = pol.buffer
        E = buf.C                      # rows per step: agents x envs, agent-major (ppo.py:710-795)
        # ppo.py:1580-1586: a hard reset per rollout unless soft_resets (then the env carries on, and
        # a filter stack counts the carried observation once more, ppo_env_wrappers.py:149-199)
        if self._obs is None or not self.soft_resets():
            obs, critic_obs = self.apply_policy_reset_constraints(*env.reset())
        else:
            soft = getattr(env, "soft_reset", None)
            obs, critic_obs = soft() if callable(soft) else self._obs
        ep_ts = torch.zeros(E, dtype=torch.int32, device=self.device)
        may_end_early = getattr(env, "term_table", True) is not None or self.max_ts_per_ep < T
        fused_step = (self.update_mode != "torch" and self.device.type == "cuda"
                      and pol.fused_step_unsupported_reason() == "")
        vn = self.value_normalizers[policy_id] if self.normalize_values else None
        grouped = pol.agent_grouping
        if grouped:
            # MAT: rows are envs, agents side by side in the policy's (shuffled) slot order
            # (ppo.py:1643-1644, 753-770): [A*E, .] agent-major env tensors -> [E, A, .]
            pol.shuffle_agent_ids()
            order = torch.as_tensor(pol.agent_slot_order(), device=self.device)
            n_ag = order.numel()
            group = lambda x: x.reshape((n_ag, n_envs) + tuple(x.shape[1:]))[order].transpose(0, 1).contiguous()
            ungroup = lambda x: x.transpose(0, 1)[torch.argsort(order)].reshape((n_ag * n_envs,) + tuple(x.shape[2:]))
        # per-step intrinsic rewards in the buffer's own layout ([T, E*A] rows of agents side by side when grouped)
        intr_buf = torch.zeros(T, E * (pol.num_agents if grouped else 1), dtype=torch.float32,
                               device=self.device) if pol.enable_icm else None

        def grouped_intrinsic(prev_obs, prev_cobs, nxt_obs, nxt_cobs, act_env):
            """ppo.py:1219-1288 for an agent-grouped policy: agent-major [A*E] intrinsic rewards (one shared value
            per env repeated for its agents, or one per (agent, env) row); observations = the actor's view."""
            o1, o2 = (prev_cobs, nxt_cobs) if pol.expanded_actor_space else (prev_obs, nxt_obs)
            if pol.agent_shared_icm:
                return pol.get_agent_shared_intrinsic_rewards(o1, o2, act_env).repeat(n_ag)
            return pol.get_intrinsic_reward(o1, o2, act_env)

        if pol.enable_icm:
            may_end_early = True          # bootstrap rewards carry the "surprise" term: dense end table
        # replay of a recorded rollout: `self.replay_raw_actions` ([T, rows, .] device tensor in the buffer's row
        # layout, or None) makes every step log the recorded raw action instead of sampling one
        rec = getattr(self, "replay_raw_actions", None)
        replay = (lambda t: None) if rec is None else (lambda t: rec[t].contiguous())
        nat_buf = self._scratch(f"nat_buf_{T}_{env.num_agents if hasattr(env, 'num_agents') else 1}_{n_envs}",
                                T * (E if not grouped else E * pol.num_agents), torch.float32).view(T, -1)
        for t in range(T):
            if fused_step and grouped:
                # K16: encoder + autoregressive decoder passes + sampling + values + the buffer row in one launch
                g_obs, g_cobs = group(obs), group(critic_obs)
                action = pol.rollout_step(t, g_cobs, g_cobs if pol.expanded_actor_space else g_obs, vn, forced_raw_action=replay(t))
                act_env = ungroup(action)
                nxt_obs, nxt_cobs, reward, terminated, truncated, term_obs = self.apply_policy_step_constraints(*env.step(act_env))
                nat_buf[t].copy_(self._natural_reward(env, reward))
                if self.ext_reward_weight != 1.0:
                    reward = reward * self.ext_reward_weight
                if pol.enable_icm:
                    intr = grouped_intrinsic(obs, critic_obs, nxt_obs, nxt_cobs, act_env)
                    intr_buf[t] = group(intr).reshape(-1)
                    reward = reward + intr
                pol.finish_step(t, group(reward), group(nxt_cobs if pol.expanded_actor_space else term_obs) if pol.enable_icm else None)
                terminated, truncated = terminated[:n_envs], truncated[:n_envs]   # agents of an env end together
            elif fused_step:
                # K6+K7: inference, sampling, log-probs, values and the buffer row in one launch
                action = pol.rollout_step(t, obs.contiguous(), critic_obs.contiguous(), vn, forced_raw_action=replay(t))
                nxt_obs, nxt_cobs, reward, terminated, truncated, term_obs = self.apply_policy_step_constraints(*env.step(action))
                nat_buf[t].copy_(self._natural_reward(env, reward))
                if self.ext_reward_weight != 1.0:
                    reward = reward * self.ext_reward_weight
                if pol.enable_icm:            # ppo.py:1719-1723 -> apply_intrinsic_rewards :1219-1288
                    intr_buf[t] = pol.get_intrinsic_reward(obs, nxt_obs, action)   # ppo.py:1719-1723: the post-step observation
                    reward = reward + intr_buf[t]
                pol.finish_step(t, reward, term_obs)
            elif grouped:
                g_obs, g_cobs = group(obs), group(critic_obs)
                raw_action, action, log_prob = pol.get_rollout_actions(g_cobs if pol.expanded_actor_space else g_obs, forced_raw_action=replay(t))
                value = self.get_policy_values(policy_id, g_cobs)
                act_env = ungroup(action)
                nxt_obs, nxt_cobs, reward, terminated, truncated, term_obs = self.apply_policy_step_constraints(*env.step(act_env))
                nat_buf[t].copy_(self._natural_reward(env, reward))
                if self.ext_reward_weight != 1.0:
                    reward = reward * self.ext_reward_weight
                if pol.enable_icm:
                    intr = grouped_intrinsic(obs, critic_obs, nxt_obs, nxt_cobs, act_env)
                    intr_buf[t] = group(intr).reshape(-1)
                    reward = reward + intr
                buf.write_step(t, slice(0, E), g_cobs, g_cobs if pol.expanded_actor_space else g_obs,
                               group(nxt_cobs if pol.expanded_actor_space else term_obs) if pol.enable_icm else None,
                               raw_action, action, value, log_prob, group(reward))
                pol._t = t + 1
                terminated, truncated = terminated[:n_envs], truncated[:n_envs]   # agents of an env end together
            else:
                raw_action, action, log_prob = pol.get_rollout_actions(obs, forced_raw_action=replay(t))
                value = self.get_policy_values(policy_id, critic_obs)
                nxt_obs, nxt_cobs, reward, terminated, truncated, term_obs = self.apply_policy_step_constraints(*env.step(action))
                nat_buf[t].copy_(self._natural_reward(env, reward))
                if self.ext_reward_weight != 1.0:
                    reward = reward * self.ext_reward_weight
                if pol.enable_icm:
                    intr_buf[t] = pol.get_intrinsic_reward(obs, nxt_obs, action)   # ppo.py:1719-1723: the post-step observation
                    reward = reward + intr_buf[t]
                buf.write_step(t, slice(0, E), critic_obs, obs, term_obs, raw_action, action, value,
                               log_prob, reward)
                pol._t = t + 1
                if pol.using_lstm:
                    pol.store_hidden_states(t, terminated)
            if may_end_early:
                ep_ts += 1
                last = t == T - 1
                boot = (~terminated) & ((ep_ts >= self.max_ts_per_ep) | truncated | last)
                if pol.using_lstm and not last:
                    # ppo.py:1863-1881: whenever an episode is cut (max length / truncation) the reference
                    # evaluates the critic on the NEXT observation for the whole batch, which also steps the
                    # stateful LSTM critic once more.  Branch-free here: the extra step is computed every
                    # time and kept only if the reference would have taken it.
                    cut = (((ep_ts == self.max_ts_per_ep) & ~terminated).any() | truncated.any())   # ep_ts of terminated envs is 0 by then (:1851)
                    h_old = tuple(x.clone() for x in pol.critic.hidden_state)
                    v_next = self.get_policy_values(policy_id, nxt_cobs)
                    pol.critic.hidden_state = tuple(torch.where(cut, n, o) for n, o in zip(pol.critic.hidden_state, h_old))
                    buf.boot_value[t].copy_(v_next)
                buf.end_kind[t] = torch.where(terminated, 1, torch.where(boot, 2, 0)).to(torch.int8)
                ep_ts = torch.where(terminated | boot, torch.zeros_like(ep_ts), ep_ts)
                buf.fixed_length = False        # (also at the last row: an env may TERMINATE there -- ts_per_rollout = 1 made it visible)
            obs, critic_obs = nxt_obs, nxt_cobs
        # bootstrap values: V(next obs).  For ends before the last row the next
        # observation's value is the value logged at t+1 (same critic, same
        # normaliser state during a rollout); the last row needs one more pass.
        next_value = self.get_policy_values(policy_id, group(critic_obs) if grouped else critic_obs)
        if may_end_early:
            if not pol.using_lstm:           # stateless critics: V(next obs) is the value logged at t + 1
                buf.boot_value[:-1].copy_(buf.values[1:])
            buf.boot_value[T - 1].copy_(next_value)
            buf.boot_reward.copy_(buf.boot_value)
            buf.boot_stats = None
            if pol.enable_icm:
                # the "bootstrap range / avg" statistics read next_reward BEFORE the surprise lands (ppo.py:1901-1912)
                buf.boot_stats = buf.boot_value.clone()
                # ppo.py:1926-1930: bootstrap reward += intrinsic reward of the step - "intrinsic score avg"
                ism = float(self.status_dict[policy_id].get("intrinsic score avg", 0.0))
                buf.boot_reward.add_((intr_buf - ism).view_as(buf.boot_reward))
                # quirk Q12 (pinned by fixtures g12_c2_icm / g12_c3_full / g12_c3_b256, all recorded on the reference's
                # CPU device): next_reward is a numpy VIEW of the next_value tensor there (ppo.py:1115-1141), so its
                # in-place `+=` puts the surprise into the ending VALUE of the GAE as well.  On a CUDA device the
                # reference's `.cpu()` copies and the ending value stays V(next obs): reference_device="cuda".
                if self.reference_device == "cpu":
                    buf.boot_value.copy_(buf.boot_reward)
        else:
            buf.boot_stats = None
            buf.end_kind[T - 1].fill_(2)
            buf.boot_value[T - 1].copy_(next_value)
            buf.boot_reward[T - 1].copy_(next_value)
        self._obs = (obs, critic_obs)
        pol.finalize_dataset()
        self._publish_rollout_statistics(policy_id, buf, nat_buf, intr_buf, n_envs, T, obs, grouped)
        gs = self.status_dict["global status"]
        gs["timesteps"] += self.ts_per_rollout * mpi_utils.get_num_procs()     # env steps (ppo.py:1653), not agent steps
        torch.cuda.synchronize() if self.device.type == "cuda" else None
        gs["rollout time"] = time.time() - start
        return pol.dataset

    def _rollout_multi_policy(self):
        """
        ppo.py:1534-2110 with several policies (`policy_mapping_fn` partitions the env's agents, ppo.py:329-345,
        710-858): every policy acts on its own agents' rows of the agent-major env tensors and logs into its own
        rollout buffer; the env is stepped once with the combined actions; episode ends are shared (the agents of an
        env end together).  Covers feed-forward PPOPolicy instances (independent or team-wise PPO).  The env hands over
        either agent-major tensors [A*E, .] (all agents share shapes) or, as the reference's multi-agent wrappers do,
        dicts keyed by agent id (agents of different policies may then differ in observation / action space; the
        agents of ONE policy share them, as in the reference).  Grouped (MAT), LSTM and ICM policies are
        single-policy features here.
        """
        start = time.time()
        env = self.env
        n_envs = env.get_batch_size()
        T = self.ts_per_rollout // n_envs
        agent_ids = list(env.agent_ids)
        A = len(agent_ids)
        ctxs = []
        for policy_id, pol in self.policies.items():
            if pol.agent_grouping or pol.using_lstm or pol.enable_icm:
                raise NotImplementedError("several policies in one run: feed-forward PPOPolicy without ICM only")
            pol.initialize_dataset()
            pol.eval()
            pol.initialize_episodes(n_envs, self.status_dict, ts_per_rollout=self.ts_per_rollout)
            idx = torch.as_tensor(sorted(agent_ids.index(a) for a in pol.agent_ids), device=self.device)
            fused = (self.update_mode != "torch" and self.device.type == "cuda"
                     and pol.fused_step_unsupported_reason() == "")
            ctxs.append(dict(id=policy_id, pol=pol, buf=pol.buffer, idx=idx, n=int(idx.numel()), fused=fused,
                             agents=[agent_ids[int(i)] for i in idx],          # the policy's agents in env order
                             vn=self.value_normalizers[policy_id] if self.normalize_values else None,
                             nat=torch.zeros(T, int(idx.numel()) * n_envs, dtype=torch.float32, device=self.device)))
        if sorted(int(i) for c in ctxs for i in c["idx"]) != list(range(A)):
            raise ValueError("policy_mapping_fn must assign every agent of the env to exactly one policy")
        if self._obs is None or not self.soft_resets():
            obs, critic_obs = self.apply_policy_reset_constraints(*env.reset())
        else:
            soft = getattr(env, "soft_reset", None)
            obs, critic_obs = soft() if callable(soft) else self._obs
        dict_env = isinstance(obs, dict)

        def rows(x, c):
            """The policy's rows of an env quantity, agent-major: [n * E, .]."""
            if isinstance(x, dict):
                return torch.cat([x[a] for a in c["agents"]], 0)
            return x.reshape((A, n_envs) + tuple(x.shape[1:]))[c["idx"]].reshape((c["n"] * n_envs,) + tuple(x.shape[1:]))

        per_env = lambda x: (x[agent_ids[0]] if isinstance(x, dict) else x[:n_envs])   # agents of an env end together
        ep_ts = torch.zeros(n_envs, dtype=torch.int32, device=self.device)
        may_end_early = getattr(env, "term_table", True) is not None or self.max_ts_per_ep < T
        actions = None
        for t in range(T):
            for c in ctxs:
                pol, o, co = c["pol"], rows(obs, c).contiguous(), rows(critic_obs, c).contiguous()
                if c["fused"]:
                    a = pol.rollout_step(t, o, co, c["vn"])
                else:
                    raw, a, lp = pol.get_rollout_actions(o)
                    c["pending"] = (co, o, raw, a, self.get_policy_values(c["id"], co), lp)
                if dict_env:
                    if actions is None:
                        actions = {}
                    per_agent = a.reshape((c["n"], n_envs) + tuple(a.shape[1:]))
                    for i, agent_id in enumerate(c["agents"]):
                        actions[agent_id] = per_agent[i]
                    continue
                if actions is None:
                    actions = torch.zeros((A, n_envs) + tuple(a.shape[1:]), dtype=a.dtype, device=self.device)
                elif a.dtype != actions.dtype or tuple(a.shape[1:]) != tuple(actions.shape[2:]):
                    raise ValueError(
                        f"policy {c['id']} produces actions {a.dtype}{tuple(a.shape[1:])} but the env's combined action tensor "
                        f"holds {actions.dtype}{tuple(actions.shape[2:])}: policies with different action spaces need an env "
                        "that exchanges dicts keyed by agent id (environments/synthetic.py:SyntheticMixedAgentsEnv)")
                actions[c["idx"]] = a.reshape((c["n"], n_envs) + tuple(a.shape[1:]))
            env_action = actions if dict_env else actions.reshape((A * n_envs,) + tuple(actions.shape[2:]))
            nxt_obs, nxt_cobs, reward, terminated, truncated, term_obs = self.apply_policy_step_constraints(*env.step(env_action))
            nat = self._natural_reward(env, reward)
            if self.ext_reward_weight != 1.0 and not dict_env:
                reward = reward * self.ext_reward_weight
            for c in ctxs:
                c["nat"][t].copy_(rows(nat, c))
                r = rows(reward, c)
                if self.ext_reward_weight != 1.0 and dict_env:
                    r = r * self.ext_reward_weight
                if c["fused"]:
                    c["pol"].finish_step(t, r, None)
                else:
                    co, o, raw, a, v, lp = c.pop("pending")
                    c["buf"].write_step(t, slice(0, c["buf"].C), co, o, None, raw, a, v, lp, r)
                    c["pol"]._t = t + 1
            if may_end_early:
                term_e, trunc_e = per_env(terminated), per_env(truncated)
                ep_ts += 1
                last = t == T - 1
                boot = (~term_e) & ((ep_ts >= self.max_ts_per_ep) | trunc_e | last)
                kind = torch.where(term_e, 1, torch.where(boot, 2, 0)).to(torch.int8)
                for c in ctxs:
                    c["buf"].end_kind[t] = kind.repeat(c["n"])
                    c["buf"].fixed_length = False
                ep_ts = torch.where(term_e | boot, torch.zeros_like(ep_ts), ep_ts)
            obs, critic_obs = nxt_obs, nxt_cobs
        self._obs = (obs, critic_obs)
        gs = self.status_dict["global status"]
        episodes_before = gs["total episodes"]
        for c in ctxs:
            buf = c["buf"]
            next_value = self.get_policy_values(c["id"], rows(critic_obs, c).contiguous())
            if may_end_early:
                buf.boot_value[:-1].copy_(buf.values[1:])
                buf.boot_value[T - 1].copy_(next_value)
                buf.boot_reward.copy_(buf.boot_value)
            else:
                buf.end_kind[T - 1].fill_(2)
                buf.boot_value[T - 1].copy_(next_value)
                buf.boot_reward[T - 1].copy_(next_value)
            c["pol"].finalize_dataset()
            self._publish_rollout_statistics(c["id"], buf, c["nat"], None, n_envs, T, rows(obs, c), False)
        # every policy's statistics pass added the (shared, per-env) episode count: keep it once (ppo.py:2089)
        gs["total episodes"] = episodes_before + (gs["total episodes"] - episodes_before) / len(ctxs)
        gs["timesteps"] += self.ts_per_rollout * mpi_utils.get_num_procs()
        torch.cuda.synchronize() if self.device.type == "cuda" else None
        gs["rollout time"] = time.time() - start
        return {c["id"]: c["pol"].dataset for c in ctxs}

    @staticmethod
    def _natural_reward(env, reward):
        """info["natural reward"] of the filter wrappers, else the env's own reward (ppo.py:1689-1693)."""
        nat = getattr(env, "natural_reward", None)
        return reward if nat is None else nat

    def _publish_rollout_statistics(self, policy_id, buf, nat_buf, intr_buf, n_envs, T, last_obs, grouped):
        """ppo.py:1978-2099 (see utils/rollout_stats.py) -> status_dict."""
        from .utils.rollout_stats import rollout_statistics
        pol = self.policies[policy_id]
        A = nat_buf.shape[1] // n_envs
        per_env = (lambda x: x.reshape(T, n_envs, A).sum(2)) if grouped else (lambda x: x.reshape(T, A, n_envs).sum(1))
        nat_env = nat_buf.view(T, A, n_envs).sum(1)                      # the env hands agent-major columns
        ek = buf.end_kind[:, :n_envs]
        mm = lambda x: torch.aminmax(x)
        o_lo, o_hi = mm(last_obs)
        if T > 1:
            b_lo, b_hi = mm(buf.observations[1:])
            o_lo, o_hi = torch.minimum(o_lo, b_lo), torch.maximum(o_hi, b_hi)
        boot_stats = getattr(buf, "boot_stats", None)
        st = rollout_statistics(per_env(buf.rewards), nat_env, ek == 1, ek == 2,
                                buf.boot_reward if boot_stats is None else boot_stats, A, mm(buf.rewards),
                                mm(nat_buf), (o_lo, o_hi), self.ts_per_rollout / self.envs_per_proc,
                                None if intr_buf is None else per_env(intr_buf),
                                None if intr_buf is None else mm(intr_buf))
        sd, gs = self.status_dict[policy_id], self.status_dict["global status"]
        if not self.normalize_obs:                                       # ppo.py:2010-2017: a running range
            st["obs range"] = (min(sd["obs range"][0], st["obs range"][0]), max(sd["obs range"][1], st["obs range"][1]))
        sd["top natural reward"] = max(sd["top natural reward"], st["natural reward range"][1])
        for k in ("score avg", "natural score avg", "top score", "obs range", "reward range", "natural reward range",
                  "bootstrap range", "bootstrap avg", "intrinsic score avg", "intr reward range"):
            if k in st:
                sd[k] = st[k]
        sd["frozen"] = pol.frozen
        gs["total episodes"] += st["total episodes"]
        for k in ("longest episode", "shortest episode", "average episode"):
            gs[k] = st[k]

    # ------------------------------------------------------------------ update
    def learn(self, num_timesteps):
        """ppo.py:2112-2272."""
        gs = self.status_dict["global status"]
        ts_max = gs["timesteps"] + num_timesteps
        best = {policy_id: -np.inf for policy_id in self.policies}
        iter_start = iter_stop = time.time()
        while gs["timesteps"] < ts_max:
            self.freeze_scheduler()                               # ppo.py:2139
            pre_rollout_timesteps = gs["timesteps"]
            self.rollout()
            for policy_id in self.policies:                       # ppo.py:2144-2150
                score = self.status_dict[policy_id]["natural score avg"]
                if score >= best[policy_id]:
                    best[policy_id] = score
                    self.save(tag=f"{policy_id}_best")
            gs["running time"] += (iter_stop - iter_start) + gs["rollout time"]
            iter_start = time.time()
            if self.verbose:
                self.print_status()
            self.save()
            if gs["iteration"] % self.checkpoint_every == 0:
                self.save(tag=str(gs["iteration"]))
            self._save_curves(pre_rollout_timesteps)
            self.train_on_rollout()
            gs["iteration"] += 1
            for policy_id, pol in self.policies.items():        # ppo.py:1406-1426
                pol.update_learning_rate()
                sd = self.status_dict[policy_id]
                sd["lr"] = pol.lr()
                sd["entropy weight"] = pol.entropy_weight()
                if pol.enable_icm:
                    sd["icm lr"] = pol.icm_lr()
                    sd["intr reward weight"] = pol.intr_reward_weight()
            if sum(pol.lr() for pol in self.policies.values()) <= 0.0:
                rank_print("Learning rate has bottomed out. Terminating early")
                break
            iter_stop = time.time()

    def print_status(self):
        """ppo.py:1383-1404."""
        rank_print("\n--------------------------------------------------------")
        rank_print("Status Report:")
        for section, d in self.status_dict.items():
            rank_print("  {}:".format(section))
            for key, val in d.items():
                rank_print("    {}: {}".format(key, val))
        rank_print("--------------------------------------------------------")

    # ------------------------------------------------------------ state on disk
    def save(self, tag="latest"):
        """
        ppo.py:2569-2618: rank 0 writes `<state>/<policy>-policy/<tag>/...` (networks, optimisers),
        `<state>/env_info/<tag>/...` (filter and value-normaliser statistics) and `<state>/state_0.pickle`
        (the status dict), in the reference's file names and payload formats (utils/reference_io.py).
        """
        if not self.save_state:
            return
        if mpi_utils.get_rank() == 0:
            info = os.path.join(self.env_info_path, tag)
            os.makedirs(info, exist_ok=True)
            if callable(getattr(self.env, "save_info", None)):
                self.env.save_info(info)
            for pol in self.policies.values():
                pol.save(self.state_path, tag)
            for vn in self.value_normalizers.values():
                vn.save_info(info)
            with open(os.path.join(self.state_path, "state_0.pickle"), "wb") as fh:
                pickle.dump(self.status_dict, fh, protocol=pickle.HIGHEST_PROTOCOL)
            self.freeze_scheduler.save_info()                # ppo.py:2606
        if mpi_utils.distributed_path():
            torch.distributed.barrier()

    def apply_policy_reset_constraints(self, obs, critic_obs):
        """ppo.py:1468-1491.  In this build the environment hands over batched device tensors
        ([agents x envs, .], agent-major) instead of per-agent dicts; the hooks receive those."""
        if self.have_policy_reset_constraints:
            for pol in self.policies.values():
                obs, critic_obs = pol.apply_reset_constraints(obs, critic_obs)
        return obs, critic_obs

    def apply_policy_step_constraints(self, obs, critic_obs, reward, terminated, truncated, info):
        """ppo.py:1493-1532; `info` is this build's terminal-observation tensor (the only info field the path reads)."""
        if self.have_policy_step_constraints:
            for pol in self.policies.values():
                obs, critic_obs, reward, terminated, truncated, info = pol.apply_step_constraints(
                    obs, critic_obs, reward, terminated, truncated, info)
        return obs, critic_obs, reward, terminated, truncated, info

    def direct_load_policy(self, policy_id, policy_path):
        """ppo.py:2664-2686: weights straight from `<state>/<name>-policy/<tag>`, value normaliser from that tag."""
        self.policies[policy_id].direct_load(policy_path)
        tag = os.path.basename(os.path.dirname(os.path.join(os.path.abspath(policy_path), "")))
        if self.normalize_values and policy_id in self.value_normalizers:
            for t in (tag, "latest"):
                try:
                    self.value_normalizers[policy_id].load_info(os.path.join(self.env_info_path, t))
                    break
                except OSError:
                    continue

    def load_policies(self, state_path, tag):
        """ppo.py:2688-2701."""
        for policy_id in self.policies:
            self.load_policy(policy_id, state_path, tag)

    def load_status(self, state_path):
        with open(os.path.join(state_path, "state_0.pickle"), "rb") as fh:
            return pickle.load(fh)

    def load_policy(self, policy_id, state_path, tag="latest"):
        self.policies[policy_id].load(state_path, tag)
        if policy_id in self.value_normalizers:
            try:
                self.value_normalizers[policy_id].load_info(os.path.join(state_path, "env_info", tag))
            except OSError:
                pass

    def load(self, state_path, tag="latest"):
        """ppo.py:2704-2721."""
        for policy_id in self.policies:
            self.load_policy(policy_id, state_path, tag)
        info = os.path.join(state_path, "env_info", tag)
        if callable(getattr(self.env, "load_info", None)) and os.path.isdir(info):
            try:
                self.env.load_info(info)
            except OSError:
                pass
        return self.load_status(state_path)

    def _save_curves(self, timestep):
        """ppo.py:2723-2851: `<timestep> <value>` rows appended to the files `ppoaf plot` reads."""
        if not self.save_state or mpi_utils.get_rank() != 0:
            return
        gs = self.status_dict["global status"]

        def put(sub, name, value):
            d = os.path.join(self.curve_path, sub)
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, name), "ab") as fh:
                np.savetxt(fh, np.array([[timestep, value]], dtype=np.float64))

        for policy_id in self.policies:
            sd = self.status_dict[policy_id]
            if self.save_train_scores:
                put("scores", f"{policy_id}_scores.npy", sd["natural score avg"])
            if self.save_bs_info:
                put("bs_min", f"{policy_id}_bs_min.npy", sd["bootstrap range"][0])
                put("bs_max", f"{policy_id}_bs_max.npy", sd["bootstrap range"][1])
                put("bs_avg", f"{policy_id}_bs_avg.npy", sd["bootstrap avg"])
        if self.save_avg_ep_len:
            put("episode_length", "average_episode.npy", gs["average episode"])
        if self.save_running_time:
            put("runtime", "running_time.npy", gs["running time"])

    def train_on_rollout(self):
        """The epoch loop of ppo.py:2178-2249 (KL early stop :2222-2232)."""
        start = time.time()
        for policy_id, pol in self.policies.items():
            if pol.frozen:
                continue
            pol.train()
            loader = PermutationLoader(pol.dataset, self.batch_size, self.loader_generator, self._perm_cache)
            for epoch_idx in range(self.epochs_per_iter):
                if epoch_idx > 0 and self.recalc_advantages:
                    loader.dataset.recalculate_advantages()
                if not self._ppo_icm_epoch_overlapped(loader, policy_id):
                    self._ppo_batch_train(loader, policy_id)
                    if pol.enable_icm:
                        self._icm_batch_train(loader, policy_id)
                if self.status_dict[policy_id]["kl avg"] > pol.target_kl:
                    if self.verbose:
                        rank_print(f"Target KL of {pol.target_kl} has been reached. "
                                   f"Ending early (after {epoch_idx + 1} epochs)")
                    break
            pol.clear_dataset()
        self._guard_replicas()
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        self.status_dict["global status"]["train time"] = time.time() - start

    def _replicated_state(self):
        """Every tensor that synchronous DD-PPO keeps bitwise identical on all ranks: parameter buckets + Adam moments."""
        out = []
        for pol in self.policies.values():
            if getattr(pol, "agent_grouping", False) and hasattr(pol, "actor_critic"):
                opt = pol.actor_critic_optim
                out += [pol.actor_critic.flat_params, opt.exp_avg, opt.exp_avg_sq]
            elif hasattr(pol, "policy_exp_avg"):
                out += [pol.policy_params, pol.policy_exp_avg, pol.policy_exp_avg_sq]
            if pol.enable_icm and hasattr(pol.icm_model, "flat_params"):
                out += [pol.icm_model.flat_params, pol.icm_optim.exp_avg, pol.icm_optim.exp_avg_sq]
        return out

    def _guard_replicas(self):
        """
        Safety net of the K17 peer exchange (N > 1 only; one 8-byte all-gather per iteration).  The exchange
        is self-tested at start-up and sums in a fixed order, so replicas stay bitwise identical; should they
        ever differ (a peer mapping misbehaving in a way the self-test did not show), say so loudly, restore
        rank 0's state everywhere and continue on the RCCL all-reduce path.  Returns True if all was well.
        """
        if mpi_utils.get_num_procs() == 1:
            return True
        # every N > 1 path: on the all-reduce path each rank folds its own clip norm with atomics in an order of its
        # own, so the coefficient can differ in the last bit between ranks -- the same checksum catches that drift
        if mpi_utils.replicas_agree(self._replicated_state()):
            return True
        self._heal_replicas("replicas diverged")
        return False

    def _heal_replicas(self, why):
        """Collective (every rank reaches it through the same all-reduced verdict): rank 0's parameter and optimiser
        buckets everywhere, peer exchanges closed, RCCL all-reduce path from here on."""
        active = [f for f in getattr(self, "_fused", {}).values() if f is not None and getattr(f, "xchg", None) is not None]
        print(f"[ppo_and_friends_amd] rank {mpi_utils.get_rank()}: {why}; restoring one known-good state on every rank"
              + (" and switching from the peer gradient exchange to the RCCL all-reduce path" if active else ""),
              file=sys.stderr, flush=True)
        snap = getattr(self, "_epoch_snapshot", None)
        if snap is not None and "ran out of time" in why:
            # a timed-out exchange fed garbage gradients to the rest of that epoch on EVERY rank (rank 0 included):
            # go back to the state all ranks held when the epoch began, then make it rank 0's everywhere
            for t, keep in zip(self._replicated_state(), snap):
                t.copy_(keep)
        for t in self._replicated_state():
            mpi_utils.broadcast_flat(t)
        for f in active:
            f.xchg.close()
            f.xchg, f.xchg_reason = None, f"disabled: {why}"
            for name in ("xchg_sp",):
                if getattr(f, name, None) is not None:
                    getattr(f, name).close()
                    setattr(f, name, None)
            if type(f).__name__ == "FusedPolicyUpdate":
                f.split, f.split_reason = f._split_wanted()     # the all-reduce loops run the slab chain
                f._split_space = None
            f._graphs.clear()
            f._args = {}
        self.status_dict["global status"]["peer exchange disabled"] = True

    def _ppo_icm_epoch_overlapped(self, loader, policy_id):
        """
        One PPO epoch and the ICM epoch that follows it (ppo.py:2208-2216) on two HIP streams at once.
        Within an epoch pair the two passes are independent: the PPO pass reads / writes the actor-critic
        bucket, `values` and the value normaliser; the ICM pass reads observations / actions and writes the
        ICM bucket.  Both are latency-bound chains on ~32 workgroups each, so running them side by side
        nearly halves the pair's wall time; shuffles are drawn in the reference's order, results are
        identical.  Both fused updaters only; on N > 1 ranks only when BOTH exchange their gradients through K17 (two
        independent exchange objects, each a kernel launch on its own stream: every rank issues the same sequence per
        object, nothing orders the two against each other) -- host-side collectives stay on one stream, so the RCCL /
        gloo paths run the two epochs one after the other as the reference does.
        """
        pol = self.policies[policy_id]
        if not pol.enable_icm or not getattr(self, "overlap_icm", True) or self.device.type != "cuda" \
                or os.environ.get("PPOAF_OVERLAP_ICM", "1") == "0":            # (measurement switch: the two epochs in turn)
            return False
        fused = self._fused_updater(policy_id, loader.batch_size)
        fused_icm = self._fused_icm_updater(policy_id)
        if fused is None or fused_icm is None:
            return False
        if mpi_utils.distributed_path() and (fused.xchg is None or fused_icm.xchg is None):
            return False
        # each chain's fwd_bwd launches on its own half of the XCDs: weights and panels of one chain stay out of the other's
        # four L2s (C3: +2 % env-steps/s; PPO.xcd_halves = False: both use every XCD)
        halves = getattr(self, "xcd_halves", True)
        fused.xcd_half, fused_icm.xcd_half = (1, 2) if halves else (0, 0)
        fused.begin_epoch(loader.epoch_permutation())
        fused_icm.begin_epoch(loader.epoch_permutation())
        main = torch.cuda.current_stream()
        sa, sb = K.concurrent_stream_pair(self.device)         # two streams on different hardware queues
        sa.wait_stream(main); sb.wait_stream(main)
        with torch.cuda.stream(sa):
            fused.run_epoch()
        with torch.cuda.stream(sb):
            fused_icm.run_epoch()
        main.wait_stream(sa); main.wait_stream(sb)
        loader.prefetch()
        self._publish_epoch_stats(policy_id, fused.end_epoch())
        t = fused_icm.end_epoch()
        self.status_dict[policy_id]["icm loss"] = t[0] / max(t[1], 1.0)
        return True

    def _minibatch_step(self, policy_id, dataset, perm_batch, records, totals):
        """
        One mini-batch of ppo.py:2292-2469 on static buffers.  `records` is the
        (already rank-gathered) moment record of this mini-batch's rewards-to-go.
        """
        pol = self.policies[policy_id]
        windows = dataset.sequence_length > 1
        mb = dataset.gather_sequences(perm_batch) if windows else dataset.gather_minibatch(perm_batch)
        if pol.using_lstm and not windows:
            N = dataset.buffer.num_transitions
            rows = dataset.row_map[perm_batch].long()
            mb = dict(mb, **{k: t.view((N,) + tuple(t.shape[2:]))[rows] for k, t in dataset.buffer.hidden.items()})
        rtg = mb["rewards_to_go"]
        if self.normalize_values:
            vn = self.value_normalizers[policy_id]
            vn.running_stats.integrate_records(records)
            rtg = vn.normalize(rtg, update_stats=False)
        if perm_batch.numel() == 1:          # ppo.py:2305-2306 (after the normaliser update, quirk Q9)
            return
        if pol.using_lstm:
            pol.load_hidden_states(mb)                              # ppo.py:2312-2319
        values, log_probs, entropy = pol.evaluate(mb["critic_obs"], mb["obs"], mb["raw_actions"])
        last = dataset.last_positions(perm_batch)
        dataset.scatter_values(last, values)
        if pol.using_lstm:
            pol.write_back_hidden_states(dataset, last)             # ppo.py:2450-2466
        lp, ent = log_probs.reshape(-1), entropy.reshape(-1)
        sc, dlp, dent, dval = K.ppo_loss_fwd_bwd(
            lp.detach(), mb["log_probs"].reshape(-1), mb["advantages"].reshape(-1), ent.detach(),
            values.detach().reshape(-1), rtg.reshape(-1), self.normalize_adv, pol.surr_clip,
            pol.entropy_weight(), pol.kl_loss_weight, pol.use_huber_loss, 10.0)
        pol.policy_grads.zero_()
        torch.autograd.backward([lp, ent, values.reshape(-1)], [dlp, dent, dval])
        totals[:8] += sc
        totals[8] += 1.0

    def _optimizer_step(self, policy_id):
        self.policies[policy_id].optimizer_step(1.0 / mpi_utils.get_num_procs())

    def _ppo_batch_train(self, data_loader, policy_id):
        """ppo.py:2274-2485: one epoch of shuffled mini-batches; fills status_dict like :2478-2485."""
        pol = self.policies[policy_id]
        ds = data_loader.dataset
        B = data_loader.batch_size
        N = len(ds)
        world = mpi_utils.get_num_procs()
        perm = data_loader.epoch_permutation()
        fused = self._fused_updater(policy_id, B)
        if fused is not None:
            if getattr(fused, "xchg", None) is not None:
                # known-good state to return to should an exchange wait run out of time during the epoch (<= 3 buckets
                # of a few hundred KB: a device-to-device copy per epoch)
                self._epoch_snapshot = [t.clone() for t in self._replicated_state()]
            fused.begin_epoch(perm)
            fused.run_epoch()                      # enqueued asynchronously
            data_loader.prefetch()                 # next shuffle drawn on the host while the GPU works
            self._publish_epoch_stats(policy_id, fused.end_epoch())
            return
        n_full, tail = N // B, N % B
        totals = self._scratch("totals", 9, torch.float64)
        totals.zero_()
        records_all = self._epoch_records(policy_id, ds, perm, B) if self.normalize_values else None
        W3 = 3
        rec_static = self._scratch("rec_static", world * W3, torch.float64)
        perm_static = self._scratch(f"perm_static_{B}", B, torch.int64)

        def run(perm_slice, k, static):
            if records_all is not None:
                rec_static.copy_(records_all[:, k].reshape(-1))
            if static:
                perm_static.copy_(perm_slice)
                self._replay_or_capture(("fb", policy_id, B, pol.entropy_weight(), pol.surr_clip), lambda: self._minibatch_step(
                    policy_id, ds, perm_static, rec_static.view(world, W3), totals))
            else:
                self._minibatch_step(policy_id, ds, perm_slice.contiguous(), rec_static.view(world, W3), totals)
            if perm_slice.numel() == 1:
                return
            mpi_utils.allreduce_sum_(pol.policy_grads)          # identity without a process group
            if static:
                self._replay_or_capture(("opt", policy_id), lambda: self._optimizer_step(policy_id))
            else:
                self._optimizer_step(policy_id)

        # the stateful LSTM modules are run eagerly; everything else replays two hipGraphs per mini-batch.  (Rounds 1-2 also
        # kept the torch-ROCm fallback of agent-grouped (MAT) policies eager because its replayed results drifted by ~1e-4
        # in later iterations.  Root cause, round 3: hipMemsetAsync nodes captured into a hipGraph do not reliably write
        # their value on replay on this stack -- tools/probes/memset_capture_probe.py -- so the clip norm's accumulator,
        # zeroed by a memset inside the optimiser graph, started from junk.  No captured path of the library uses a
        # memset any more: tests/test_gpu_graph_replay.py.)
        graphs = self.use_graphs and not pol.using_lstm
        for k in range(n_full):
            run(perm[k * B:(k + 1) * B], k, graphs)
        if tail:
            run(perm[n_full * B:], n_full, False)

        t = totals.clone()
        mpi_utils.allreduce_sum_(t)
        self._publish_epoch_stats(policy_id, t.cpu().numpy())

    def _icm_batch_train(self, data_loader, policy_id):
        """
        ppo.py:2487-2567: a second shuffled pass over the dataset that trains the ICM:
        icm_loss = (1 - beta) * f_loss + beta * inv_loss, backward, gradient averaging, Adam (no clip).
        """
        pol = self.policies[policy_id]
        ds = data_loader.dataset
        buf = ds.buffer
        B = data_loader.batch_size
        perm = data_loader.epoch_permutation()
        N = perm.numel()
        world = mpi_utils.get_num_procs()
        fused = self._fused_icm_updater(policy_id)
        if fused is not None:
            fused.begin_epoch(perm)
            fused.run_epoch()
            data_loader.prefetch()
            t = fused.end_epoch()
            self.status_dict[policy_id]["icm loss"] = t[0] / max(t[1], 1.0)
            return
        total = self._scratch("icm_total", 1, torch.float64)
        total.zero_()
        counter = 0
        flat = lambda t: t.view((buf.num_transitions,) + tuple(t.shape[2:]))
        perm_static = self._scratch(f"icm_perm_static_{B}", B, torch.int64)
        if pol.agent_shared_icm:             # a device copy refreshed per epoch: the captured launches read it
            agent_order = self._scratch(f"icm_agent_order_{policy_id}", len(pol.agent_idxs), torch.int64)
            agent_order.copy_(torch.as_tensor(np.asarray(pol.agent_idxs), dtype=torch.int64))

        def fwd_bwd(idx):
            n = idx.numel()
            obs = torch.empty((n,) + tuple(buf.observations.shape[2:]), dtype=torch.float32, device=self.device)
            nxt = torch.empty_like(obs)
            act = torch.empty((n,) + tuple(buf.actions.shape[2:]), dtype=buf.actions.dtype, device=self.device)
            K.minibatch_gather([(flat(buf.observations), obs), (flat(buf.next_observations), nxt),
                                (flat(buf.actions), act)], idx, buf.row_map)
            if pol.agent_shared_icm:
                # ppo.py:2520-2538 (case 2): agents re-ordered by the policy's agent_idxs, then side by side in one
                # row.  (agent_idxs is the in-place shuffled index vector of shuffle_agent_ids, as in the reference.)
                obs, nxt, act = (x.reshape(n, buf.A, -1).index_select(1, agent_order).reshape(n, -1) for x in (obs, nxt, act))
            elif pol.agent_grouping:
                # ppo.py:2540-2545 (case 3): every (row, agent) pair is one ICM sample
                rows = n * buf.A
                obs, nxt, act = obs.reshape(rows, -1), nxt.reshape(rows, -1), act.reshape(rows, -1)
            _, inv_loss, f_loss = pol.icm_model(obs, nxt, act)
            icm_loss = (1.0 - pol.icm_beta) * f_loss + pol.icm_beta * inv_loss
            total.add_(icm_loss.detach().double())
            pol.icm_optim.zero_grad()
            icm_loss.backward()

        opt = lambda: pol.icm_optim.step(grad_scale=1.0 / world, max_norm=None)
        for o in range(0, N, B):
            idx = perm[o:o + B]
            if self.use_graphs and idx.numel() == B:
                # same two hipGraphs per mini-batch as the torch PPO path: gather + forward + backward,
                # [gradient all-reduce], Adam
                perm_static.copy_(idx)
                self._replay_or_capture(("icm_fb", policy_id, B), lambda: fwd_bwd(perm_static))
                mpi_utils.allreduce_sum_(pol.icm_model.flat_grads)
                self._replay_or_capture(("icm_opt", policy_id), opt)
            else:
                fwd_bwd(idx.contiguous())
                mpi_utils.allreduce_sum_(pol.icm_model.flat_grads)
                opt()
            counter += 1
        t = torch.cat([total, torch.tensor([float(counter)], dtype=torch.float64, device=self.device)])
        mpi_utils.allreduce_sum_(t)
        t = t.cpu().numpy()
        self.status_dict[policy_id]["icm loss"] = t[0] / max(t[1], 1.0)

    def _fused_icm_updater(self, policy_id):
        if self.update_mode == "torch" or self.device.type != "cuda":
            return None
        key = ("icm", policy_id)
        if key not in self._fused:
            from .fused_update import FusedIcmUpdate
            why = FusedIcmUpdate.unsupported_reason(self.policies[policy_id])
            if why and self.verbose:
                rank_print(f"policy {policy_id}: torch ICM update path ({why})")
            self._fused[key] = None if why else FusedIcmUpdate(self, policy_id)
        return self._fused[key]

    def _fused_updater(self, policy_id, B):
        if self.update_mode == "torch" or self.device.type != "cuda":
            return None
        key = (policy_id, B)
        if key not in self._fused:
            from .fused_update import FusedMatUpdate, FusedPolicyUpdate
            if self.policies[policy_id].agent_grouping:
                FusedPolicyUpdate = FusedMatUpdate                      # K15 instead of K12
            why = FusedPolicyUpdate.unsupported_reason(self.policies[policy_id], B)
            if why:
                if self.update_mode == "fused":
                    raise NotImplementedError(f"update_mode='fused' but {why}")
                if self.verbose:
                    rank_print(f"policy {policy_id}: torch update path ({why})")
                self._fused[key] = None
            else:
                self._fused[key] = FusedPolicyUpdate(self, policy_id)
        return self._fused[key]

    def _publish_epoch_stats(self, policy_id, t):
        """ppo.py:2471-2485: counters and sums across ranks, then per-mini-batch averages."""
        pol = self.policies[policy_id]
        counter = max(t[8], 1.0)
        if t[K.SC_BAD] > 0:
            raise FloatingPointError("ratios are nan or inf (ppo.py:2361-2387)")
        ew = pol.entropy_weight()
        sd = self.status_dict[policy_id]
        sd["weighted entropy"] = (t[K.SC_ENTROPY] * ew / counter) if ew != 0.0 else 0.0
        sd["actor loss"] = t[K.SC_SURR] / counter
        sd["critic loss"] = t[K.SC_CRITIC] / counter
        sd["kl avg"] = t[K.SC_KL] / counter

    # ------------------------------------------------------------------ helpers
    def _scratch(self, name, n, dtype):
        key = ("scratch", name)
        if key not in self._graphs:
            self._graphs[key] = torch.zeros(n, dtype=dtype, device=self.device)
        return self._graphs[key]

    def _epoch_records(self, policy_id, ds, perm, B, field="rewards_to_go", gather=True):
        """
        (n, mean, M2) of the rewards-to-go of EVERY mini-batch of the epoch,
        all-gathered across ranks once: [R, n_batches, 3].  The reference gathers
        the raw data inside every mini-batch (ppo.py:2299-2303 -> stats.py:47-50).
        """
        N = perm.numel()
        A = ds.buffer.A                      # grouped rows carry A values each: a mini-batch holds B*A of them
        rtg = getattr(ds.buffer, field).view(ds.buffer.num_transitions, A)[ds.row_map.long()[ds.last_positions(perm)]].reshape(-1)
        N, B = N * A, B * A
        nb = (N + B - 1) // B
        pad = nb * B - N
        x = rtg.double()
        if pad:
            x = torch.cat([x, torch.zeros(pad, dtype=torch.float64, device=x.device)])
        x = x.view(nb, B)
        cnt = torch.full((nb,), float(B), dtype=torch.float64, device=x.device)
        if pad:
            cnt[-1] = float(B - pad)
        mask = torch.ones(nb, B, dtype=torch.bool, device=x.device)
        if pad:
            mask[-1, B - pad:] = False
        mean = (x * mask).sum(1) / cnt
        m2 = (((x - mean[:, None]) ** 2) * mask).sum(1)
        rec = torch.stack([cnt, mean, m2], dim=1)                     # [nb, 3]
        world = mpi_utils.get_num_procs()
        if not gather or not mpi_utils.distributed_path():
            return rec.unsqueeze(0)
        out = mpi_utils.allgather_records(rec.reshape(-1))
        return out.view(world, nb, 3)

    def _replay_or_capture(self, key, fn):
        """Capture `fn` (kernel launches on static buffers) into a hipGraph once, then replay."""
        g = self._graphs.get(key)
        if g is None:
            # warm-up on a side stream (allocator + lazy init), then capture
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn()
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            self._graphs[key] = g
            # the warm-up + capture passes executed the step once for real (warm-up);
            # capture itself does not execute.  Callers account for that (see below).
            return
        g.replay()
