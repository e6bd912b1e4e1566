"""
MATPolicy -- the Multi-Agent-Transformer policy surface (policies/mat_policy.py:24-1086 of the
reference) on the device rollout buffer.

Same hooks as the reference: agent_grouping = True, one actor_critic network and ONE optimiser
(summed actor + critic loss, one clip over all parameters: mat_policy.py:677-699), the critic
encodes the whole agent sequence once and the decoder samples the agents autoregressively during
rollouts (:441-519), evaluation is teacher-forced with the shifted one-hot action block (:378-439),
use_huber_loss defaults to True (:27-29).

Layout: a dataset row is one env step of one env with its A agents side by side
([A, .] items of PPOSharedEpisodeDataset); tensors are [rows, A, .] on the device.  The attention
core is the f32-MFMA kernel K9; linears / LayerNorm / GELU are torch-ROCm.
"""
import numpy as np
import torch
import torch.nn.functional as t_func

from ..networks.multi_agent_transformer import MATActorCritic
from ..spaces import get_agent_shared_space, get_space_shape
from ..utils import mpi_utils
from ..utils.episode_info import PPODataset, RolloutBuffer
from .ppo_policy import FlatAdam, PPOPolicy


class MATPolicy(PPOPolicy):

    def __init__(self, ac_network=MATActorCritic, mat_kw_args={}, use_huber_loss=True, **kw_args):
        super().__init__(ac_network=ac_network, mat_kw_args=mat_kw_args, use_huber_loss=use_huber_loss, **kw_args)
        self.agent_grouping = True
        self.expanded_actor_space = False
        if get_space_shape(self.actor_obs_space) != get_space_shape(self.critic_obs_space):
            # mat_policy.py:75-79: the actor then sees the critic's (wider) observation
            self.expanded_actor_space = True
            self.have_step_constraints = True
            self.have_reset_constraints = True
            self.actor_obs_space = self.critic_obs_space

    # ------------------------------------------------------------------ setup
    def finalize(self, status_dict, device):
        """mat_policy.py:191-245."""
        if len(self.agent_ids) == 0:
            self.register_agent("agent0")
        self._env_agent_index = {a: i for i, a in enumerate(self.agent_ids)}    # env's agent order
        self.agent_idxs = np.arange(len(self.agent_ids))
        self.num_agents = self.agent_idxs.size
        self.device = torch.device(device)
        self._initialize_networks(**self.network_args)
        for c in (self.lr, self.icm_lr, self.entropy_weight, self.intr_reward_weight):
            c.finalize(status_dict)
        if self.have_bootstrap_clip:
            self.bootstrap_clip[0].finalize(status_dict)
            self.bootstrap_clip[1].finalize(status_dict)
        self.actor_critic_optim = FlatAdam(self.actor_critic, self.lr(), eps=1e-5)
        self.icm_optim = FlatAdam(self.icm_model, self.icm_lr(), eps=1e-5) if self.enable_icm else None   # mat_policy.py:224-227
        self.shuffle_agent_ids()

    def _initialize_networks(self, ac_network, enable_icm, icm_network, mat_kw_args, icm_kw_args, **kw_args):
        """mat_policy.py:88-176."""
        if not issubclass(ac_network, MATActorCritic):
            raise TypeError(f"ac_network for MATPolicy must be a subtype of MATActorCritic, got {ac_network}")
        self.actor_critic = ac_network(name="actor_critic", obs_space=self.critic_obs_space,
                                       action_space=self.action_space, num_agents=len(self.agent_ids),
                                       test_mode=self.test_mode,
                                       seed=self.random_seed + 7919 * mpi_utils.get_rank(), **mat_kw_args)
        self.actor_critic.to(self.device)
        mpi_utils.broadcast_model_parameters(self.actor_critic)
        self.actor = self.actor_critic.actor
        self.critic = self.actor_critic.critic
        self.policy_params = self.actor_critic.flat_params
        self.policy_grads = self.actor_critic.flat_grads
        if enable_icm:
            # mat_policy.py:132-176.  Two forms: one ICM over each agent's own (observation, action) -- rows are
            # then (env, agent) pairs -- or, with agent_shared_icm, one ICM over the concatenation of the
            # group's observations / actions in the ORIGINAL agent order (icm_agent_ids), one reward per env.
            from ..networks.icm import ICM
            self.icm_agent_ids = None
            icm_obs_space, icm_action_space = self.actor_obs_space, self.action_space
            if self.agent_shared_icm:
                if self.expanded_actor_space:
                    raise ValueError("agent_shared_icm can only be enabled with the multi-agent transformer "
                                     "if critic view is set to local (mat_policy.py:146-152)")
                n = len(self.agent_ids)
                icm_obs_space = get_agent_shared_space(self.actor_obs_space, n)
                icm_action_space = get_agent_shared_space(self.action_space, n)
                self.icm_agent_ids = np.array(self.agent_ids).copy()
            icm_cls = ICM if icm_network is None else icm_network
            self.icm_model = icm_cls(name="icm", obs_space=icm_obs_space, action_space=icm_action_space,
                                     test_mode=self.test_mode, **icm_kw_args)
            self.icm_model.to(self.device)
            mpi_utils.broadcast_model_parameters(self.icm_model)

    def agent_slot_order(self):
        """Index of the env's agent that sits in each slot of the grouped rows (after the MAT shuffles)."""
        return np.array([self._env_agent_index[a] for a in self.agent_ids], dtype=np.int64)

    def to(self, device):
        self.device = torch.device(device)
        self.actor_critic.to(self.device)

    def eval(self):
        self.actor_critic.eval()

    def train(self):
        self.actor_critic.train()

    # ---------------------------------------------------------------- rollout
    def initialize_dataset(self):
        """mat_policy.py:179-189 (PPOSharedEpisodeDataset: rows of [A, .])."""
        self.dataset = PPODataset(device=self.device, action_dtype=self.action_dtype, sequence_length=1)
        self.dataset.shared = True
        # quirk Q14 (replicated; pinned by fixture g12_c5_mat): the reference creates the dataset BEFORE the rollout
        # reshuffles the agents (ppo.py:1546-1547 vs 1643-1644) and PPOSharedEpisodeDataset keeps the agent_ids array it
        # was handed (episode_info.py:1012), so the dataset's agent axis stays in the order the policy had at this point
        self._dataset_slot_order = self.agent_slot_order()

    def finalize_dataset(self):
        """ppo_policy.py:714-719, with the dataset's agent axis put into its creation-time order (quirk Q14)."""
        rollout_order = self.agent_slot_order()
        want = getattr(self, "_dataset_slot_order", rollout_order)
        if not np.array_equal(want, rollout_order):
            k = torch.as_tensor(np.argsort(rollout_order)[want], device=self.device)      # dataset slot j <- rollout slot k[j]
            b = self.buffer
            for name in ("observations", "critic_observations", "next_observations", "actions", "raw_actions", "values",
                         "log_probs", "rewards", "boot_value", "boot_reward"):
                x = getattr(b, name)
                if x is not None:
                    x.copy_(x.index_select(2, k))
            if getattr(b, "boot_stats", None) is not None:
                b.boot_stats = b.boot_stats.index_select(2, k)
        self.dataset.build()

    def initialize_episodes(self, env_batch_size, status_dict, ts_per_rollout=None):
        if ts_per_rollout is None:
            raise ValueError("initialize_episodes needs ts_per_rollout (total steps over all envs)")
        T = int(ts_per_rollout) // int(env_batch_size)
        self.env_batch_size = int(env_batch_size)
        A = len(self.agent_ids)
        if A < 2:
            raise NotImplementedError("MATPolicy groups the agents of an env into one sequence: it needs at least two "
                                      "agents (a single agent is an ordinary PPOPolicy)")
        obs_dim = int(np.prod(get_space_shape(self.actor_obs_space)))
        cobs_dim = int(np.prod(get_space_shape(self.critic_obs_space)))
        if self.buffer is None or (self.buffer.T, self.buffer.C) != (T, self.env_batch_size):
            self.buffer = RolloutBuffer(T, self.env_batch_size, obs_dim, cobs_dim, self.action_dim,
                                        self.action_dtype, self.device, keep_next_observations=self.enable_icm,
                                        agents_per_row=A)
        else:
            self.buffer.end_kind.zero_()
            self.buffer.fixed_length = True
            self.buffer.steps_written = 0
        self._t = 0
        self.dataset.attach(self.buffer, self.gamma, self.lambd, self.get_bs_clip_range(None), self.use_gae)

    def fused_step_unsupported_reason(self):
        """'' when the K16 rollout-step kernel covers this policy (same coverage as the fused update K15)."""
        from ..fused_update import _describe_mat
        return _describe_mat(self)[1]

    def rollout_step(self, t, critic_obs, actor_obs, value_normalizer=None, forced_raw_action=None):
        """
        forced_raw_action (optional int64 device tensor [E, A] / [E, A, 1]): log these actions instead of sampling
        (replay of a recorded rollout).
        One env step of get_rollout_actions (encoder + A autoregressive decoder passes + sampling) and
        get_critic_values (+ denormalisation) as ONE launch (K16) that stores straight into row t of the
        rollout buffer.  critic_obs / actor_obs: grouped device tensors [E, A, .].  Returns the action row
        (a view of the buffer, [E, A, 1]) for env.step; rewards are added by `finish_step`.
        """
        from .. import _lib
        from .. import kernels as K
        from ..fused_update import _describe_mat
        import ctypes as C
        buf = self.buffer
        E, A = buf.C, buf.A
        a = getattr(self, "_step_args", None)
        if a is None:
            topo, why = _describe_mat(self)
            if topo is None:
                raise _lib.PpoafError(f"mat rollout_step: {why}")
            a = _lib.MatStepArgs()
            a.obs_dim, a.num_agents, a.num_actions, a.embedding = topo["obs_dim"], topo["num_agents"], topo["num_actions"], 64
            for i, o in enumerate(topo["offsets"]):
                a.offsets[i] = o
            a.params = self.actor_critic.flat_params.data_ptr()
            a.E = E
            a.actor_obs_dim = int(buf.observations.shape[-1])
            self._step_args = a
        if critic_obs.shape != (E, A, a.obs_dim) or not critic_obs.is_cuda:
            raise _lib.PpoafError(f"mat rollout_step: critic_obs must be a device tensor [{E}, {A}, {a.obs_dim}]")
        critic_obs = critic_obs.contiguous()
        actor_obs = actor_obs.contiguous()
        a.critic_obs, a.actor_obs = critic_obs.data_ptr(), actor_obs.data_ptr()
        a.seed, a.offset = self.actor.distribution.rng.take(E * A)
        a.forced_action = None
        if forced_raw_action is not None:
            if not (forced_raw_action.is_cuda and forced_raw_action.dtype == torch.int64 and forced_raw_action.is_contiguous()
                    and forced_raw_action.numel() == E * A):
                raise _lib.PpoafError(f"mat rollout_step: forced_raw_action must be a contiguous int64 device tensor [{E}, {A}]")
            a.forced_action = forced_raw_action.data_ptr()
        a.normalize_values = int(value_normalizer is not None)
        if value_normalizer is not None:
            a.vn_mean = value_normalizer.running_stats.mean_t.data_ptr()
            a.vn_var = value_normalizer.running_stats.var_t.data_ptr()
        a.action_out, a.raw_action_out = buf.actions[t].data_ptr(), buf.raw_actions[t].data_ptr()
        a.logp_out, a.value_out = buf.log_probs[t].data_ptr(), buf.values[t].data_ptr()
        a.critic_obs_copy_out, a.obs_copy_out = buf.critic_observations[t].data_ptr(), buf.observations[t].data_ptr()
        _lib.check(_lib.load().ppoaf_mat_policy_step(C.byref(a), K.stream()), "mat_policy_step")
        return buf.actions[t]

    def finish_step(self, t, rewards, next_obs=None):
        buf = self.buffer
        buf.rewards[t].copy_(rewards.reshape(buf.rewards[t].shape))
        if buf.next_observations is not None and next_obs is not None:
            buf.next_observations[t].copy_(next_obs.reshape(buf.next_observations[t].shape))
        buf.steps_written = max(buf.steps_written, t + 1)
        self._t = t + 1

    def _get_tokened_action_block(self, batch_size):
        """mat_policy.py:308-344."""
        A = len(self.agent_ids)
        if self.action_dtype == "continuous":
            return torch.zeros((batch_size, A, self.action_pred_size), device=self.device)
        blk = torch.zeros((batch_size, A, self.action_pred_size + 1), device=self.device)
        blk[:, 0, 0] = 1
        return blk

    def _get_autoregressive_actions(self, encoded_obs, forced_raw_action=None):
        """mat_policy.py:441-519: A decoder passes, agent i conditioned on the actions of agents < i.
        forced_raw_action [B, A, .]: recorded raw actions to log instead of sampling (replay)."""
        B, A = encoded_obs.shape[0], len(self.agent_ids)
        block = self._get_tokened_action_block(B)
        off = 1 if self.action_dtype == "discrete" else 0
        adt = torch.int64 if self.action_dtype == "discrete" else torch.float32
        out_a = torch.zeros((B, A, self.action_dim), dtype=adt, device=self.device)
        out_raw = torch.zeros_like(out_a)
        out_lp = torch.zeros((B, A, 1), dtype=torch.float32, device=self.device)
        with torch.no_grad():
            for i in range(A):
                pred = self.actor(block, encoded_obs)[:, i, :].contiguous()
                if forced_raw_action is None:
                    action, raw_action, log_prob = self.actor.distribution.sample_distribution(pred)
                else:
                    raw_action = forced_raw_action.reshape(B, A, -1)[:, i, :].contiguous()
                    log_prob, _ = self.actor.distribution.get_log_probs_and_entropy(pred, raw_action)
                    action = raw_action if self.action_dtype == "discrete" else self.actor.distribution.refine_prediction(raw_action)
                out_a[:, i, :] = action.reshape(B, self.action_dim)
                out_raw[:, i, :] = raw_action.reshape(B, self.action_dim)
                out_lp[:, i, :] = log_prob.reshape(B, 1)
                if i + 1 < A:
                    if self.action_dtype == "discrete":
                        block[:, i + 1, off:] = t_func.one_hot(action.reshape(B), num_classes=self.action_pred_size).float()
                    else:
                        block[:, i + 1, off:] = action.reshape(B, -1)
        return out_a, out_raw, out_lp

    def get_rollout_actions(self, obs, forced_raw_action=None):
        """
        mat_policy.py:587-626.  Device tensors arrive grouped, [E, A, O], and grouped tensors are
        returned ([E, A, .]); numpy arrives in the reference's [A, E, O] and is swapped like there.
        """
        as_numpy = not torch.is_tensor(obs)
        t_obs = self._to_device(np.swapaxes(obs, 0, 1) if as_numpy else obs)
        with torch.no_grad():
            encoded_obs, _ = self.critic(t_obs)
        actions, raw_actions, log_probs = self._get_autoregressive_actions(encoded_obs, forced_raw_action)
        if as_numpy:
            return (torch.swapaxes(raw_actions, 0, 1).cpu().numpy(), torch.swapaxes(actions, 0, 1).cpu().numpy(),
                    torch.swapaxes(log_probs, 0, 1).detach())
        return raw_actions, actions, log_probs

    def get_critic_values(self, obs):
        """mat_policy.py:660-675."""
        _, values = self.critic(obs)
        return values

    # ----------------------------------------------------------------- update
    def evaluate(self, batch_critic_obs, batch_obs, batch_actions):
        """mat_policy.py:378-439,628-658 -> values [B,A,1], log_probs [B,A,1], entropy [B,A,1]."""
        B, A = batch_critic_obs.shape[0], len(self.agent_ids)
        block = self._get_tokened_action_block(B)
        if self.action_dtype == "discrete":
            acts = batch_actions.reshape(B, A)
            block[:, 1:, 1:] = t_func.one_hot(acts, num_classes=self.action_pred_size)[:, :-1, :].float()
        else:
            block[:, 1:, :] = batch_actions[:, :-1, :]
        values, pred = self.actor_critic(batch_critic_obs, block)
        pred = pred.reshape(-1, self.action_pred_size)
        flat_actions = batch_actions.reshape(-1, self.action_dim)
        log_probs, entropy = self.actor.distribution.get_log_probs_and_entropy(pred, flat_actions)
        return values, log_probs.reshape(B, A, -1), entropy.reshape(B, A, -1)

    def optimizer_step(self, grad_scale):
        self.actor_critic_optim.step(grad_scale=grad_scale, max_norm=self.gradient_clip)

    def update_weights(self, actor_loss, critic_loss):
        """mat_policy.py:677-699."""
        if self.frozen:
            return
        self.policy_grads.zero_()
        (actor_loss + critic_loss).backward()
        mpi_utils.allreduce_sum_(self.policy_grads)
        self.optimizer_step(1.0 / mpi_utils.get_num_procs())

    def update_learning_rate(self):
        if not self.frozen:
            self.actor_critic_optim.set_lr(self.lr())
            if self.enable_icm:                          # mat_policy.py:895-896
                self.icm_optim.set_lr(self.icm_lr())

    def save(self, save_path, tag="latest"):
        import os
        path = os.path.join(save_path, f"{self.name}-policy", str(tag))
        os.makedirs(path, exist_ok=True)
        self.actor_critic.save(path)                     # actor_critic_<rank>.model, as the reference
        torch.save(self.actor_critic_optim.state_dict(),
                   os.path.join(path, f"actor_critic_optim_{mpi_utils.get_rank()}"))
        if self.enable_icm:                              # mat_policy.py:1105-1111
            self.icm_model.save(path)
            torch.save(self.icm_optim.state_dict(), os.path.join(path, f"icm_optim_{mpi_utils.get_rank()}"))

    def load(self, load_path, tag="latest"):
        import os
        path = os.path.join(load_path, f"{self.name}-policy", str(tag))
        self.actor_critic.load(path)
        f = os.path.join(path, f"actor_critic_optim_{mpi_utils.get_rank()}")
        if not os.path.exists(f):
            f = os.path.join(path, "actor_critic_optim_0")
        self.actor_critic_optim.load_state_dict(torch.load(f, map_location="cpu", weights_only=False))
        if self.enable_icm:
            self.icm_model.load(path)
            f = os.path.join(path, f"icm_optim_{mpi_utils.get_rank()}")
            if not os.path.exists(f):
                f = os.path.join(path, "icm_optim_0")
            if os.path.exists(f):
                self.icm_optim.load_state_dict(torch.load(f, map_location="cpu", weights_only=False))

    def direct_load(self, policy_load_path):
        """mat_policy.py (direct_load): the networks only, from the directory itself."""
        self.actor_critic.load(policy_load_path)
        if self.enable_icm:
            self.icm_model.load(policy_load_path)

    def get_agent_shared_intrinsic_rewards(self, prev_obs, obs, actions):
        """
        mat_policy.py:1012-1090.  The reference takes dicts keyed by agent id and stacks them in
        icm_agent_ids order; here the environment's agent-major batches [A*E, .] (rows of agent a at
        a*E..(a+1)*E, the env's = the original agent order) are regrouped to [E, A*.]: one ICM row and one
        reward per env, already weighted.  Returns float32 [E].
        """
        A = len(self.agent_ids)
        obs_1 = self._to_device(prev_obs)
        if obs_1.dim() < 2:
            raise ValueError(f"get_agent_shared_intrinsic_rewards expects a batch of observations, got {obs_1.shape}")
        E = obs_1.shape[0] // A
        join = lambda x: x.reshape(A, E, -1).transpose(0, 1).reshape(E, -1)
        obs_1, obs_2 = join(obs_1), join(self._to_device(obs))
        adt = torch.int64 if self.action_dtype in ("discrete", "multi-discrete") else torch.float32
        act = join(self._to_device(actions, adt))
        with torch.no_grad():
            intr, _, _ = self.icm_model(obs_1, obs_2, act)
        return intr.reshape(-1) * float(self.intr_reward_weight())

    def apply_step_constraints(self, obs, critic_obs, reward, terminated, truncated, info):
        """
        mat_policy.py:808-853: with a non-local critic view the actor must see the critic's observation
        (`obs[agent] = critic_obs[agent]`).  This build makes that substitution where the observation is
        consumed -- the rollout hands the critic's observation to K16 / get_rollout_actions for the actor
        when `expanded_actor_space` is set (ppo.py rollout) -- so the env tuple passes through unchanged.
        """
        return obs, critic_obs, reward, terminated, truncated, info

    def apply_reset_constraints(self, obs, critic_obs):
        """mat_policy.py:855-878: see apply_step_constraints."""
        return obs, critic_obs
