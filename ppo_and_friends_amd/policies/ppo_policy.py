"""
PPOPolicy -- the policy surface PPO drives (policies/ppo_policy.py:26-1419 of the
reference), rebuilt on the device-resident rollout buffer and the HIP kernels.

Same constructor keywords, attributes and method names as the reference for the
hot path: register_agent, finalize, initialize_dataset,
initialize_episodes, get_rollout_actions, get_critic_values, add_episode_info,
end_episodes, finalize_dataset, clear_dataset, evaluate, update_weights,
update_learning_rate, get_bs_clip_range, save / load.

What changed underneath
  * episodes are flags in a `[T, A*E]` SoA buffer, not E Python objects
    (utils/episode_info.py here); observations / actions may arrive as numpy
    arrays (reference call shape) or as device tensors (no host round trip);
  * the distribution is evaluated on the device (the reference moves the actor
    output to the CPU: ppo_policy.py:770,930);
  * update_weights = backward -> one RCCL all-reduce of the flat gradient bucket
    -> fused clip + Adam kernels (reference: per-tensor pickled MPI allreduce,
    clip_grad_norm_, Adam.step: ppo_policy.py:1032-1055).
"""
import os

import numpy as np
import torch

from .. import kernels as K
from ..networks.distributions import get_actor_distribution
from ..networks.feed_forward import FeedForwardNetwork
from ..spaces import (get_action_prediction_shape, get_flattened_space_length,
                      get_space_dtype_str, get_space_shape)
from ..utils import mpi_utils
from ..utils.episode_info import PPODataset, RolloutBuffer
from ..utils.mpi_utils import rank_print
from ..utils.schedulers import CallableValue


def _callable(v):
    return v if callable(v) else CallableValue(v)


class FlatAdam:
    """
    Adam(eps=1e-5) + clip_grad_norm_ over a flat bucket (ppo_policy.py:336-343,
    1037-1042) as two HIP launches; step counter and lr live on the device.
    """

    def __init__(self, network, lr, eps=1e-5, betas=(0.9, 0.999), storage=None):
        """storage = (exp_avg, exp_avg_sq, step_count[1], lr[1], norm_scratch or None) views of a policy-wide
        allocation (so the fused update kernels see actor and critic state adjacent).  norm_scratch is this
        optimiser's own float64[2 + partials] (K11's / K15's per-workgroup squared-norm partials)."""
        self.network = network
        dev = network.flat_params.device
        if storage is None:
            storage = (torch.zeros_like(network.flat_params), torch.zeros_like(network.flat_params),
                       torch.zeros(1, dtype=torch.int64, device=dev),
                       torch.full((1,), float(lr), dtype=torch.float32, device=dev), None)
        self.exp_avg, self.exp_avg_sq, self.step_count, self.lr, self.norm_scratch = storage
        if self.norm_scratch is None:
            self.norm_scratch = torch.zeros(max(K.NORM_SCRATCH_DOUBLES, 2 + (network.flat_params.numel() + 1023) // 1024),
                                            dtype=torch.float64, device=dev)
        self.lr.fill_(float(lr))
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.eps = eps
        self.betas = betas
        self.param_groups = [{"lr": float(lr)}]      # torch.optim surface for update_optimizer_lr

    def zero_grad(self):
        self.network.flat_grads.zero_()

    def step(self, grad_scale=1.0, max_norm=None):
        K.clip_adam_step(self.network.flat_params, self.network.flat_grads, self.exp_avg,
                         self.exp_avg_sq, self.step_count, self.lr, self.norm_scratch,
                         beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                         grad_scale=grad_scale, max_norm=max_norm, grad_norm_out=self.grad_norm)

    def set_lr(self, lr):
        self.param_groups[0]["lr"] = float(lr)
        self.lr.fill_(float(lr))

    def state_dict(self):
        """torch.optim.Adam.state_dict() layout (what the reference torch.saves, ppo_policy.py:1239-1247)."""
        from ..utils.reference_io import adam_state_dict
        return adam_state_dict(self.network, self.exp_avg, self.exp_avg_sq, int(self.step_count.item()),
                               self.param_groups[0]["lr"], self.betas, self.eps)

    def load_state_dict(self, sd):
        if "param_groups" in sd:
            from ..utils.reference_io import load_adam_state_dict
            step, lr = load_adam_state_dict(sd, self.network, self.exp_avg, self.exp_avg_sq)
            self.step_count.fill_(step); self.set_lr(lr)
            return
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])      # this package's old layout
        self.step_count.fill_(int(sd["step"])); self.set_lr(sd["lr"])


class PPOPolicy:

    def __init__(self, name, action_space, actor_observation_space, critic_observation_space,
                 envs_per_proc, bootstrap_clip=(-100., 100.), ac_network=FeedForwardNetwork,
                 actor_kw_args={}, critic_kw_args={}, icm_kw_args={}, target_kl=100.,
                 surr_clip=0.2, vf_clip=None, gradient_clip=0.5, lr=3e-4, icm_lr=3e-4,
                 entropy_weight=0.01, kl_loss_weight=0.0, use_gae=True, gamma=0.99, lambd=0.95,
                 dynamic_bs_clip=False, enable_icm=False, agent_shared_icm=False, icm_network=None,
                 intr_reward_weight=1.0, icm_beta=0.8, use_huber_loss=False, test_mode=False,
                 verbose=False, random_seed=0, **kw_args):
        # ppo_policy.py:164-212
        self.name = name
        self.action_space = action_space
        self.actor_obs_space = actor_observation_space
        self.critic_obs_space = critic_observation_space
        self.enable_icm = enable_icm
        self.agent_shared_icm = agent_shared_icm
        self.test_mode = test_mode
        self.use_gae = use_gae
        self.gamma = gamma
        self.lambd = lambd
        self.dynamic_bs_clip = dynamic_bs_clip
        self.using_lstm = False
        self.dataset = None
        self.buffer = None
        self.device = torch.device("cpu")
        self.agent_ids = np.array([])
        self.icm_beta = icm_beta
        self.target_kl = target_kl
        self.surr_clip = surr_clip
        self.vf_clip = vf_clip
        self.gradient_clip = gradient_clip
        self.kl_loss_weight = kl_loss_weight
        self.envs_per_proc = envs_per_proc
        self.agent_grouping = False
        self.have_step_constraints = False
        self.have_reset_constraints = False
        self.verbose = verbose
        self.use_huber_loss = use_huber_loss
        self.frozen = False
        self.random_seed = random_seed
        self.lr = _callable(lr)
        self.icm_lr = _callable(icm_lr)
        self.entropy_weight = _callable(entropy_weight)
        self.intr_reward_weight = _callable(intr_reward_weight)
        if vf_clip is not None:
            # the reference's vf_clip branch dereferences self.user_huber_loss (ppo.py:2432): unusable there too
            raise NotImplementedError("vf_clip raises AttributeError in the reference (ppo.py:2432); not reproduced")
        self.dynamic_bs_clip = bool(dynamic_bs_clip)

        self.action_dtype = get_space_dtype_str(self.action_space)
        if self.action_dtype not in ("discrete", "continuous", "multi-discrete", "multi-binary"):
            raise NotImplementedError(f"{name}: action dtype {self.action_dtype} is outside the hot-path scope")
        self.have_bootstrap_clip = bootstrap_clip is not None
        self.bootstrap_clip = (None if bootstrap_clip is None
                               else (_callable(bootstrap_clip[0]), _callable(bootstrap_clip[1])))
        self.action_dim = get_flattened_space_length(self.action_space)
        self.action_pred_size = get_action_prediction_shape(self.action_space)[0]
        self.network_args = dict(ac_network=ac_network, enable_icm=enable_icm, icm_network=icm_network,
                                 actor_kw_args=actor_kw_args, critic_kw_args=critic_kw_args,
                                 icm_kw_args=icm_kw_args)
        self.network_args.update(kw_args)
        self._t = 0

    # ------------------------------------------------------------------ setup
    def register_agent(self, agent_id):
        """ppo_policy.py:355-365 (insertion order kept instead of set order: deterministic columns)."""
        if agent_id not in list(self.agent_ids):
            self.agent_ids = np.array(list(self.agent_ids) + [agent_id])

    def finalize(self, status_dict, device):
        """ppo_policy.py:302-345."""
        if len(self.agent_ids) == 0:
            self.register_agent("agent0")
        self.agent_idxs = np.arange(len(self.agent_ids))
        self.num_agents = self.agent_idxs.size
        self._agent_col = {a: i for i, a in enumerate(self.agent_ids)}
        self.device = torch.device(device)
        self._initialize_networks(**self.network_args)
        for c in (self.lr, self.icm_lr, self.entropy_weight, self.intr_reward_weight):
            c.finalize(status_dict)
        if self.have_bootstrap_clip:
            self.bootstrap_clip[0].finalize(status_dict)
            self.bootstrap_clip[1].finalize(status_dict)
        na = self.actor.bucket_size()
        dev = self.device
        self.policy_exp_avg = torch.zeros_like(self.policy_params)
        self.policy_exp_avg_sq = torch.zeros_like(self.policy_params)
        self.policy_step_counts = torch.zeros(2, dtype=torch.int64, device=dev)
        self.policy_lr = torch.full((1,), float(self.lr()), dtype=torch.float32, device=dev)
        # K12: squared norms (2) + Adam bias corrections (4) + per-workgroup squared-norm partials of the bucket
        self.policy_norm_scratch = torch.zeros(6 + 2 * ((self.policy_params.numel() + 1023) // 1024), dtype=torch.float64, device=dev)
        self.actor_optim = FlatAdam(self.actor, self.lr(), eps=1e-5, storage=(
            self.policy_exp_avg[:na], self.policy_exp_avg_sq[:na], self.policy_step_counts[0:1], self.policy_lr, None))
        self.critic_optim = FlatAdam(self.critic, self.lr(), eps=1e-5, storage=(
            self.policy_exp_avg[na:], self.policy_exp_avg_sq[na:], self.policy_step_counts[1:2], self.policy_lr, None))
        self.icm_optim = FlatAdam(self.icm_model, self.icm_lr(), eps=1e-5) if self.enable_icm else None

    def _initialize_networks(self, ac_network, enable_icm, icm_network, actor_kw_args,
                             critic_kw_args, icm_kw_args, **kw_args):
        """ppo_policy.py:390-472: actor out gain 0.01, critic out gain 1.0; rank-0 broadcast."""
        from ..networks.lstm import PPOLSTMNetwork
        self.using_lstm = isinstance(ac_network, type) and issubclass(ac_network, PPOLSTMNetwork)   # ppo_policy.py:420-422
        self.actor = ac_network(name="actor", in_shape=get_space_shape(self.actor_obs_space),
                                out_shape=get_action_prediction_shape(self.action_space),
                                out_init=0.01, test_mode=self.test_mode, **actor_kw_args)
        self.critic = ac_network(name="critic", in_shape=get_space_shape(self.critic_obs_space),
                                 out_shape=(1,), out_init=1.0, test_mode=self.test_mode,
                                 **critic_kw_args)
        self.actor.distribution = get_actor_distribution(
            self.action_space, seed=self.random_seed + 7919 * mpi_utils.get_rank(), **actor_kw_args)
        self._place_networks()
        mpi_utils.broadcast_flat(self.policy_params)      # one message for actor + critic
        if enable_icm:                                    # ppo_policy.py:461-472
            if self.agent_shared_icm:
                raise NotImplementedError("agent_shared_icm needs agent grouping (MAT); not built")
            from ..networks.icm import ICM
            icm_cls = ICM if icm_network is None else icm_network
            self.icm_model = icm_cls(name="icm", obs_space=self.actor_obs_space,
                                     action_space=self.action_space, test_mode=self.test_mode,
                                     **icm_kw_args)
            self.icm_model.to(self.device)
            mpi_utils.broadcast_model_parameters(self.icm_model)

    def seed(self, seed):
        self.random_seed = seed
        self.actor.distribution.rng.seed = int(seed) + 7919 * mpi_utils.get_rank()

    def _place_networks(self):
        """Actor and critic buckets adjacent in one allocation: one broadcast, one all-reduce."""
        na, nc = self.actor.bucket_size(), self.critic.bucket_size()
        self.policy_params = torch.zeros(na + nc, dtype=torch.float32, device=self.device)
        self.policy_grads = torch.zeros(na + nc, dtype=torch.float32, device=self.device)
        self.actor.flatten_parameters_(self.device, (self.policy_params[:na], self.policy_grads[:na]))
        self.critic.flatten_parameters_(self.device, (self.policy_params[na:], self.policy_grads[na:]))

    def to(self, device):
        self.device = torch.device(device)
        self._place_networks()

    def eval(self):
        self.actor.eval(); self.critic.eval()

    def train(self):
        self.actor.train(); self.critic.train()

    def freeze(self):
        self.frozen = True

    def unfreeze(self):
        self.frozen = False

    def shuffle_agent_ids(self):
        np.random.shuffle(self.agent_idxs)
        self.agent_ids = self.agent_ids[self.agent_idxs]

    # ---------------------------------------------------------------- rollout
    def initialize_dataset(self):
        """ppo_policy.py:506-526."""
        sequence_length = 1
        if self.using_lstm:
            self.actor.reset_hidden_state(batch_size=1, device=self.device)
            self.critic.reset_hidden_state(batch_size=1, device=self.device)
            sequence_length = self.actor.sequence_length
        self.dataset = PPODataset(device=self.device, action_dtype=self.action_dtype,
                                  sequence_length=sequence_length)

    def initialize_episodes(self, env_batch_size, status_dict, ts_per_rollout=None):
        """
        ppo_policy.py:474-504.  The reference allocates E EpisodeInfo objects per
        agent; here one `[T, A*E]` buffer.  T = timesteps per env in this rollout
        (ts_per_rollout / envs_per_proc, ppo.py:317-318,1646-1653).
        """
        if ts_per_rollout is None:
            ts_per_rollout = status_dict["global status"]["ts per rollout"] if status_dict else None
        if ts_per_rollout is None:
            raise ValueError("initialize_episodes needs ts_per_rollout (total steps over all envs)")
        T = int(ts_per_rollout) // int(env_batch_size)
        self.env_batch_size = int(env_batch_size)
        C = self.env_batch_size * len(self.agent_ids)
        obs_dim = int(np.prod(get_space_shape(self.actor_obs_space)))
        cobs_dim = int(np.prod(get_space_shape(self.critic_obs_space)))
        if self.buffer is None or (self.buffer.T, self.buffer.C) != (T, C):
            lstm_spec = None
            if self.using_lstm:
                lstm_spec = ((self.actor.num_lstm_layers, self.actor.lstm_hidden_size),
                             (self.critic.num_lstm_layers, self.critic.lstm_hidden_size))
            self.buffer = RolloutBuffer(T, C, obs_dim, cobs_dim, self.action_dim, self.action_dtype,
                                        self.device, keep_next_observations=self.enable_icm, lstm_spec=lstm_spec)
        else:
            self.buffer.end_kind.zero_()
            self.buffer.fixed_length = True
            self.buffer.steps_written = 0
        self._t = 0
        self._agents_written = 0
        self.dataset.attach(self.buffer, self.gamma, self.lambd, self.get_bs_clip_range(None), self.use_gae)

    def _to_device(self, x, dtype=torch.float32):
        if torch.is_tensor(x):
            return x.to(device=self.device, dtype=dtype)
        return torch.as_tensor(np.asarray(x), dtype=dtype).to(self.device)

    def get_rollout_actions(self, obs, forced_raw_action=None):
        """
        ppo_policy.py:729-794 -> (raw_action, action, log_prob).  numpy in -> numpy
        actions out (reference contract); device tensor in -> device tensors out.
        forced_raw_action: recorded raw actions to log instead of sampling (replay of a rollout).
        """
        if len(obs.shape) < 2:
            raise ValueError(f"get_rollout_actions expects a batch of observations, got shape {obs.shape}")
        as_numpy = not torch.is_tensor(obs)
        t_obs = self._to_device(obs)
        with torch.no_grad():
            pred = self.actor.forward_logits(t_obs)
            if forced_raw_action is None:
                action, raw_action, log_prob = self.actor.distribution.sample_distribution(pred)
            else:
                dist = self.actor.distribution
                raw_action = self._to_device(forced_raw_action, torch.float32 if self.action_dtype == "continuous" else torch.int64)
                raw_action = raw_action.reshape(pred.shape[0], -1)
                log_prob, _ = dist.get_log_probs_and_entropy(pred, raw_action)
                action = dist.refine_prediction(raw_action) if self.action_dtype == "continuous" else raw_action
        if as_numpy:
            return raw_action.cpu().numpy(), action.cpu().numpy(), log_prob.detach()
        return raw_action, action, log_prob

    # ---- LSTM hidden states (ppo_policy.py:593-627, ppo.py:2312-2319,2450-2466)
    def store_hidden_states(self, t, terminated):
        """
        Row t of the buffer receives the networks' (hidden, cell) AFTER this step's forward passes,
        zeroed for the envs that terminated at this step.  The networks' own state is NOT reset
        (the reference never resets it inside a rollout).
        """
        keep = (~terminated).to(torch.float32).view(-1, 1, 1)
        h = self.buffer.hidden
        for name, net in (("actor", self.actor), ("critic", self.critic)):
            hid, cell = net.hidden_state                       # [layers, C, H]
            h[name + "_hidden"][t].copy_(hid.transpose(0, 1) * keep)
            h[name + "_cell"][t].copy_(cell.transpose(0, 1) * keep)

    def load_hidden_states(self, mb):
        """ppo.py:2312-2319: the mini-batch's stored states become the networks' initial states."""
        self.actor.hidden_state = (mb["actor_hidden"].transpose(0, 1).contiguous(),
                                   mb["actor_cell"].transpose(0, 1).contiguous())
        self.critic.hidden_state = (mb["critic_hidden"].transpose(0, 1).contiguous(),
                                    mb["critic_cell"].transpose(0, 1).contiguous())

    def write_back_hidden_states(self, dataset, batch_idxs):
        """ppo.py:2450-2466: the states the networks ended the window with replace the stored ones."""
        dataset.actor_hidden[batch_idxs] = self.actor.hidden_state[0].detach().transpose(0, 1)
        dataset.critic_hidden[batch_idxs] = self.critic.hidden_state[0].detach().transpose(0, 1)
        dataset.actor_cell[batch_idxs] = self.actor.hidden_state[1].detach().transpose(0, 1)
        dataset.critic_cell[batch_idxs] = self.critic.hidden_state[1].detach().transpose(0, 1)

    def fused_step_unsupported_reason(self):
        """'' when the K6+K7 rollout-step kernel covers this policy (same coverage as the fused update)."""
        from ..fused_update import FusedPolicyUpdate
        return FusedPolicyUpdate.unsupported_reason(self, 2)

    def rollout_step(self, t, obs, critic_obs, value_normalizer=None, forced_raw_action=None):
        """
        forced_raw_action (optional, [E] / [E,1] int64 or [E,D] float32 device tensor): log these raw actions
        instead of sampling -- replay of a recorded rollout (log-probs, refined actions and values are computed
        as usual).
        One env step of get_rollout_actions + get_critic_values (+ denormalisation) +
        add_episode_info's action/value/log-prob/observation writes, as ONE launch that
        stores straight into row t of the rollout buffer.  Returns the action row (a
        view of the buffer) for env.step; rewards / next observations are added by
        `finish_step` once the environment has answered.
        """
        from .. import _lib
        from ..fused_update import _describe
        from ..networks.distributions import GaussianDistribution
        buf = self.buffer
        E = buf.C                       # agents x envs rows, agent-major
        a = getattr(self, "_step_args", None)
        if a is None:
            a = _lib.PolicyStepArgs()
            gauss = isinstance(self.actor.distribution, GaussianDistribution)
            a.actor, _ = _describe(self.actor, self.policy_params, gauss)
            a.critic, _ = _describe(self.critic, self.policy_params, False)
            a.params = self.policy_params.data_ptr()
            a.E = E
            a.head_kind = K.HEAD_GAUSSIAN if gauss else K.HEAD_CATEGORICAL
            a.min_std = float(getattr(self.actor.distribution, "min_std", 0.01))
            lo, hi = self.actor.distribution.bound_tensors() if gauss else (None, None)
            a.act_lo = None if lo is None else lo.data_ptr()
            a.act_hi = None if hi is None else hi.data_ptr()
            self._step_args = a
        K._req(obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous() and obs.numel() == E * a.actor.in_dim,
               "rollout_step: obs must be a contiguous float32 device tensor [E, obs_dim]")
        K._req(critic_obs.is_cuda and critic_obs.dtype == torch.float32 and critic_obs.is_contiguous()
               and critic_obs.numel() == E * a.critic.in_dim,
               "rollout_step: critic_obs must be a contiguous float32 device tensor [E, critic_obs_dim]")
        a.obs = obs.data_ptr(); a.critic_obs = critic_obs.data_ptr()
        a.forced_raw_action = None
        if forced_raw_action is not None:
            want = torch.float32 if a.head_kind == K.HEAD_GAUSSIAN else torch.int64
            K._req(forced_raw_action.is_cuda and forced_raw_action.dtype == want and forced_raw_action.is_contiguous()
                   and forced_raw_action.numel() == buf.raw_actions[t].numel(),
                   "rollout_step: forced_raw_action must be a contiguous device tensor shaped like the step's raw actions")
            a.forced_raw_action = forced_raw_action.data_ptr()
        a.seed, a.offset = self.actor.distribution.rng.take(E)
        if value_normalizer is not None:
            a.normalize_values = 1
            a.vn_mean = value_normalizer.running_stats.mean_t.data_ptr()
            a.vn_var = value_normalizer.running_stats.var_t.data_ptr()
        else:
            a.normalize_values = 0
        a.raw_action_out = buf.raw_actions[t].data_ptr(); a.action_out = buf.actions[t].data_ptr()
        a.logp_out = buf.log_probs[t].data_ptr(); a.value_out = buf.values[t].data_ptr()
        a.obs_copy_out = buf.observations[t].data_ptr()
        a.critic_obs_copy_out = buf.critic_observations[t].data_ptr()
        K.policy_step(a)
        return buf.actions[t]

    def finish_step(self, t, rewards, next_obs=None):
        """The environment's answer for step t: rewards (and next observations when ICM keeps them)."""
        buf = self.buffer
        buf.rewards[t].copy_(rewards.reshape(-1))
        if buf.next_observations is not None and next_obs is not None:
            buf.next_observations[t].copy_(next_obs.reshape(buf.next_observations[t].shape))
        buf.steps_written = max(buf.steps_written, t + 1)
        self._t = t + 1

    def get_intrinsic_reward(self, prev_obs, obs, action):
        """ppo_policy.py:954-1007: ICM forward without gradients -> intrinsic reward [n] (already weighted)."""
        if len(obs.shape) < 2:
            raise ValueError(f"get_intrinsic_reward expects a batch of observations, got shape {obs.shape}")
        obs_1 = self._to_device(prev_obs)
        obs_2 = self._to_device(obs)
        adt = torch.int64 if self.action_dtype in ("discrete", "multi-discrete") else torch.float32
        act = self._to_device(action, adt)
        if act.dim() != 2:
            act = act.unsqueeze(1)
        if obs_1.is_cuda and getattr(self, "fused_icm_reward", True):
            fused = self._fused_intrinsic_reward(obs_1, obs_2, act)
            if fused is not None:
                return fused
        with torch.no_grad():
            intr, _, _ = self.icm_model(obs_1, obs_2, act)
        return intr.reshape(-1) * float(self.intr_reward_weight())

    def _fused_intrinsic_reward(self, obs_1, obs_2, act):
        """K14's encoder + forward-model kernels on the env batch (two launches); None when not covered."""
        import ctypes as C
        from .. import _lib
        from .. import kernels as K
        st = getattr(self, "_icm_reward_state", None)
        n = obs_1.shape[0]
        if st is None or st["n"] != n:
            from ..fused_update import _describe_icm
            topo, why = _describe_icm(self.icm_model, self.action_dtype)
            if topo is None:
                self.fused_icm_reward = False
                return None
            a = _lib.IcmUpdateArgs()
            for k, v in topo.items():
                setattr(a, k, v)
            nT = (n + 15) // 16
            scratch = torch.zeros(2, 4, 16 * nT, topo["hidden"], dtype=torch.float32, device=self.device)
            a.params = self.icm_model.flat_params.data_ptr()
            a.act_scratch = scratch.data_ptr()
            a.B, a.batch_stride, a.n_rows, a.fused_adam = n, n, n, 0
            st = self._icm_reward_state = dict(n=n, args=a, scratch=scratch)
        a = st["args"]
        obs_1, obs_2, act = obs_1.reshape(n, -1).contiguous(), obs_2.reshape(n, -1).contiguous(), act.contiguous()
        a.obs, a.next_obs, a.actions = obs_1.data_ptr(), obs_2.data_ptr(), act.data_ptr()
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        scale = float(self.intr_reward_weight()) * float(self.icm_model.reward_scale) / 2.0
        _lib.check(_lib.load().ppoaf_icm_intrinsic_reward(C.byref(a), scale, out.data_ptr(), K.stream()),
                   "icm_intrinsic_reward")
        return out

    def get_inference_actions(self, obs, deterministic):
        t_obs = self._to_device(obs)
        with torch.no_grad():
            pred = self.actor.forward_logits(t_obs)
            if deterministic:
                return self.actor.distribution.refine_prediction(pred)
            return self.actor.distribution.sample_distribution(pred)[0]

    def get_critic_values(self, obs):
        """ppo_policy.py:1057-1071."""
        return self.critic(obs)

    def add_episode_info(self, agent_id, critic_observations, observations, next_observations,
                         raw_actions, actions, values, log_probs, rewards, where_done):
        """ppo_policy.py:545-651: one env step of E transitions for `agent_id` -> row t of the buffer."""
        col = self._agent_col[agent_id]
        E = self.env_batch_size
        self.buffer.write_step(self._t, slice(col * E, (col + 1) * E), critic_observations,
                               observations, next_observations, raw_actions, actions, values,
                               log_probs, rewards)
        self._agents_written += 1
        if self._agents_written == len(self.agent_ids):      # every agent of this policy has logged step t
            self._agents_written = 0
            self._t += 1

    def end_episodes(self, agent_id, env_idxs, episode_lengths, terminal, ending_values, ending_rewards):
        """
        ppo_policy.py:653-712.  Called after add_episode_info of the same step.
        ending_values / ending_rewards may have one entry per env_idx (what the
        terminal case passes, ppo.py:1817-1819) or one per env (what the
        bootstrapped case passes, ppo.py:1937-1938).  In the second form the
        reference indexes them by POSITION in env_idxs (ppo_policy.py:684-689),
        i.e. env i can receive env j's bootstrap (quirk Q1); here each env gets
        its own value.  The two coincide whenever env_idxs == arange(E).
        """
        if self.frozen:
            return
        col = self._agent_col[agent_id]
        E = self.env_batch_size
        idx = self._to_device(env_idxs, torch.int64).reshape(-1)
        if idx.numel() == 0:
            return
        ev = self._to_device(ending_values).reshape(-1)
        er = self._to_device(ending_rewards).reshape(-1)
        if ev.numel() == E and idx.numel() != E:
            ev = ev[idx]
        if er.numel() == E and idx.numel() != E:
            er = er[idx]
        t = self._t - 1 if self._agents_written == 0 else self._t
        self.buffer.mark_ends(t, idx + col * E, self._to_device(terminal, torch.bool).reshape(-1), ev, er)

    def finalize_dataset(self):
        """ppo_policy.py:714-719."""
        self.dataset.build()

    def clear_dataset(self):
        """ppo_policy.py:721-727 (the HBM buffer is kept for the next rollout)."""
        self.dataset = None

    def get_bs_clip_range(self, ep_rewards):
        """ppo_policy.py:1086-1112."""
        if not self.have_bootstrap_clip:
            return None
        if self.dynamic_bs_clip:
            # :1104-1106: (min, max) of the episode's own rewards.  The device buffer resolves it per episode
            # segment in RolloutBuffer.compute_advantages; this marker selects that path.
            return "dynamic"
        return (self.bootstrap_clip[0](), self.bootstrap_clip[1]())

    # ----------------------------------------------------------------- update
    def evaluate(self, batch_critic_obs, batch_obs, batch_actions):
        """ppo_policy.py:891-952 -> (values, log_probs [B,1], entropy [B])."""
        values = self.critic(batch_critic_obs).squeeze()
        pred = self.actor.forward_logits(batch_obs)
        log_probs, entropy = self.actor.distribution.get_log_probs_and_entropy(pred, batch_actions)
        return values, log_probs, entropy

    def update_weights(self, actor_loss, critic_loss):
        """ppo_policy.py:1012-1055."""
        if self.frozen:
            return
        scale = 1.0 / mpi_utils.get_num_procs()
        self.policy_grads.zero_()
        actor_loss.backward()
        critic_loss.backward()
        mpi_utils.allreduce_sum_(self.policy_grads)        # actor + critic in one message
        self.actor_optim.step(grad_scale=scale, max_norm=self.gradient_clip)
        self.critic_optim.step(grad_scale=scale, max_norm=self.gradient_clip)

    def optimizer_step(self, grad_scale):
        """clip + Adam for both networks (the tail of update_weights, on already averaged-by-sum gradients)."""
        self.actor_optim.step(grad_scale=grad_scale, max_norm=self.gradient_clip)
        self.critic_optim.step(grad_scale=grad_scale, max_norm=self.gradient_clip)

    def update_learning_rate(self):
        """ppo_policy.py:1073-1084."""
        if self.frozen:
            return
        self.actor_optim.set_lr(self.lr())
        self.critic_optim.set_lr(self.lr())
        if self.enable_icm:
            self.icm_optim.set_lr(self.icm_lr())

    # ------------------------------------------------------------- save / load
    def save(self, save_path, tag="latest"):
        """ppo_policy.py:1215-1247 (`<name>-policy/<tag>/{actor,critic}_<rank>.model`, `*_optim_<rank>`)."""
        policy_save_path = os.path.join(save_path, f"{self.name}-policy", str(tag))     # :1164-1165: any tag type
        os.makedirs(policy_save_path, exist_ok=True)
        self.actor.save(policy_save_path)
        self.critic.save(policy_save_path)
        if self.enable_icm:
            self.icm_model.save(policy_save_path)
        r = mpi_utils.get_rank()
        torch.save(self.actor_optim.state_dict(), os.path.join(policy_save_path, f"actor_optim_{r}"))
        torch.save(self.critic_optim.state_dict(), os.path.join(policy_save_path, f"critic_optim_{r}"))
        if self.enable_icm:
            torch.save(self.icm_optim.state_dict(), os.path.join(policy_save_path, f"icm_optim_{r}"))

    def load(self, load_path, tag="latest"):
        policy_load_path = os.path.join(load_path, f"{self.name}-policy", str(tag))
        self.actor.load(policy_load_path)
        self.critic.load(policy_load_path)
        if self.enable_icm:
            self.icm_model.load(policy_load_path)
        r = mpi_utils.get_rank()
        opts = [("actor", self.actor_optim), ("critic", self.critic_optim)]
        if self.enable_icm:
            opts.append(("icm", self.icm_optim))
        for net, opt in opts:
            f = os.path.join(policy_load_path, f"{net}_optim_{r}")
            if not os.path.exists(f):
                f = os.path.join(policy_load_path, f"{net}_optim_0")
            if net == "icm" and not os.path.exists(f):
                continue                                   # checkpoints written before the ICM optimiser was saved
            opt.load_state_dict(torch.load(f, map_location="cpu", weights_only=False))

    def direct_load(self, policy_load_path):
        """ppo_policy.py:1287-1300: networks only, from the directory itself (no `<name>-policy/<tag>` below it)."""
        self.actor.load(policy_load_path)
        self.critic.load(policy_load_path)
        if self.enable_icm:
            self.icm_model.load(policy_load_path)

    # ------------------------------------------------------------- hooks
    def apply_step_constraints(self, *args):
        """ppo_policy.py:1114-1135: identity; policies that must alter what the environment returns override it
        (args = obs, critic_obs, reward, terminated, truncated, info as PPO.apply_policy_step_constraints passes them)."""
        return args

    def apply_reset_constraints(self, *args):
        """ppo_policy.py:1137-1151: identity (args = obs, critic_obs)."""
        return args

    def get_agent_shared_intrinsic_rewards(self, *args):
        """ppo_policy.py:1009-1010: only agent-grouped policies can share an ICM."""
        raise NotImplementedError

    def __eq__(self, other):
        return isinstance(other, PPOPolicy) and self.name == other.name
