"""
Rollout buffer + dataset: the stand-in for utils/episode_info.py of the reference.

Reference model (episode_info.py:169-987): one Python `EpisodeInfo` per
(agent, env) trajectory holding Python lists, E*A `add_info` calls per env step,
a Python reverse scan per finished episode, then `PPODataset.build` turns the
lists into tensors in episode-completion order.

MI355X model: one SoA buffer per policy resident in HBM, time-major
`[T, C, .]` with C = agents*envs columns (coalesced writes per env step,
coalesced column walks for the scans).  Episode boundaries are a dense int8
`end_kind[T, C]` (+ the bootstrap value / reward of bootstrapped ends), so
"end_episode" is a flag write and ALL GAE / rewards-to-go scans are ONE launch of
ppoaf_gae_rtg_tmajor at finalize.  The reference's flattened episode order is
kept as an index map (`row_map`: dataset position -> buffer row), never as a
physical reorder; mini-batches are gathered through it.

Classes
  RolloutBuffer  -- the HBM-resident SoA
  PPODataset     -- the reference's Dataset surface over a RolloutBuffer
                    (__len__, __getitem__ 13-tuple, .values, recalculate_advantages)
  EpisodeInfo    -- per-trajectory drop-in of episode_info.py:169-482 for callers
                    that still drive single episodes (one-wave HIP scan)
"""
import numpy as np
import torch

from .. import kernels as K
from .mpi_utils import rank_print

END_NONE, END_TERMINAL, END_BOOTSTRAP = 0, 1, 2


class RolloutBuffer:
    """Device-resident `[T, C, .]` transition store for one policy (C = num_agents * num_envs)."""

    def __init__(self, T, C, obs_dim, critic_obs_dim, action_dim, action_dtype, device,
                 keep_next_observations=False, agents_per_row=1, lstm_spec=None):
        """
        agents_per_row = A > 1 is the layout of agent-grouped policies (MAT): a row is one env and
        carries its A agents side by side -- the [N, A, .] items of PPOSharedEpisodeDataset
        (utils/episode_info.py:584-637,1058-1084).  Scalar fields are then [T, C, A]; the scans see
        them as T x (C*A) columns, the episode bookkeeping (end_kind, row_map) stays per env row.
        """
        self.T, self.C, self.A = int(T), int(C), int(agents_per_row)
        self.device = torch.device(device)
        self.action_dtype = action_dtype
        f32 = dict(dtype=torch.float32, device=self.device)
        adt = torch.int64 if action_dtype in ("discrete", "multi-discrete") else torch.float32
        ag = () if self.A == 1 else (self.A,)
        self.observations = torch.zeros((T, C) + ag + (obs_dim,), **f32)
        self.critic_observations = torch.zeros((T, C) + ag + (critic_obs_dim,), **f32)
        self.next_observations = torch.zeros((T, C) + ag + (obs_dim,), **f32) if keep_next_observations else None
        self.actions = torch.zeros((T, C) + ag + (action_dim,), dtype=adt, device=self.device)
        self.raw_actions = torch.zeros((T, C) + ag + (action_dim,), dtype=adt, device=self.device)
        self.values = torch.zeros((T, C) + ag, **f32)
        self.log_probs = torch.zeros((T, C) + ag, **f32)
        self.rewards = torch.zeros((T, C) + ag, **f32)
        self.end_kind = torch.zeros(T, C, dtype=torch.int8, device=self.device)
        self.boot_value = torch.zeros((T, C) + ag, **f32)
        self.boot_reward = torch.zeros((T, C) + ag, **f32)
        self.advantages = torch.zeros((T, C) + ag, **f32)
        self.rewards_to_go = torch.zeros((T, C) + ag, **f32)
        # LSTM policies: the (hidden, cell) pair of actor and critic AFTER each step, zeroed where the
        # env terminated (ppo_policy.py:598-627); lstm_spec = ((layers, H) actor, (layers, H) critic)
        self.hidden = None
        if lstm_spec is not None:
            (la, ha), (lc, hc) = lstm_spec
            self.hidden = dict(actor_hidden=torch.zeros(T, C, la, ha, **f32), actor_cell=torch.zeros(T, C, la, ha, **f32),
                               critic_hidden=torch.zeros(T, C, lc, hc, **f32), critic_cell=torch.zeros(T, C, lc, hc, **f32))
        self.fixed_length = True          # flips when an end is recorded before row T-1
        self.steps_written = 0
        # persistent across rollouts (hipGraph replays hold these addresses)
        self.row_map = torch.zeros(T * C, dtype=torch.int32, device=self.device)
        self._batch = {}

    @property
    def num_transitions(self):
        return self.T * self.C

    def write_step(self, t, cols, critic_obs, obs, next_obs, raw_actions, actions, values,
                   log_probs, rewards):
        """One env step for the columns `cols` (slice): the batched form of EpisodeInfo.add_info (:303-399)."""
        as_t = self._as_tensor
        self.observations[t, cols].copy_(as_t(obs, torch.float32).reshape(self.observations[t, cols].shape))
        self.critic_observations[t, cols].copy_(
            as_t(critic_obs, torch.float32).reshape(self.critic_observations[t, cols].shape))
        if self.next_observations is not None:
            self.next_observations[t, cols].copy_(
                as_t(next_obs, torch.float32).reshape(self.next_observations[t, cols].shape))
        adt = self.actions.dtype
        self.actions[t, cols].copy_(as_t(actions, adt).reshape(self.actions[t, cols].shape))
        self.raw_actions[t, cols].copy_(as_t(raw_actions, adt).reshape(self.raw_actions[t, cols].shape))
        self.values[t, cols].copy_(as_t(values, torch.float32).reshape(self.values[t, cols].shape))
        self.log_probs[t, cols].copy_(as_t(log_probs, torch.float32).reshape(self.log_probs[t, cols].shape))
        self.rewards[t, cols].copy_(as_t(rewards, torch.float32).reshape(self.rewards[t, cols].shape))
        self.steps_written = max(self.steps_written, t + 1)

    def _as_tensor(self, x, dtype):
        if torch.is_tensor(x):
            return x.detach().to(device=self.device, dtype=dtype)
        return torch.as_tensor(np.asarray(x), dtype=dtype).to(self.device)

    def mark_ends(self, t, col_idxs, terminal, ending_values, ending_rewards):
        """
        Batched EpisodeInfo.end_episode bookkeeping (:419-443): record that the
        trajectories of `col_idxs` end after step t.  The scans run at finalize.
        """
        col_idxs = self._as_tensor(col_idxs, torch.int64)
        if col_idxs.numel() == 0:
            return
        terminal = self._as_tensor(terminal, torch.bool)
        kind = torch.where(terminal, torch.tensor(END_TERMINAL, dtype=torch.int8, device=self.device),
                           torch.tensor(END_BOOTSTRAP, dtype=torch.int8, device=self.device))
        self.end_kind[t].index_copy_(0, col_idxs, kind.expand(col_idxs.numel()).contiguous())
        self.boot_value[t].index_copy_(0, col_idxs, self._as_tensor(ending_values, torch.float32).reshape(-1))
        self.boot_reward[t].index_copy_(0, col_idxs, self._as_tensor(ending_rewards, torch.float32).reshape(-1))
        if t != self.T - 1 or bool(terminal.any()):      # (a terminal end at the last row is not a bootstrapped one)
            self.fixed_length = False

    def compute_advantages(self, gamma, lambd, bootstrap_clip, use_gae, adv_only=False, timing_events=None):
        """All GAE + rewards-to-go scans of the rollout: one launch (K1)."""
        rtg_out = torch.empty_like(self.rewards_to_go) if adv_only else self.rewards_to_go
        T, cols = self.T, self.C * self.A
        v2 = lambda t: t.view(T, cols)
        ek_kernel = None
        if isinstance(bootstrap_clip, str):
            # dynamic_bs_clip (ppo_policy.py:1104-1106): every ending reward is clipped to the (min, max) of its
            # own episode's rewards -- a segment reduction over the buffer, applied to the stored ending rewards
            # in place (idempotent, so recalculate_advantages may run it again)
            assert bootstrap_clip == "dynamic", bootstrap_clip
            ek_kernel = self._clip_boot_rewards_to_episode_range()
            bootstrap_clip = None
        if self.fixed_length:
            K.gae_rtg_tmajor(v2(self.rewards), v2(self.values), self.boot_value[T - 1].reshape(-1),
                             self.boot_reward[T - 1].reshape(-1), None, gamma, lambd, bootstrap_clip,
                             use_gae, v2(self.advantages), v2(rtg_out), timing_events=timing_events)
        else:
            ek = ek_kernel if ek_kernel is not None else (self.end_kind if self.A == 1 else
                                                          self.end_kind.unsqueeze(-1).expand(T, self.C, self.A).contiguous().view(T, cols))
            K.gae_rtg_tmajor(v2(self.rewards), v2(self.values), v2(self.boot_value), v2(self.boot_reward),
                             ek, gamma, lambd, bootstrap_clip, use_gae,
                             v2(self.advantages), v2(rtg_out), timing_events=timing_events)

    def _clip_boot_rewards_to_episode_range(self):
        T, cols = self.T, self.C * self.A
        r = self.rewards.view(T, cols)
        br = self.boot_reward.view(T, cols)
        if self.fixed_length:                       # one episode per column, bootstrapped at the last row
            br[T - 1].copy_(torch.minimum(torch.maximum(br[T - 1], r.amin(0)), r.amax(0)))
            return None
        ek = self.end_kind if self.A == 1 else \
            self.end_kind.unsqueeze(-1).expand(T, self.C, self.A).contiguous().view(T, cols)
        ends = (ek != 0).to(torch.int64)
        seg = torch.cumsum(ends, 0) - ends                                  # episode index of every cell in its column
        seg = (seg + torch.arange(cols, device=self.device, dtype=torch.int64)[None, :] * (T + 1)).reshape(-1)
        n = cols * (T + 1)
        lo = torch.full((n,), float("inf"), dtype=torch.float32, device=self.device)
        hi = torch.full((n,), float("-inf"), dtype=torch.float32, device=self.device)
        lo.scatter_reduce_(0, seg, r.reshape(-1), "amin", include_self=True)
        hi.scatter_reduce_(0, seg, r.reshape(-1), "amax", include_self=True)
        lo_c, hi_c = lo[seg].view(T, cols), hi[seg].view(T, cols)
        clipped = torch.minimum(torch.maximum(br, lo_c), hi_c)
        # The reference clips the ending reward of TERMINAL episodes too (episode_info.py:450-454 runs for every
        # end): their 0.0 becomes the nearer bound whenever the episode's rewards do not straddle zero.  For the
        # scan such an end is a bootstrapped one with ending value 0 and that clipped ending reward; end_kind
        # itself (dataset order, statistics) stays as it is.
        term = ek == 1
        zero_clipped = torch.minimum(torch.maximum(torch.zeros_like(br), lo_c), hi_c)
        br.copy_(torch.where(ek == 2, clipped, torch.where(term, zero_clipped, br)))
        bv = self.boot_value.view(T, cols)
        bv.copy_(torch.where(term, torch.zeros_like(bv), bv))
        return torch.where(term, torch.full_like(ek, 2), ek).contiguous()

    def build_row_map(self):
        """
        dataset position (the reference's episode-completion order: for each t,
        terminal ends by column, then bootstrapped ends by column --
        ppo.py:1810-1819,1873-1877,1932-1938 + combine_episodes :85-118) -> buffer
        row t*C + c, plus the episode lengths in that order.
        """
        T, C = self.T, self.C
        if self.fixed_length:
            n = torch.arange(T * C, device=self.device, dtype=torch.int64)
            self.row_map.copy_(((n % T) * C + n // T).to(torch.int32))
            return self.row_map, torch.full((C,), T, dtype=torch.int64, device=self.device)
        ek = self.end_kind
        ends = ek != 0
        # segment start of every cell: 1 + index of the previous end in its column (0 if none)
        t_idx = torch.arange(T, device=self.device, dtype=torch.int64)[:, None].expand(T, C)
        prev_end = torch.where(ends, t_idx + 1, torch.zeros_like(t_idx))
        prev_end = torch.cat([torch.zeros(1, C, dtype=torch.int64, device=self.device),
                              torch.cummax(prev_end, dim=0).values[:-1]], dim=0)      # start t0 per cell
        # segment end of every cell: next end at or after t (reverse cummin)
        big = torch.full_like(t_idx, T)
        nxt = torch.where(ends, t_idx, big)
        seg_end = torch.flip(torch.cummin(torch.flip(nxt, [0]), dim=0).values, [0])
        if bool((seg_end >= T).any()):
            raise RuntimeError("rollout buffer has an unfinished episode (every column must end at row T-1; "
                               "ppo.py:1870-1871)")
        # order key of a segment ending at (t1, kind, c): (t1, kind-1, c)
        kind_end = torch.gather(ek.to(torch.int64), 0, seg_end)
        seg_len_at_end = torch.where(ends, t_idx + 1 - prev_end, torch.zeros_like(t_idx))     # [T,C]
        lens3 = torch.zeros(T, 2, C, dtype=torch.int64, device=self.device)
        lens3[:, 0][ek == END_TERMINAL] = seg_len_at_end[ek == END_TERMINAL]
        lens3[:, 1][ek == END_BOOTSTRAP] = seg_len_at_end[ek == END_BOOTSTRAP]
        flat_lens = lens3.reshape(-1)
        offs = torch.cumsum(flat_lens, 0) - flat_lens                                         # exclusive
        c_idx = torch.arange(C, device=self.device, dtype=torch.int64)[None, :].expand(T, C)
        key = seg_end * (2 * C) + (kind_end - 1) * C + c_idx
        pos = offs[key] + (t_idx - prev_end)                                                  # dataset position
        self.row_map[pos.reshape(-1)] = (t_idx * C + c_idx).reshape(-1).to(torch.int32)
        ep_lens = flat_lens[flat_lens > 0]
        return self.row_map, ep_lens

    def minibatch_buffers(self, B):
        """Static per-batch-size gather destinations (graph-capture friendly)."""
        if B not in self._batch:
            mk = lambda t: torch.empty((B,) + tuple(t.shape[2:]), dtype=t.dtype, device=self.device)
            d = dict(critic_obs=mk(self.critic_observations), obs=mk(self.observations),
                     raw_actions=mk(self.raw_actions), advantages=mk(self.advantages),
                     log_probs=mk(self.log_probs), rewards_to_go=mk(self.rewards_to_go))
            if self.next_observations is not None:
                d["next_obs"] = mk(self.next_observations)
                d["actions"] = mk(self.actions)
            self._batch[B] = d
        return self._batch[B]


class _ValuesProxy:
    """`dataset.values[batch_idxs] = v` (ppo.py:2340) routed through the row map."""

    def __init__(self, dataset):
        self._d = dataset

    def __setitem__(self, idx, src):
        self._d.scatter_values(self._d._idx_tensor(idx), src)

    def __getitem__(self, idx):
        idx = self._d._idx_tensor(idx)
        b = self._d.buffer
        return b.values.view((b.num_transitions,) + tuple(b.values.shape[2:]))[self._d.row_map[idx].long()]

    def __len__(self):
        return len(self._d)


class _HiddenProxy:
    """`dataset.actor_hidden[batch_idxs] = h` / `[...]` (ppo.py:2462-2466) routed through the row map."""

    def __init__(self, dataset, key):
        self._d, self._k = dataset, key

    def _flat(self):
        t = self._d.buffer.hidden[self._k]
        return t.view((self._d.buffer.num_transitions,) + tuple(t.shape[2:]))

    def __setitem__(self, idx, src):
        self._flat()[self._d.row_map[self._d._idx_tensor(idx)].long()] = src.detach().to(torch.float32)

    def __getitem__(self, idx):
        return self._flat()[self._d.row_map[self._d._idx_tensor(idx)].long()]


class PPODataset:
    """
    The Dataset surface PPO consumes (episode_info.py:647-987) over a RolloutBuffer:
      len(ds), ds[idx] -> 13-tuple (:940-952), ds.values[...] = ..., ds.recalculate_advantages().
    `gather_minibatch(perm)` is the fused form of DataLoader + __getitem__ + collate (K4).
    """

    def __init__(self, device, action_dtype, sequence_length=1):
        """
        sequence_length S > 1 (LSTM policies, episode_info.py:686-720): item i is the WINDOW of dataset
        positions [i, i+S-1]; its observations are that window (zeroed after a terminal transition inside
        it, :775-809,976-987), every other field -- and the index handed back -- belongs to the window's
        LAST position; len = N - (S - 1) (:916-920).  Windows run across episode boundaries exactly as the
        reference's flat concatenation does.
        """
        self.device = torch.device(device)
        self.action_dtype = action_dtype
        self.sequence_length = int(sequence_length)
        if self.sequence_length < 1:
            raise ValueError("sequence_length must be >= 1")
        self.buffer = None
        self.is_built = False
        self.row_map = None
        self.ep_lens = None
        self.gae_args = None

    def attach(self, buffer, gamma, lambd, bootstrap_clip, use_gae):
        self.buffer = buffer
        self.gae_args = (gamma, lambd, bootstrap_clip, use_gae)

    def add_episode(self, episode):
        """
        episode_info.py:689-719: the reference's way of filling a dataset -- finished EpisodeInfo objects, one
        at a time, concatenated in the order they are added.  For callers that drive single episodes; the
        trainer itself writes whole env steps into an attached RolloutBuffer.
        """
        if self.buffer is not None and not getattr(self, "_from_episodes", False):
            raise RuntimeError("this dataset is attached to a rollout buffer; episodes cannot be added to it")
        if not episode.is_finished:
            raise RuntimeError("attempting to add an unfinished episode to the dataset")
        if episode.has_hidden_states:
            raise NotImplementedError("episode-list datasets with LSTM states: use the rollout buffer path (PPOPolicy)")
        self._from_episodes = True
        self.__dict__.setdefault("episodes", []).append(episode)

    def _buffer_from_episodes(self):
        """One column holding the added episodes back to back; episode ends carry the stored (already clipped)
        ending rewards, so the buffer's scan reproduces end_episode and serves recalculate_advantages."""
        eps = self.episodes
        N = sum(len(e.rewards) for e in eps)
        if N == 0:
            raise RuntimeError("attempting to build a dataset without transitions")
        first = eps[0]
        as2d = lambda rows: np.asarray(rows, dtype=np.float32).reshape(len(rows), -1)
        obs_dim, cobs_dim = as2d(first.observations).shape[1], as2d(first.critic_observations).shape[1]
        act = np.asarray(first.actions).reshape(len(first.actions), -1)
        b = RolloutBuffer(N, 1, obs_dim, cobs_dim, act.shape[1], self.action_dtype, self.device, keep_next_observations=True)
        t = 0
        for e in eps:
            L = len(e.rewards)
            sl = slice(t, t + L)
            put = lambda dst, rows, dt=torch.float32: dst[sl, 0].copy_(torch.as_tensor(np.asarray(rows), dtype=dt).reshape(dst[sl, 0].shape))
            put(b.observations, as2d(e.observations)); put(b.next_observations, as2d(e.next_observations))
            put(b.critic_observations, as2d(e.critic_observations))
            put(b.actions, np.asarray(e.actions).reshape(L, -1), b.actions.dtype)
            put(b.raw_actions, np.asarray(e.raw_actions).reshape(L, -1), b.raw_actions.dtype)
            put(b.values, e.values); put(b.rewards, e.rewards)
            put(b.log_probs, [float(x) for x in e.log_probs])
            terminal = bool(getattr(e, "terminal", False))
            b.end_kind[t + L - 1, 0] = END_TERMINAL if terminal else END_BOOTSTRAP
            b.boot_value[t + L - 1, 0] = 0.0 if terminal else e.ending_value
            b.boot_reward[t + L - 1, 0] = 0.0 if terminal else e._ending_reward
            t += L
        b.steps_written = N
        b.fixed_length = False
        self.attach(b, first.gamma, first.lambd, first.bootstrap_clip, first.use_gae)

    def build(self):
        """episode_info.py:745-914: scans + ordering; no list->tensor conversion is left to do."""
        if self.is_built:
            raise RuntimeError("attempting to build a dataset that has already been built")
        if getattr(self, "_from_episodes", False):
            self._buffer_from_episodes()
        b = self.buffer
        if b.steps_written != b.T:
            raise RuntimeError(f"rollout buffer holds {b.steps_written} of {b.T} steps")
        b.compute_advantages(*self.gae_args)
        self.row_map, self.ep_lens = b.build_row_map()
        self.total_timestates = b.num_transitions
        self.values = _ValuesProxy(self)
        if self.sequence_length > 1:
            # terminal flag of every dataset position (last transition of a terminated episode)
            self.terminal_positions = (b.end_kind.view(-1)[self.row_map.long()] == END_TERMINAL)
        if b.hidden is not None:
            for k in ("actor_hidden", "critic_hidden", "actor_cell", "critic_cell"):
                setattr(self, k, _HiddenProxy(self, k))
        self.is_built = True

    def recalculate_advantages(self):
        """episode_info.py:721-743: new values -> new GAE advantages; rewards-to-go stay."""
        if not self.is_built:
            rank_print("WARNING: recalculate_advantages was called before the dataset has been built. Ignoring call.")
            return
        self.buffer.compute_advantages(*self.gae_args, adv_only=True)

    def __len__(self):
        return self.total_timestates - (self.sequence_length - 1)

    def last_positions(self, idx):
        """Dataset position the non-observation fields of item `idx` come from (episode_info.py:960-962)."""
        return idx + (self.sequence_length - 1)

    def window_rows(self, idx):
        """-> (buffer rows [B, S], zero mask [B, S]) of the observation windows of items `idx` [B]."""
        S = self.sequence_length
        pos = idx.reshape(-1, 1) + torch.arange(S, device=self.device, dtype=torch.int64)
        rows = self.row_map[pos].long()
        term = self.terminal_positions[pos].to(torch.int64)
        mask = (torch.cumsum(term, dim=1) - term) > 0          # True strictly after a terminal transition
        return rows, mask

    def gather_sequences(self, perm_batch):
        """
        The S > 1 form of gather_minibatch (torch-ROCm indexing; the LSTM forward that consumes it is
        MIOpen): observation windows [B, S, .] (actor observations zeroed after a terminal, critic
        observations as they are -- episode_info.py:976-987), everything else at the last position.
        """
        b = self.buffer
        N = b.num_transitions
        v = lambda t: t.view((N,) + tuple(t.shape[2:]))
        rows, mask = self.window_rows(perm_batch)
        last = rows[:, -1]
        obs = v(b.observations)[rows]
        obs = obs.masked_fill(mask.unsqueeze(-1), 0.0)
        out = dict(obs=obs, critic_obs=v(b.critic_observations)[rows], raw_actions=v(b.raw_actions)[last],
                   advantages=v(b.advantages)[last], log_probs=v(b.log_probs)[last],
                   rewards_to_go=v(b.rewards_to_go)[last])
        if b.next_observations is not None:
            out["next_obs"] = v(b.next_observations)[rows].masked_fill(mask.unsqueeze(-1), 0.0)
            out["actions"] = v(b.actions)[last]
        if b.hidden is not None:
            for k, t in b.hidden.items():
                out[k] = v(t)[last]
        return out

    def _idx_tensor(self, idx):
        if torch.is_tensor(idx):
            return idx.to(device=self.device, dtype=torch.int64).reshape(-1).contiguous()
        return torch.as_tensor(np.atleast_1d(np.asarray(idx)), dtype=torch.int64).to(self.device)

    # flat views in the reference's order (materialised on demand; not used by the hot loop)
    def _flat(self, t):
        """[T, C, ...] -> [N, ...] in the reference's dataset order (rows of grouped policies keep [A, .])."""
        return t.reshape((self.buffer.num_transitions,) + tuple(t.shape[2:]))[self.row_map.long()]

    @property
    def observations(self): return self._flat(self.buffer.observations)
    @property
    def critic_observations(self): return self._flat(self.buffer.critic_observations)
    @property
    def next_observations(self):
        return None if self.buffer.next_observations is None else self._flat(self.buffer.next_observations)
    @property
    def actions(self): return self._flat(self.buffer.actions)
    @property
    def raw_actions(self): return self._flat(self.buffer.raw_actions)
    @property
    def advantages(self): return self._flat(self.buffer.advantages)
    @property
    def log_probs(self): return self._flat(self.buffer.log_probs)
    @property
    def rewards_to_go(self): return self._flat(self.buffer.rewards_to_go)

    def __getitem__(self, idx):
        """13-tuple of episode_info.py:940-987 (hidden/cell slots are the uint8 zeros of :860-866 without an LSTM)."""
        b = self.buffer
        if self.sequence_length > 1 or b.hidden is not None:
            i = torch.as_tensor([int(idx)], dtype=torch.int64, device=self.device)
            if self.sequence_length > 1:
                g = self.gather_sequences(i)
            else:
                g = self.gather_minibatch(i, out={})
                N = b.num_transitions
                for k, t in b.hidden.items():
                    g[k] = t.view((N,) + tuple(t.shape[2:]))[self.row_map[i].long()]
            nxt = g["next_obs"][0] if "next_obs" in g else torch.zeros_like(g["obs"][0])
            act = g["actions"][0] if "actions" in g else self._flat(b.actions)[int(idx) + self.sequence_length - 1]
            return (g["critic_obs"][0], g["obs"][0], nxt, g["raw_actions"][0], act, g["advantages"][0],
                    g["log_probs"][0], g["rewards_to_go"][0], g["actor_hidden"][0], g["critic_hidden"][0],
                    g["actor_cell"][0], g["critic_cell"][0], int(idx) + self.sequence_length - 1)
        row = int(self.row_map[idx])
        f = lambda t: t.reshape((b.num_transitions,) + tuple(t.shape[2:]))[row]
        empty = torch.zeros((), dtype=torch.uint8)
        nxt = f(b.next_observations) if b.next_observations is not None else torch.zeros_like(f(b.observations))
        return (f(b.critic_observations), f(b.observations), nxt, f(b.raw_actions), f(b.actions),
                f(b.advantages), f(b.log_probs), f(b.rewards_to_go), empty, empty, empty, empty, idx)

    def minibatch_buffers(self, B):
        return self.buffer.minibatch_buffers(B)

    def gather_minibatch(self, perm_batch, out=None):
        """K4: every field of the mini-batch in one launch; rows = row_map[perm_batch]."""
        b = self.buffer
        B = perm_batch.numel()
        if out is not None and len(out) == 0:       # fresh destinations (the __getitem__ path)
            mk = lambda t: torch.empty((B,) + tuple(t.shape[2:]), dtype=t.dtype, device=self.device)
            out.update(critic_obs=mk(b.critic_observations), obs=mk(b.observations), raw_actions=mk(b.raw_actions),
                       advantages=mk(b.advantages), log_probs=mk(b.log_probs), rewards_to_go=mk(b.rewards_to_go))
        out = self.minibatch_buffers(B) if out is None else out
        N = b.num_transitions
        v = lambda t: t.view((N,) + tuple(t.shape[2:]))
        pairs = [(v(b.critic_observations), out["critic_obs"]), (v(b.observations), out["obs"]),
                 (v(b.raw_actions), out["raw_actions"]), (v(b.advantages), out["advantages"]),
                 (v(b.log_probs), out["log_probs"]), (v(b.rewards_to_go), out["rewards_to_go"])]
        if "next_obs" in out:
            pairs += [(v(b.next_observations), out["next_obs"]), (v(b.actions), out["actions"])]
        K.minibatch_gather(pairs, perm_batch, self.row_map)
        return out

    def scatter_values(self, perm_batch, values):
        """ppo.py:2340."""
        b = self.buffer
        if b.A == 1:
            K.scatter_rows_f32(values.detach().reshape(-1).contiguous().float(), perm_batch,
                               b.values.view(-1), self.row_map)
        else:   # grouped rows [A]: one row of A values per dataset position
            rows = self.row_map[perm_batch].long()
            b.values.view(b.num_transitions, b.A)[rows] = values.detach().reshape(-1, b.A).float()


class EpisodeInfo:
    """
    Per-trajectory container with the reference's constructor and methods
    (episode_info.py:169-482) for callers that drive single episodes.  Steps are
    staged on the host exactly as the reference stages them (Python lists); the
    arithmetic of end_episode -- rewards-to-go and GAE scans -- runs on the GPU
    (one wave, ppoaf_gae_rtg_traj).  The throughput path is RolloutBuffer.
    """

    def __init__(self, starting_ts=0, use_gae=False, gamma=0.99, lambd=0.95,
                 bootstrap_clip=(-10., 10.), device="cuda"):
        self.starting_ts = starting_ts
        self.ending_ts = -1
        self.use_gae = use_gae
        self.gamma = gamma
        self.lambd = lambd
        self.bootstrap_clip = bootstrap_clip
        self.device = torch.device(device)
        self.critic_observations, self.observations, self.next_observations = [], [], []
        self.actions, self.raw_actions, self.log_probs = [], [], []
        self.rewards, self.values = [], []
        self.actor_hidden, self.critic_hidden, self.actor_cell, self.critic_cell = [], [], [], []
        self.rewards_to_go = None
        self.advantages = None
        self.length = 0
        self.is_finished = False
        self.has_hidden_states = False

    def add_info(self, observation, next_observation, raw_action, action, value, log_prob, reward,
                 critic_observation=np.empty(0), actor_hidden=np.empty(0), actor_cell=np.empty(0),
                 critic_hidden=np.empty(0), critic_cell=np.empty(0)):
        given = [len(h) > 0 for h in (actor_hidden, actor_cell, critic_hidden, critic_cell)]
        if any(given) and not all(given):          # episode_info.py:372-384
            raise ValueError("if hidden state is provided for either the actor or the critic, both must be provided")
        if given[0]:                               # :386-399: LSTM states of the step, staged as numpy like the rest
            self.has_hidden_states = True
            to_np = lambda h: h.detach().cpu().numpy() if torch.is_tensor(h) else np.asarray(h)
            self.actor_hidden.append(to_np(actor_hidden)); self.critic_hidden.append(to_np(critic_hidden))
            self.actor_cell.append(to_np(actor_cell)); self.critic_cell.append(to_np(critic_cell))
        self.observations.append(observation)
        self.next_observations.append(next_observation)
        self.actions.append(action)
        self.raw_actions.append(raw_action)
        self.values.append(value)
        self.log_probs.append(log_prob)
        self.rewards.append(reward)
        self.critic_observations.append(critic_observation)

    def _scan(self):
        L = len(self.rewards)
        dev = self.device
        r = torch.tensor(np.asarray(self.rewards, dtype=np.float32), device=dev)
        v = torch.tensor(np.asarray(self.values, dtype=np.float32), device=dev)
        ev = torch.tensor([self.ending_value], dtype=torch.float32, device=dev)
        er = torch.tensor([self._ending_reward], dtype=torch.float32, device=dev)
        start = torch.zeros(1, dtype=torch.int64, device=dev)
        ln = torch.tensor([L], dtype=torch.int32, device=dev)
        return K.gae_rtg_traj(r, v, ev, er, start, ln, self.gamma, self.lambd, self.bootstrap_clip,
                              self.use_gae)

    def compute_advantages(self):
        adv, _ = self._scan()
        self.advantages = adv.cpu().numpy()

    def end_episode(self, ending_ts, terminal, ending_value, ending_reward):
        self.ending_ts = ending_ts
        self.terminal = terminal
        self.length = self.ending_ts - self.starting_ts
        self.is_finished = True
        self.ending_value = float(ending_value)
        self._ending_reward = float(ending_reward)
        adv, rtg = self._scan()
        self.rewards_to_go = rtg.cpu().numpy()
        self.advantages = adv.cpu().numpy()
        self.values = np.array(self.values).astype(np.float32)


class AgentSharedEpisode:
    """
    The per-env container of an agent group's episodes (episode_info.py:485-644): one finished EpisodeInfo per
    agent of the team, all of the same length, merged into rows of [A, .] in the order of `agent_ids`.
    """

    def __init__(self, agent_ids):
        self.agent_ids = np.asarray(agent_ids)
        self.agent_limit = len(self.agent_ids)
        self.agent_episodes = [None] * self.agent_limit
        self.added_agents = []
        self.agent_count = 0
        self.is_finished = False
        self.length = 0

    def verify_agent(self, agent_id):
        if agent_id not in self.agent_ids:
            raise KeyError(f"{agent_id} not in list of accepted agents {list(self.agent_ids)}")

    def verify_episode(self, episode):
        if not episode.is_finished:
            raise RuntimeError("attempting to add an unfinished episode to an AgentSharedEpisode")

    def add_episode(self, agent_id, episode):
        if self.is_finished:
            raise RuntimeError("AgentSharedEpisode is finished: cannot add more episodes")
        self.verify_agent(agent_id)
        self.verify_episode(episode)
        self.agent_episodes[int(np.where(self.agent_ids == agent_id)[0][0])] = episode
        self.added_agents.append(agent_id)
        self.agent_count += 1
        if self.agent_count == self.agent_limit:
            lens = {len(e.rewards) for e in self.agent_episodes}
            if len(lens) != 1 or 0 in lens:
                raise RuntimeError(f"mismatched or empty episodes in AgentSharedEpisode: lengths {sorted(lens)}")
            self.length = lens.pop()
            self.is_finished = True


class PPOSharedEpisodeDataset(PPODataset):
    """
    PPODataset whose rows hold the A agents of a team side by side (episode_info.py:990-1084), filled the
    reference's way: every agent's finished episode of env `env_idx` is handed over separately; when all agents
    of that env have delivered, the shared episode joins the dataset (completion order).
    """

    def __init__(self, num_envs, agent_ids, *args, **kw_args):
        super().__init__(*args, **kw_args)
        self.num_envs, self.agent_ids, self.shared = int(num_envs), np.asarray(agent_ids), True
        self.episode_queue = [AgentSharedEpisode(self.agent_ids) for _ in range(self.num_envs)]
        self.episodes = []
        self._from_episodes = True

    def add_shared_episode(self, episode, agent_id, env_idx):
        q = self.episode_queue[env_idx]
        q.add_episode(agent_id, episode)
        if q.is_finished:
            self.episodes.append(q)
            self.episode_queue[env_idx] = AgentSharedEpisode(self.agent_ids)

    def add_episode(self, episode):
        raise RuntimeError("a shared-episode dataset is filled with add_shared_episode(episode, agent_id, env_idx)")

    def _buffer_from_episodes(self):
        """One column of [A, .] rows holding the shared episodes back to back (see PPODataset._buffer_from_episodes)."""
        A = len(self.agent_ids)
        N = sum(s.length for s in self.episodes)
        if N == 0:
            raise RuntimeError("attempting to build a dataset without transitions")
        first = self.episodes[0].agent_episodes[0]
        as2d = lambda rows: np.asarray(rows, dtype=np.float32).reshape(len(rows), -1)
        act_dim = np.asarray(first.actions).reshape(len(first.actions), -1).shape[1]
        b = RolloutBuffer(N, 1, as2d(first.observations).shape[1], as2d(first.critic_observations).shape[1], act_dim,
                          self.action_dtype, self.device, keep_next_observations=True, agents_per_row=A)
        t = 0
        for s in self.episodes:
            L = s.length
            sl = slice(t, t + L)
            terminal = bool(getattr(s.agent_episodes[0], "terminal", False))
            for a, e in enumerate(s.agent_episodes):
                put = lambda dst, rows, dt=torch.float32: dst[sl, 0, a].copy_(
                    torch.as_tensor(np.asarray(rows), dtype=dt).reshape(dst[sl, 0, a].shape))
                put(b.observations, as2d(e.observations)); put(b.next_observations, as2d(e.next_observations))
                put(b.critic_observations, as2d(e.critic_observations))
                put(b.actions, np.asarray(e.actions).reshape(L, -1), b.actions.dtype)
                put(b.raw_actions, np.asarray(e.raw_actions).reshape(L, -1), b.raw_actions.dtype)
                put(b.values, e.values); put(b.rewards, e.rewards)
                put(b.log_probs, [float(x) for x in e.log_probs])
                b.boot_value[t + L - 1, 0, a] = 0.0 if terminal else e.ending_value
                b.boot_reward[t + L - 1, 0, a] = 0.0 if terminal else e._ending_reward
            b.end_kind[t + L - 1, 0] = END_TERMINAL if terminal else END_BOOTSTRAP
            t += L
        b.steps_written = N
        b.fixed_length = False
        self.attach(b, first.gamma, first.lambd, first.bootstrap_clip, first.use_gae)
