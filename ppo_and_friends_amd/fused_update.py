"""
Host driver of the fused mini-batch update kernels (K12, csrc/ppo_update.hip).

One epoch of PPO._ppo_batch_train (ppo.py:2274-2485) for an MLP actor/critic:
  begin_epoch : value-normaliser records of every mini-batch (one launch, one
                all-gather across ranks), cursor / totals reset
  run_epoch   : per mini-batch  fwd_bwd -> reduce -> [all-reduce] -> adam;
                on a single rank, `graph_chunk` consecutive mini-batches are
                captured once into a hipGraph (3 kernel nodes each, all reading
                the device cursor) and replayed
  end_epoch   : normaliser state back to its owner, totals to the host (the only
                host read of the epoch: the KL early stop needs it)
"""
import ctypes as C
import os

import torch
import torch.nn as nn

from . import _lib
from . import kernels as K
from .networks.distributions import CategoricalDistribution, GaussianDistribution
from .networks.feed_forward import FeedForwardNetwork
from .utils import mpi_utils, peer_exchange


def _activation_code(act):
    if isinstance(act, nn.ReLU):
        return K.ACT_RELU
    if isinstance(act, nn.LeakyReLU) and abs(act.negative_slope - 0.01) < 1e-12:
        return K.ACT_LEAKY_RELU
    if isinstance(act, nn.Tanh):
        return K.ACT_TANH
    return None


def _describe(net, bucket, with_log_std):
    """MlpDesc of a FeedForwardNetwork living in `bucket`, or (None, reason)."""
    if not isinstance(net, FeedForwardNetwork) or net.is_embedded:
        return None, "network is not a plain FeedForwardNetwork"
    dims = net.layer_dims()
    if len(dims) < 2:
        return None, "needs at least one hidden layer"
    H = dims[0][1]
    if any(o != H for (_, o) in dims[:-1]) or any(i != H for (i, _) in dims[1:]):
        return None, "hidden layers must share one width"
    if H not in (32, 64, 128, 256):
        return None, f"hidden width {H} is not one of the instantiated widths (32, 64, 128, 256)"
    act = _activation_code(net.activation)
    if act is None:
        return None, f"activation {net.activation} is not one of ReLU / LeakyReLU(0.01) / Tanh"
    out_dim = dims[-1][1]
    if out_dim > 8:
        return None, f"output width {out_dim} > 8"
    d = _lib.MlpDesc()
    d.in_dim, d.hidden, d.depth, d.out_dim, d.activation = dims[0][0], H, len(dims) - 1, out_dim, act
    base = bucket.data_ptr()
    d.offset = (net.flat_params.data_ptr() - base) // 4
    d.size = net.flat_params.numel()
    # the kernel assumes module order weight, bias per Linear, each padded to 4 floats
    off = 0
    lin = [m for m in net.sequential_net.modules() if isinstance(m, nn.Linear)]
    for m in lin:
        for p in (m.weight, m.bias):
            if (p.data_ptr() - net.flat_params.data_ptr()) // 4 != off:
                return None, "parameter layout differs from the kernel's layer table"
            off += (p.numel() + 3) // 4 * 4
    d.log_std_offset = -1
    if with_log_std:
        ls = net.distribution.log_std
        if (ls.data_ptr() - net.flat_params.data_ptr()) // 4 != off:
            return None, "log_std is not placed after the MLP parameters"
        d.log_std_offset = off
        off += (ls.numel() + 3) // 4 * 4
    if off != d.size:
        return None, "network holds parameters the fused kernel does not know about"
    return d, ""


def _reduce_totals(upd, totals):
    """
    End of an epoch: the loss totals summed over ranks -- and, in the same all-reduce, whether any rank's peer
    exchange ran out of time during the epoch.  If one did, the gradients of that step were garbage on that rank:
    every rank then restores rank 0's state and continues on the RCCL path (PPO._heal_replicas), instead of
    training on with diverged replicas or stopping the job.
    """
    t = totals.clone()
    if not upd.multi:
        return t.cpu().numpy()
    broken = 0.0
    for x in (upd.xchg, getattr(upd, "xchg_sp", None)):
        if x is not None and x.status()[1] != 0:
            broken = 1.0
    if hasattr(upd, "_persistent_failure") and upd._persistent_failure():
        broken = 1.0               # a bounded in-kernel wait of this rank ran out: its peers' exchanges timed out on it
    t = torch.cat([t, torch.tensor([broken], dtype=t.dtype, device=t.device)])
    mpi_utils.allreduce_sum_(t)
    out = t.cpu().numpy()
    if out[-1] > 0:
        upd.ppo._heal_replicas("a peer exchange wait ran out of time")
    return out[:-1]


class FusedPolicyUpdate:

    graph_chunk = 32

    @staticmethod
    def unsupported_reason(pol, batch_size):
        """'' when the fused kernels cover this policy, else why not (the torch path is used then)."""
        if pol.using_lstm or pol.agent_grouping:
            return "LSTM / grouped (MAT) policies are not covered by the fused MLP update"
        dist = pol.actor.distribution
        if isinstance(dist, CategoricalDistribution):
            head = K.HEAD_CATEGORICAL
        elif isinstance(dist, GaussianDistribution):
            head = K.HEAD_GAUSSIAN
        else:
            return "unknown action distribution"
        a, why = _describe(pol.actor, pol.policy_params, head == K.HEAD_GAUSSIAN)
        if a is None:
            return "actor: " + why
        c, why = _describe(pol.critic, pol.policy_params, False)
        if c is None:
            return "critic: " + why
        if c.out_dim != 1:
            return "critic must have one output"
        if a.offset != 0 or c.offset != a.size:
            return "actor and critic buckets are not adjacent"
        if (a.hidden, c.hidden) not in ((32, 32), (64, 64), (128, 128), (256, 256), (128, 256), (64, 128)):
            return f"hidden widths (actor {a.hidden}, critic {c.hidden}) are not an instantiated pair"
        if batch_size < 2:
            return "batch size < 2"
        return ""

    def __init__(self, ppo, policy_id):
        self.ppo = ppo
        self.policy_id = policy_id
        pol = ppo.policies[policy_id]
        self.pol = pol
        dev = pol.device
        self.world = mpi_utils.get_num_procs()
        self.multi = mpi_utils.distributed_path()       # collectives + eager launches (N > 1, or its rehearsal)
        self.head = K.HEAD_GAUSSIAN if isinstance(pol.actor.distribution, GaussianDistribution) \
            else K.HEAD_CATEGORICAL
        self.actor_desc, _ = _describe(pol.actor, pol.policy_params, self.head == K.HEAD_GAUSSIAN)
        self.critic_desc, _ = _describe(pol.critic, pol.policy_params, False)
        self.B = ppo.batch_size
        self.n_wg = (self.B + K.UPDATE_ROWS_PER_WG - 1) // K.UPDATE_ROWS_PER_WG
        total = pol.policy_params.numel()
        self.slabs = torch.zeros(self.n_wg, total, dtype=torch.float32, device=dev)
        self.cursor = torch.zeros(1, dtype=torch.int64, device=dev)
        self.vn_mean = torch.zeros(2, dtype=torch.float32, device=dev)
        self.vn_var = torch.ones(2, dtype=torch.float32, device=dev)
        self.vn_count = torch.full((2,), 1e-4, dtype=torch.float64, device=dev)
        self.loss_partials = torch.zeros(2, self.n_wg, 8, dtype=torch.float32, device=dev)
        self.totals = torch.zeros(9, dtype=torch.float64, device=dev)
        self._lib = _lib.load()
        self.records = None
        self.adv_records = None
        self.perm = None
        self._graphs = {}
        self._args = {}
        self._split_space = None
        # N > 1: the per-mini-batch gradient exchange.  K17 over peer mappings when every rank can (same
        # host, IPC + self-test passed: collective decision), else the RCCL all-reduce in an eager loop.
        self.xchg, self.xchg_reason = (peer_exchange.open_exchange(total, dev) if self.multi else (None, "single rank"))
        # the fused tail launch of the split-wgrad chain (csrc/ppo_update_tail.hip) carries the exchange as a phase of
        # every weight-gradient job (one exchange group per workgroup, job-major tiles in the slots: an object of its own
        # again; at most 512 workgroups = the flag words of one exchange object): two launches per mini-batch on N > 1 ranks too.
        self.xchg_sp = None
        if self.xchg is not None and type(self) is FusedPolicyUpdate and os.environ.get("PPOAF_FUSED_TAIL", "1") != "0" \
                and os.environ.get("PPOAF_SPLIT_WGRAD", "auto") != "0" and self._split_blocks() <= 512 \
                and max(self.actor_desc.in_dim, self.critic_desc.in_dim) <= 64 and self.B <= 512:
            self.xchg_sp, why = peer_exchange.open_exchange(self._tail_exchange_floats(), dev)
            if self.xchg_sp is not None and self.xchg_sp.status()[2] == 3:      # (the same kind on every rank: a collective choice)
                self.xchg_sp.close()
                self.xchg_sp, why = None, "coarse-grained slots need fences, which the fused tail launch does not use"
            if self.xchg_sp is None:
                self.xchg_reason += f"; fused-tail exchange refused ({why})"
        self.split, self.split_reason = self._split_wanted()

    def _split_blocks(self):
        """Workgroups of ppoaf_ppo_update_wgrad for this policy's shapes (csrc/ppo_update_dev.hpp: split_wgrad_blocks)."""
        def jobs(d):
            t = d.hidden // 16
            return (d.depth - 1) * t * ((t + 1) // 2) + t * (((d.in_dim + 15) // 16 + 1) // 2) + 1
        return 8 * ((jobs(self.actor_desc) + jobs(self.critic_desc) + 7) // 8)

    def _tail_exchange_floats(self):
        """Floats of one slot of the fused tail's exchange (csrc/ppo_update_tail.hip: tail_exchange_floats): job-major
        16 x 32 tiles + 16 bias sums per workgroup, then the two networks' output segments (each padded to 4)."""
        def seg(d):
            sz_w0 = (d.hidden * d.in_dim + 3) // 4 * 4
            return d.size - (sz_w0 + d.hidden + (d.depth - 1) * (d.hidden * d.hidden + d.hidden))
        return self._split_blocks() * 528 + sum((seg(d) + 3) // 4 * 4 for d in (self.actor_desc, self.critic_desc))

    def _split_wanted(self):
        """
        (bool, why): the split-wgrad chain (fwd_bwd publishes activation / dz panels, ppoaf_ppo_update_wgrad forms the complete
        weight gradients) instead of weight-gradient slabs + the slab reduce.  PPOAF_SPLIT_WGRAD = auto | 1 | 0; auto =
        shapes the panels cover and, on N > 1 ranks, a K17 exchange plus a 256-wide network: the exchange is then a launch
        of its own between wgrad and Adam (four launches), which beats the slab chain's fused reduce + exchange launch only
        where the split saves more than a launch costs (<8,16>: 46.5 + 6.9 against 59 us; <8,8>: 17.8 + 5.1 against 21.4).
        """
        import os
        mode = os.environ.get("PPOAF_SPLIT_WGRAD", "auto")
        if mode not in ("auto", "0", "1"):
            raise ValueError(f"PPOAF_SPLIT_WGRAD={mode!r}: expected auto, 0 or 1")
        if mode == "0":
            return False, "off (PPOAF_SPLIT_WGRAD=0)"
        if type(self) is not FusedPolicyUpdate:
            return False, "K12 (MLP policies) only"
        if max(self.actor_desc.in_dim, self.critic_desc.in_dim) > 64 or self.B > 512:
            return False, "the panels cover in_dim <= 64 and batch sizes <= 512"
        if self.multi:
            if self.xchg is None:
                return False, "N > 1 without K17: the all-reduce loops run the slab chain"
            if self.xchg_sp is None and mode != "1" and max(self.actor_desc.hidden, self.critic_desc.hidden) < 256:   # (no fused-tail exchange)
                return False, "N > 1, no exchange for the wgrad launch and no 256-wide network: the slab reduce launch carries K17"
        return True, ""

    def gradient_only(self, args, timing_events=(None, None)):
        """fwd_bwd + the launch that completes the gradient bucket (wgrad / slab reduce) of ONE mini-batch, no optimiser step:
        what tests and bench probes compare.  The bookkeeping of that launch (totals, step counters) runs as usual."""
        lib, st, ref = self._lib, K.stream(), C.byref(args)
        _lib.check(lib.ppoaf_ppo_update_fwd_bwd_timed(ref, timing_events[0], timing_events[1], st), "ppo_update_fwd_bwd")
        if args.split_workspace:
            _lib.check(lib.ppoaf_ppo_update_wgrad(ref, st), "ppo_update_wgrad")
        else:
            _lib.check(lib.ppoaf_ppo_update_reduce(ref, 1, st), "ppo_update_reduce")

    # ---- split-wgrad chain, 256-wide networks: a row tile on a PAIR of workgroups (csrc/ppo_update_rowpair.hpp)
    row_pairs = True                   # False: one workgroup per 16-row tile (bitwise the same results; tests compare the two)
    pair_launches = 0                  # fwd_bwd launches issued with row pairs in this process (graph replays not counted)

    def pairs_reason(self):
        """'' when fwd_bwd runs the 256-wide networks' row tiles on workgroup pairs, else why not."""
        if type(self) is not FusedPolicyUpdate:
            return "K12 (MLP policies) only"
        if not self.row_pairs:
            return "off (row_pairs = False)"
        if getattr(self, "_pairs_disabled", ""):
            return "disabled after a failed launch: " + self._pairs_disabled
        if not self.split:
            return "the slab chain runs (" + self.split_reason + ")"
        a, c = self.actor_desc, self.critic_desc
        ok = lambda d: d.hidden == 256 and 2 <= d.depth <= 4
        if not ((a.hidden == 128 and ok(c)) or (ok(a) and ok(c))):
            return "no 256-wide network of depth 2 .. 4 (beside a 128- or 256-wide actor)"
        return ""

    # ---- fused tail of the split-wgrad chain: fwd_bwd -> wgrad + clip norms + Adam in one launch (two launches per mini-batch)
    tail_wait_seconds = 2.0            # bound of the in-kernel wait for the other workgroups' norm records
    tail_launches = 0                  # launches issued in this process (tests: the path really ran; graph replays not counted)

    def tail_reason(self):
        """'' when a mini-batch of the split-wgrad chain ends in ppoaf_ppo_update_wgrad_adam, else why it takes the
        wgrad and Adam launches.  PPOAF_FUSED_TAIL = 1 (default) | 0."""
        import os
        if type(self) is not FusedPolicyUpdate:
            return "K12 (MLP policies) only"
        if os.environ.get("PPOAF_FUSED_TAIL", "1") == "0":
            return "off (PPOAF_FUSED_TAIL=0)"
        if getattr(self, "_tail_disabled", ""):
            return "disabled after a failed launch: " + self._tail_disabled
        if not self.split:
            return "the slab chain runs (" + self.split_reason + ")"
        if self._split_blocks() > 512:
            return f"{self._split_blocks()} weight-gradient workgroups (a polling wave of the fused launch holds 512 records)"
        if self.multi and os.environ.get("PPOAF_SHARE_DEVICE", "0") == "1" and self._split_blocks() * self.world > 512:
            # (tests: R ranks time-share ONE GPU.  Every rank's launch waits for the other ranks' records, so all R launches must
            #  be resident together: 2 x 369 workgroups of a 256-wide critic are not)
            return f"{self.world} ranks share one device: {self._split_blocks()} workgroups each cannot all be resident at once"
        if self.multi and self.xchg_sp is None:
            return "N > 1 without an exchange for the fused tail launch (" + self.xchg_reason + ")"
        return ""

    def _tail_ctl_ptr(self, args):
        ctl = getattr(self, "_tail_ctl", None)
        if ctl is None:
            need = C.c_int64(0)
            _lib.check(self._lib.ppoaf_ppo_update_tail_ctl_bytes(C.byref(args), C.byref(need)), "ppo_update_tail_ctl_bytes")
            n = int(need.value)
            if self.xchg_sp is not None:                  # the Python twin of the slot layout must be the library's
                _lib.check(self._lib.ppoaf_ppo_update_tail_exchange_floats(C.byref(args), C.byref(need)), "ppo_update_tail_exchange_floats")
                assert int(need.value) == self._tail_exchange_floats(), (int(need.value), self._tail_exchange_floats())
            # zeroed once, then kept: the block carries the launch tag from one launch to the next
            ctl = self._tail_ctl = torch.zeros((n + 63) // 64 * 16, dtype=torch.int32, device=self.pol.device)
        FusedPolicyUpdate.tail_launches += 1
        return ctl.data_ptr()

    # ------------------------------------------------------------------ args
    def _make_args(self, B):
        pol, ppo = self.pol, self.ppo
        buf = pol.buffer
        a = _lib.PpoUpdateArgs()
        a.actor, a.critic = self.actor_desc, self.critic_desc
        a.params = pol.policy_params.data_ptr(); a.grads = pol.policy_grads.data_ptr()
        a.exp_avg = pol.policy_exp_avg.data_ptr(); a.exp_avg_sq = pol.policy_exp_avg_sq.data_ptr()
        a.slabs = self.slabs.data_ptr(); a.bucket_total = pol.policy_params.numel()
        a.step_counts = pol.policy_step_counts.data_ptr(); a.lr = pol.policy_lr.data_ptr()
        a.norm_scratch = pol.policy_norm_scratch.data_ptr()
        a.beta1, a.beta2, a.adam_eps = 0.9, 0.999, 1e-5
        a.grad_scale = 1.0 / self.world
        a.max_norm = float(pol.gradient_clip) if pol.gradient_clip is not None else 0.0
        a.head_kind = self.head
        # inputs come from the per-epoch tables in shuffled order (begin_epoch): no index -> data dependent load
        t = self.tables
        a.obs = t["obs"].data_ptr(); a.critic_obs = t["critic_obs"].data_ptr()
        a.raw_actions = t["raw_actions"].data_ptr()
        a.advantages = t["advantages"].data_ptr(); a.old_log_probs = t["log_probs"].data_ptr()
        a.rewards_to_go = t["rewards_to_go"].data_ptr(); a.values = buf.values.data_ptr()
        a.inputs_in_batch_order = 1
        a.perm = self.rows.data_ptr(); a.row_map = None      # rows = row_map[perm], resolved once per epoch
        a.n_rows = buf.num_transitions
        a.cursor = self.cursor.data_ptr()
        a.B = B; a.batch_stride = self.B
        a.normalize_values = int(bool(ppo.normalize_values)); a.n_ranks = self.world
        a.vn_mean = self.vn_mean.data_ptr(); a.vn_var = self.vn_var.data_ptr()
        a.vn_count = self.vn_count.data_ptr()
        a.vn_records = self.records.data_ptr() if self.records is not None else None
        a.adv_records = self.adv_records.data_ptr() if self.adv_records is not None else None
        a.normalize_adv = int(bool(ppo.normalize_adv)); a.use_huber = int(bool(pol.use_huber_loss))
        a.surr_clip = float(pol.surr_clip); a.entropy_weight = float(pol.entropy_weight())
        a.kl_loss_weight = float(pol.kl_loss_weight); a.huber_delta = 10.0
        a.min_std = float(getattr(pol.actor.distribution, "min_std", 0.01))
        a.loss_partials = self.loss_partials.data_ptr(); a.totals = self.totals.data_ptr()
        a.mb_offset, a.cursor_advance = 0, 1
        a.xcd_half = getattr(self, "xcd_half", 0)        # 1 / 2: beside the ICM chain (ppo.py: _ppo_icm_epoch_overlapped)
        a.split_workspace, a.split_workspace_bytes = None, 0
        a.row_pairs = 0
        if self.split:
            a.row_pairs = int(self.pairs_reason() == "")
            if self._split_space is None:            # sized once for the full batch size; a tail mini-batch needs less
                need = C.c_int64(0)
                a.row_pairs = int(self.row_pairs)    # (room for the pairs' records whether or not they stay switched on)
                _lib.check(self._lib.ppoaf_ppo_update_split_workspace_bytes(C.byref(a), C.byref(need)), "split_workspace_bytes")
                self._split_space = torch.zeros(int(need.value), dtype=torch.uint8, device=pol.device)
                off = C.c_int64(-1)
                _lib.check(self._lib.ppoaf_ppo_update_row_pairs_error_offset(C.byref(a), C.byref(off)), "row_pairs_error_offset")
                self._pair_region = int(off.value)   # -1: these shapes run no pairs
                assert self._pair_region >= 0 or self.pairs_reason() != "", "the library runs no pairs for shapes pairs_reason() accepts"
                a.row_pairs = int(self.pairs_reason() == "")
                blocks = int(self._lib.ppoaf_ppo_update_split_blocks(C.byref(a)))
                if pol.policy_norm_scratch.numel() < 6 + 2 * blocks:      # one pair of norm partials per wgrad workgroup
                    pol.policy_norm_scratch = torch.zeros(6 + 2 * blocks, dtype=torch.float64, device=pol.device)
                    a.norm_scratch = pol.policy_norm_scratch.data_ptr()
            a.split_workspace, a.split_workspace_bytes = self._split_space.data_ptr(), self._split_space.numel()
        return a

    def _signature(self):
        """Everything baked into captured launches; a change re-captures."""
        pol, buf = self.pol, self.pol.buffer
        return (buf.observations.data_ptr(), buf.num_transitions, self.rows.data_ptr(), self.tables["obs"].data_ptr(),
                None if self.records is None else self.records.data_ptr(),
                None if self.adv_records is None else self.adv_records.data_ptr(),
                float(pol.entropy_weight()), float(pol.surr_clip), float(pol.kl_loss_weight),
                bool(pol.use_huber_loss), pol.gradient_clip, bool(self.ppo.normalize_adv),
                bool(self.ppo.normalize_values), getattr(self, "xcd_half", 0))

    # ----------------------------------------------------------------- epoch
    def begin_epoch(self, perm):
        pol, ppo = self.pol, self.ppo
        buf = pol.buffer
        N = perm.numel()
        if self.perm is None or self.perm.numel() != N:
            self.perm = torch.empty(N, dtype=torch.int64, device=pol.device)
            self._graphs.clear()
        self.perm.copy_(perm)
        # one dependent load less per mini-batch: the kernels read the buffer row directly
        if getattr(self, "rows", None) is None or self.rows.numel() != N:
            self.rows = torch.empty(N, dtype=torch.int64, device=pol.device)
            self._graphs.clear()
        torch.index_select(buf.row_map, 0, self.perm, out=self._rows32(N))
        self.rows.copy_(self._rows32(N))
        self._gather_epoch_tables(N)
        nb = (N + self.B - 1) // self.B
        if ppo.normalize_values:
            local = K.minibatch_moments(buf.rewards_to_go.view(-1), self.perm, buf.row_map, self.B)
            if self.multi:
                allr = mpi_utils.allgather_records(local.reshape(-1)).view(self.world, nb, 3)
                rec = allr.permute(1, 0, 2).contiguous()
            else:
                rec = local.view(nb, 1, 3)
            if self.records is None or self.records.shape != rec.shape:
                self.records = torch.empty_like(rec)
                self._graphs.clear()
            self.records.copy_(rec)
            rs = ppo.value_normalizers[self.policy_id].running_stats
            self.vn_mean[0:1].copy_(rs.mean_t); self.vn_var[0:1].copy_(rs.var_t)
            self.vn_count[0:1].copy_(rs.count_t)
        if ppo.normalize_adv:
            if self.adv_records is None or self.adv_records.shape[0] != nb:
                self.adv_records = torch.empty(nb, 3, dtype=torch.float64, device=pol.device)
                self._graphs.clear()
            K.minibatch_moments(buf.advantages.view(-1), self.perm, buf.row_map, self.B, out=self.adv_records)
        self.cursor.zero_()
        self.totals.zero_()
        if self._split_space is not None and getattr(self, "_pair_region", -1) >= 0:
            # the pairs' records are tagged with the mini-batch index, which restarts now
            self._split_space[self._pair_region:].zero_()
        sig = self._signature()
        if self._args.get("sig") != sig:
            self._args = {"sig": sig}
            self._graphs.clear()
        self.n_full, self.tail = N // self.B, N % self.B
        self.n_done = 0

    def _gather_epoch_tables(self, N):
        """K4 over the whole epoch: every input field of the update in shuffled order (one launch)."""
        buf = self.pol.buffer
        flat = lambda x: x.view((buf.num_transitions,) + tuple(x.shape[2:]))
        fields = dict(obs=buf.observations, critic_obs=buf.critic_observations, raw_actions=buf.raw_actions,
                      advantages=buf.advantages, log_probs=buf.log_probs, rewards_to_go=buf.rewards_to_go)
        t = getattr(self, "tables", None)
        if t is None or t["advantages"].shape[0] != N:
            t = self.tables = {k: torch.empty((N,) + tuple(v.shape[2:]), dtype=v.dtype, device=v.device)
                               for k, v in fields.items()}
            self._graphs.clear()
            self._args = {}
        K.minibatch_gather([(flat(v), t[k]) for k, v in fields.items()], self.perm, buf.row_map)

    def _rows32(self, N):
        t = getattr(self, "_rows_i32", None)
        if t is None or t.numel() != N:
            t = self._rows_i32 = torch.empty(N, dtype=torch.int32, device=self.pol.device)
        return t

    def _args_for(self, B):
        if B not in self._args:
            self._args[B] = self._make_args(B)
        return self._args[B]

    def _one(self, args):
        """One mini-batch: 3 launches (+ the gradient all-reduce on N > 1).  This is the eager path of
        multi-rank runs, so the per-call Python overhead is kept minimal: raw ctypes handles, the
        stream pointer looked up once per call."""
        lib = self._lib
        st = K.stream()
        ref = C.byref(args)
        single = not self.multi
        rc = lib.ppoaf_ppo_update_fwd_bwd(ref, st)
        if args.row_pairs:
            self._pairs_used = True
            FusedPolicyUpdate.pair_launches += 1
        if rc == 0 and args.split_workspace:
            # split-wgrad chain: complete weight gradients from the published panels, then clip + Adam
            if self.tail_reason() == "":
                # fused tail (csrc/ppo_update_tail.hip): weight gradients, [N > 1: the K17 exchange of every job's sums,]
                # clip norms and clip + Adam in ONE launch
                if single:
                    rc = lib.ppoaf_ppo_update_wgrad_adam(ref, self._tail_ctl_ptr(args), self.tail_wait_seconds, st)
                else:
                    rc = lib.ppoaf_ppo_update_wgrad_adam_exchange(ref, self._tail_ctl_ptr(args), self.tail_wait_seconds,
                                                                 self.xchg_sp.handle, self.xchg_sp.wait_seconds, st)
                self._tail_used = True
                if rc != 0:
                    _lib.check(rc, "ppo_update_wgrad_adam")
                return
            rc = lib.ppoaf_ppo_update_wgrad(ref, st)
            if rc == 0 and self.xchg is not None:
                # N > 1: K17 sums the bucket over the ranks and leaves both clip norms of the sum
                g = self.pol.policy_grads
                self.xchg.allreduce(g, g, split_floats=self.actor_desc.size, norm_scale=args.grad_scale,
                                    norm_out=self.pol.policy_norm_scratch, stream=st)
                rc = lib.ppoaf_ppo_update_adam(ref, 2, st)
            elif rc == 0:
                rc = lib.ppoaf_ppo_update_adam(ref, 3, st)
            if rc != 0:
                _lib.check(rc, "ppo_update")
            return
        if rc == 0 and self.xchg is not None and self.pol.policy_grads.numel() <= 256 * 1024:
            # slab reduce + K17 exchange in one launch (sums travel from registers to the exchange slot)
            rc = lib.ppoaf_ppo_update_reduce_exchange(ref, self.xchg.handle, self.xchg.wait_seconds, st) \
                or lib.ppoaf_ppo_update_adam_exchanged(ref, self.xchg.handle, st)
            if rc != 0:
                _lib.check(rc, "ppo_update")
            return
        if rc == 0:
            rc = lib.ppoaf_ppo_update_reduce(ref, 1 if single else 0, st)
        if rc == 0 and self.xchg is not None:
            # K17: summed gradients + both clip norms in one launch, then Adam without a norm pass
            g = self.pol.policy_grads
            self.xchg.allreduce(g, g, split_floats=self.actor_desc.size, norm_scale=args.grad_scale,
                                norm_out=self.pol.policy_norm_scratch, stream=st)
            rc = lib.ppoaf_ppo_update_adam(ref, 2, st)
        else:
            if rc == 0 and not single:
                mpi_utils.allreduce_sum_(self.pol.policy_grads)
            if rc == 0:
                rc = lib.ppoaf_ppo_update_adam(ref, 0 if single else 1, st)
        if rc != 0:
            _lib.check(rc, "ppo_update")

    _rccl_comm_cache = "unset"         # process-wide: libppoaf_hip's own RCCL communicator (or None)
    rccl_loop = "c"                    # "python": the fallback's per-mini-batch loop from Python (tests compare the two)

    def _rccl_comm(self):
        """
        The communicator of the C-level fallback loops (`ppoaf_{ppo,icm,mat}_update_chain_allreduce`): a second RCCL
        communicator owned by libppoaf_hip.so, created once per process with the id travelling over torch.distributed.
        None -- on EVERY rank -- when the backend is not RCCL, FusedPolicyUpdate.rccl_loop = "python" asks for the Python loop, or any
        rank cannot bind librccl: that is voted on BEFORE the collective init (ncclCommInitRank blocks until every rank
        has called it, so no rank may enter it alone); a second vote covers an init that returned an error.
        """
        import atexit
        import os
        import torch.distributed as dist
        cls = FusedPolicyUpdate
        if cls._rccl_comm_cache != "unset":
            return cls._rccl_comm_cache
        comm = None
        dev = self.pol.device
        lib = self._lib
        if dist.get_backend() == "nccl" and cls.rccl_loop == "c":
            rank, world = mpi_utils.get_rank(), mpi_utils.get_num_procs()

            def vote(ok):
                v = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(v, op=dist.ReduceOp.MIN)
                return int(v.item()) == 1

            buf = (C.c_char * 128)()
            if vote(lib.ppoaf_comm_unique_id(buf) == 0):                     # every rank can bind librccl (the id call is local)
                msg = torch.zeros(128, dtype=torch.uint8)
                if rank == 0:
                    msg[:] = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8)
                msg = msg.to(dev)
                dist.broadcast(msg, src=0)                                   # rank 0's id is the communicator's
                h = C.c_void_p()
                if lib.ppoaf_comm_init(rank, world, bytes(msg.cpu().numpy().tobytes()), C.byref(h)) == 0:
                    comm = h
                if not vote(comm is not None):
                    if comm is not None:
                        lib.ppoaf_comm_destroy(comm)
                    comm = None
            if comm is not None:
                atexit.register(cls._destroy_rccl_comm)
        cls._rccl_comm_cache = comm
        return comm

    @staticmethod
    def _destroy_rccl_comm():
        cls = FusedPolicyUpdate
        comm, cls._rccl_comm_cache = cls._rccl_comm_cache, None
        if comm not in ("unset", None):
            try:
                torch.cuda.synchronize()
                _lib.load().ppoaf_comm_destroy(comm)
            except Exception:                                                # interpreter shutdown: nothing left to release into
                pass

    def _eager_multi_rank(self, args, n):
        """
        n mini-batches of the N > 1 path: fwd_bwd, reduce, the gradient all-reduce (RCCL), norm + Adam.
        Everything is bound to locals once -- this loop is what the host executes 2048 times per epoch
        while the GPUs wait on each other, so no attribute lookups, wrappers or environment reads inside.
        """
        import torch.distributed as dist
        lib, ref, st = self._lib, C.byref(args), K.stream()
        grads = self.pol.policy_grads
        if not mpi_utils._needs_staging(grads):
            comm = self._rccl_comm()
            if comm is not None:
                # the whole loop from C: 5 launches per mini-batch without returning to Python (host cost below the GPU's)
                left = n
                while left > 0:
                    k = min(left, 256)
                    _lib.check(lib.ppoaf_ppo_update_chain_allreduce(ref, comm, k, st), "ppo_update_chain_allreduce")
                    left -= k
                return
        fwd, red, adam = lib.ppoaf_ppo_update_fwd_bwd, lib.ppoaf_ppo_update_reduce, lib.ppoaf_ppo_update_adam
        if mpi_utils._needs_staging(grads):                  # gloo (tests): through a host copy
            allreduce = lambda: mpi_utils.allreduce_sum_(grads)
        else:
            # (calling the c10d process group object directly, pg.allreduce([t]).wait(), measured 40 % slower)
            allreduce = lambda: dist.all_reduce(grads)
        for _ in range(n):
            rc = fwd(ref, st) or red(ref, 0, st)
            if rc == 0:
                allreduce()
                rc = adam(ref, 1, st)
            if rc != 0:
                _lib.check(rc, "ppo_update")

    def _chunk(self, args, n):
        """n consecutive mini-batches with their index baked in: one cursor update for the whole chain."""
        try:
            for j in range(n):
                args.mb_offset = j
                args.cursor_advance = n if j == n - 1 else 0
                self._one(args)
        finally:
            args.mb_offset, args.cursor_advance = 0, 1

    def _persistent_failure(self):
        """After a host synchronisation: '' or which bounded in-kernel wait of the epoch's launches ran out (row pairs, fused tail)."""
        if getattr(self, "_pairs_used", False):
            self._pairs_used = False
            word = self._split_space[self._pair_region:self._pair_region + 4].view(torch.int32)
            if int(word.item()) != 0:
                word.zero_()
                self._pairs_disabled = "a workgroup's partner did not answer in time"
                self._graphs.clear()                      # the captured chains begin with the paired launch
                self._args = {"sig": self._args.get("sig")}
                return ("ppo_update_fwd_bwd (row pairs): a wait for the partner workgroup's half ran out of time "
                        "(another process on this GPU?)")
        ctl = getattr(self, "_tail_ctl", None)
        if ctl is not None and getattr(self, "_tail_used", False):
            self._tail_used = False
            if int(ctl[2].item()) != 0:                   # TailCtl.error
                ctl[2:3].zero_()
                self._tail_disabled = "a wait for the other workgroups' norm records ran out of time"
                self._graphs.clear()                      # the captured chains end in the fused launch
                return ("ppo_update_wgrad_adam: a wait ran out of time -- the launch's workgroups were not all resident at once "
                        "(another process on this GPU?)")
        return ""

    def _check_persistent(self):
        """Raising form (tests, probes that drive single launches)."""
        why = self._persistent_failure()
        if why:
            raise _lib.PpoafError(why)

    # ---- a launch whose workgroups could not all be resident must not cost the run: the epoch is redone without that form
    def _epoch_state(self):
        pol = self.pol
        return [pol.policy_params, pol.policy_exp_avg, pol.policy_exp_avg_sq, pol.policy_step_counts, pol.policy_norm_scratch,
                self.vn_mean, self.vn_var, self.vn_count, pol.buffer.values]

    def _recover_on_the_chain(self, why):
        """Single rank: the state the epoch began with comes back, the form that failed stays switched off (with the reason:
        _persistent_failure) and the epoch's mini-batches run again."""
        import sys
        print(f"[ppo_and_friends_amd] {why}; restoring the epoch's starting state and running the epoch again without it",
              file=sys.stderr, flush=True)
        for t, keep in zip(self._epoch_state(), self._epoch_snapshot):
            t.copy_(keep)
        self.cursor.zero_()
        self.totals.zero_()
        self.n_done = 0
        self.run_epoch()

    def run_epoch(self):
        args = self._args_for(self.B)
        left = self.n_full
        self._epoch_snapshot = None
        if left > 0 and self.n_done == 0 and (self.tail_reason() == "" or self.pairs_reason() == ""):
            # what the epoch starts from (a few buckets of <= 1 MB: device-to-device copies), should the launch not complete
            self._epoch_snapshot = [t.clone() for t in self._epoch_state()]
        use_graph = self.ppo.use_graphs and (not self.multi or self.xchg is not None)   # RCCL calls are not captured
        chunk = self.graph_chunk if self.n_full < 8 * self.graph_chunk else 4 * self.graph_chunk   # long epochs: fewer, longer graphs
        while left > 0:
            if use_graph and left >= chunk:
                g = self._graphs.get(chunk)
                if g is None:
                    s = torch.cuda.Stream()
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        self._chunk(args, chunk)          # warm-up pass: these mini-batches are real
                    torch.cuda.current_stream().wait_stream(s)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._chunk(args, chunk)          # capture only
                    self._graphs[chunk] = g
                else:
                    g.replay()
                left -= chunk
                self.n_done += chunk
            elif self.multi and self.xchg is None:
                self._eager_multi_rank(args, left)
                self.n_done += left
                left = 0
            else:
                self._one(args)
                left -= 1
                self.n_done += 1
        if self.tail >= 2:
            self._one(self._args_for(self.tail))
            self.n_done += 1

    def end_epoch(self):
        """-> numpy totals[9] (sums of the 8 loss scalars over mini-batches, mini-batch count)."""
        ppo = self.ppo
        if not self.multi and (getattr(self, "_tail_used", False) or getattr(self, "_pairs_used", False)):
            torch.cuda.current_stream().synchronize()
            why = self._persistent_failure()
            if why:                                       # before anything of the failed epoch reaches the normaliser
                if self._epoch_snapshot is None:
                    raise _lib.PpoafError(why)
                self._recover_on_the_chain(why)
        if ppo.normalize_values:
            rs = ppo.value_normalizers[self.policy_id].running_stats
            slot = self.n_done & 1
            rs.mean_t.copy_(self.vn_mean[slot:slot + 1]); rs.var_t.copy_(self.vn_var[slot:slot + 1])
            rs.count_t.copy_(self.vn_count[slot:slot + 1])
            if self.tail == 1:
                # ppo.py:2299-2306: a size-1 batch still updates the normaliser, then is skipped (quirk Q9)
                rs.integrate_records(self.records[self.n_full].contiguous())
        return _reduce_totals(self, self.totals)         # synchronises with the device (N > 1: a failed launch is voted on there)


# ======================================================================================
# K14: fused ICM update
# ======================================================================================
def _describe_icm(icm, action_dtype):
    """IcmUpdateArgs topology fields of an ICM living in one flat bucket, or (None, reason)."""
    from .networks.icm import ICM, LinearObservationEncoder
    if not isinstance(icm, ICM) or not isinstance(icm.obs_encoder, LinearObservationEncoder):
        return None, "ICM with a LinearObservationEncoder is what the fused kernels cover"
    enc = [icm.obs_encoder.enc_1, icm.obs_encoder.enc_2, icm.obs_encoder.enc_3, icm.obs_encoder.enc_4]
    inv = [m for m in icm.inv_model.sequential_net.modules() if isinstance(m, nn.Linear)]
    fwd = [m for m in icm.forward_model.sequential_net.modules() if isinstance(m, nn.Linear)]
    H, O = enc[0].out_features, enc[0].in_features
    if H not in (64, 128):
        return None, f"ICM width {H} is not an instantiated width (64, 128)"
    if any((m.in_features, m.out_features) != (H, H) for m in enc[1:]):
        return None, "encoder layers must share one width (encoded_obs_dim == encoder_hidden_size)"
    if len(inv) < 2 or len(fwd) < 2 or len(inv) > 4 or len(fwd) > 4:
        return None, "inverse / forward model need 1..3 hidden layers"
    A, Ain = inv[-1].out_features, fwd[0].in_features - H
    want_inv = [(2 * H, H)] + [(H, H)] * (len(inv) - 2) + [(H, A)]
    want_fwd = [(H + Ain, H)] + [(H, H)] * (len(fwd) - 2) + [(H, H)]
    if [(m.in_features, m.out_features) for m in inv] != want_inv or \
            [(m.in_features, m.out_features) for m in fwd] != want_fwd:
        return None, "inverse / forward model widths must equal the encoder width"
    if not (1 <= A <= 8 and 1 <= Ain <= 8):
        return None, f"action widths ({A}, {Ain}) must be in [1, 8]"
    if action_dtype not in ("discrete", "continuous") or (action_dtype == "discrete" and Ain != A):
        return None, "unsupported action space for the fused ICM update"
    acts = {_activation_code(a) for a in (icm.activation, icm.obs_encoder.activation, icm.inv_model.activation,
                                           icm.forward_model.activation)}
    if len(acts) != 1 or None in acts:
        return None, "activation is not one shared ReLU / LeakyReLU(0.01) / Tanh"
    base = icm.flat_params.data_ptr()
    off, marks = 0, []
    for group in (enc, inv, fwd):
        marks.append(off)
        for m in group:
            for p in (m.weight, m.bias):
                if (p.data_ptr() - base) // 4 != off:
                    return None, "parameter layout differs from the kernel's layer table"
                off += (p.numel() + 3) // 4 * 4
    if off != icm.flat_params.numel():
        return None, "the ICM holds parameters the fused kernel does not know about"
    return dict(obs_dim=O, hidden=H, action_dim=A, fwd_action_dim=Ain, depth_inv=len(inv) - 1,
                depth_fwd=len(fwd) - 1, activation=acts.pop(), discrete=int(action_dtype == "discrete"),
                enc_offset=marks[0], inv_offset=marks[1], fwd_offset=marks[2], bucket_total=off), ""


class FusedIcmUpdate:
    """
    Host driver of K14 (csrc/icm_update.hip): one epoch of PPO._icm_batch_train (ppo.py:2487-2567).
    Per mini-batch: fwd_bwd (3 launches) -> reduce [+ Adam]; with more ranks reduce -> all-reduce ->
    K11 Adam.  On a single rank `graph_chunk` mini-batches are captured into a hipGraph and replayed
    (all launches read the device cursor).
    """

    graph_chunk = 32

    @staticmethod
    def unsupported_reason(pol):
        if not pol.enable_icm:
            return "no ICM"
        if pol.agent_grouping:
            return "agent-grouped policy: the ICM rows are regrouped per mini-batch (ppo.py:2509-2545), torch path"
        _, why = _describe_icm(pol.icm_model, pol.action_dtype)
        return why

    def __init__(self, ppo, policy_id):
        self.ppo, self.policy_id = ppo, policy_id
        pol = self.pol = ppo.policies[policy_id]
        dev = pol.device
        self.topo, _ = _describe_icm(pol.icm_model, pol.action_dtype)
        self.world = mpi_utils.get_num_procs()
        self.multi = mpi_utils.distributed_path()       # collectives + eager launches (N > 1, or its rehearsal)
        self.B = ppo.batch_size
        nT = (self.B + K.UPDATE_ROWS_PER_WG - 1) // K.UPDATE_ROWS_PER_WG
        H, total = self.topo["hidden"], self.topo["bucket_total"]
        self.slabs = torch.zeros(2 * nT, total, dtype=torch.float32, device=dev)
        self.act_scratch = torch.zeros(2, 4, 16 * nT, H, dtype=torch.float32, device=dev)
        self.denc_scratch = torch.zeros(2, 2, 16 * nT, H, dtype=torch.float32, device=dev)
        self.loss_partials = torch.zeros(nT + 1, 2, dtype=torch.float32, device=dev)    # + the step's Adam constants
        self.totals = torch.zeros(2, dtype=torch.float64, device=dev)
        self.cursor = torch.zeros(1, dtype=torch.int64, device=dev)
        self._lib = _lib.load()
        self.perm = None
        self._graphs, self._args = {}, {}
        self.xchg, self.xchg_reason = (peer_exchange.open_exchange(pol.icm_model.flat_grads.numel(), dev)
                                       if self.multi else (None, "single rank"))
        # split-wgrad chain (csrc/icm_update.hip: icm_wgrad_kernel): PPOAF_SPLIT_WGRAD = auto | 1 (default) | 0.  The reduce entry
        # point keeps its contract, so graphs, K17 and the RCCL loop are the same with either form.
        mode = "0" if os.environ.get("PPOAF_SPLIT_WGRAD", "auto") == "0" else "1"
        self.split = mode == "1"
        self._split_space = None

    def _make_args(self, B):
        pol, buf, opt = self.pol, self.pol.buffer, self.pol.icm_optim
        a = _lib.IcmUpdateArgs()
        for k, v in self.topo.items():
            setattr(a, k, v)
        icm = pol.icm_model
        a.params, a.grads = icm.flat_params.data_ptr(), icm.flat_grads.data_ptr()
        a.exp_avg, a.exp_avg_sq = opt.exp_avg.data_ptr(), opt.exp_avg_sq.data_ptr()
        a.slabs = self.slabs.data_ptr()
        a.step_count, a.lr = opt.step_count.data_ptr(), opt.lr.data_ptr()
        a.beta1, a.beta2, a.adam_eps = opt.betas[0], opt.betas[1], opt.eps
        a.grad_scale = 1.0 / self.world
        t = self.tables                        # per-epoch inputs in shuffled order (begin_epoch)
        a.obs, a.next_obs, a.actions = t["obs"].data_ptr(), t["next_obs"].data_ptr(), t["actions"].data_ptr()
        a.inputs_in_batch_order = 1
        a.perm, a.row_map, a.n_rows = self.perm.data_ptr(), buf.row_map.data_ptr(), buf.num_transitions
        a.cursor, a.B, a.batch_stride = self.cursor.data_ptr(), B, self.B
        a.icm_beta = float(pol.icm_beta)
        a.fused_adam = int(not self.multi)
        a.act_scratch, a.denc_scratch = self.act_scratch.data_ptr(), self.denc_scratch.data_ptr()
        a.loss_partials, a.totals = self.loss_partials.data_ptr(), self.totals.data_ptr()
        a.xcd_half = getattr(self, "xcd_half", 0)
        a.split_workspace, a.split_workspace_bytes = None, 0
        a.fuse_kernels = 0
        if self.split:
            # one launch for the encoder / model / encoder-backward kernels (csrc/icm_update.hip: icm_fused_kernel); its exchange
            # records sit at the start of the workspace, so the flag must not change once the workspace exists
            a.fuse_kernels = int(self.fuse_kernels and not getattr(self, "_fuse_disabled", ""))
            if self._split_space is None:                # sized once, for the full batch size (a tail mini-batch needs less)
                need = C.c_int64(0)
                _lib.check(self._lib.ppoaf_icm_update_split_workspace_bytes(C.byref(a), C.byref(need)), "icm_update_split_workspace_bytes")
                self._split_space = torch.zeros(int(need.value), dtype=torch.uint8, device=pol.device)
                self._split_fused_layout = a.fuse_kernels
            elif self._split_fused_layout != a.fuse_kernels:      # switched off after a failed launch: panels move to the front
                self._split_fused_layout = a.fuse_kernels
            a.split_workspace, a.split_workspace_bytes = self._split_space.data_ptr(), self._split_space.numel()
            self._fuses = a.fuse_kernels == 1 and self._lib.ppoaf_icm_update_fuses_kernels(C.byref(a)) == 1
        return a

    fuse_kernels = True                # False: the three kernels as three launches (bitwise the same; tests compare the two)
    fused_launches = 0                 # single launches issued in this process (graph replays not counted)
    _REC_BYTES = 256 + 2 * 32 * 2 * 16384     # csrc/icm_update.hip: kIcmRecBytes

    def fuse_reason(self):
        """'' when a mini-batch's encoder / model / encoder-backward kernels run as one launch, else why not."""
        if not self.fuse_kernels:
            return "off (fuse_kernels = False)"
        if getattr(self, "_fuse_disabled", ""):
            return "disabled after a failed launch: " + self._fuse_disabled
        if not self.split:
            return "the slab chain runs (PPOAF_SPLIT_WGRAD=0)"
        self._args_for(self.B)
        return "" if getattr(self, "_fuses", False) else "hidden width other than 128, or no LDS room for the three phases"

    def begin_epoch(self, perm):
        pol, buf = self.pol, self.pol.buffer
        N = perm.numel()
        if self.perm is None or self.perm.numel() != N:
            self.perm = torch.empty(N, dtype=torch.int64, device=pol.device)
            self._graphs.clear()
        self.perm.copy_(perm)
        flat = lambda x: x.view((buf.num_transitions,) + tuple(x.shape[2:]))
        fields = dict(obs=buf.observations, next_obs=buf.next_observations, actions=buf.actions)
        t = getattr(self, "tables", None)
        if t is None or t["obs"].shape[0] != N:
            t = self.tables = {k: torch.empty((N,) + tuple(v.shape[2:]), dtype=v.dtype, device=v.device)
                               for k, v in fields.items()}
            self._graphs.clear()
            self._args = {}
        K.minibatch_gather([(flat(v), t[k]) for k, v in fields.items()], self.perm, buf.row_map)   # one launch per epoch
        self.cursor.zero_()
        self.totals.zero_()
        if self._split_space is not None and getattr(self, "_split_fused_layout", 0):
            self._split_space[:self._REC_BYTES].zero_()         # the exchange records are tagged with the cursor, which restarts now
        sig = (t["obs"].data_ptr(), buf.observations.data_ptr(), buf.next_observations.data_ptr(), buf.actions.data_ptr(),
               buf.num_transitions, self.perm.data_ptr(), float(pol.icm_beta), getattr(self, "xcd_half", 0))
        if self._args.get("sig") != sig:
            self._args = {"sig": sig}
            self._graphs.clear()
        self.n_full, self.tail = N // self.B, N % self.B

    def _args_for(self, B):
        if B not in self._args:
            self._args[B] = self._make_args(B)
        return self._args[B]

    def _one(self, args):
        lib, st, ref = self._lib, K.stream(), C.byref(args)
        rc = lib.ppoaf_icm_update_fwd_bwd(ref, st)
        if args.fuse_kernels and getattr(self, "_fuses", False):
            self._fused_used = True
            FusedIcmUpdate.fused_launches += 1
        if rc == 0:
            rc = lib.ppoaf_icm_update_reduce(ref, st)
        if rc != 0:
            _lib.check(rc, "icm_update")
        if self.multi:
            g = self.pol.icm_model.flat_grads
            if self.xchg is not None:
                self.xchg.allreduce(g, g, stream=st)                                  # K17, in-graph
            else:
                mpi_utils.allreduce_sum_(g)
            self.pol.icm_optim.step(grad_scale=1.0 / self.world, max_norm=None)

    def _c_loop(self, args, n):
        """The RCCL fallback (no K17 exchange) issued from C: ppoaf_icm_update_chain_allreduce, <= 256 mini-batches per call.
        False when the library has no RCCL communicator of its own (gloo tests, rccl_loop = "python")."""
        g = self.pol.icm_model.flat_grads
        if not self.multi or self.xchg is not None or mpi_utils._needs_staging(g):
            return False
        comm = FusedPolicyUpdate._rccl_comm(self)
        if comm is None:
            return False
        opt, st, ref = self.pol.icm_optim, K.stream(), C.byref(args)
        while n > 0:
            k = min(n, 256)
            _lib.check(self._lib.ppoaf_icm_update_chain_allreduce(ref, comm, k, opt.norm_scratch.data_ptr(), opt.grad_norm.data_ptr(), st),
                       "icm_update_chain_allreduce")
            n -= k
        return True

    def _epoch_state(self):
        opt, icm = self.pol.icm_optim, self.pol.icm_model
        return [icm.flat_params, opt.exp_avg, opt.exp_avg_sq, opt.step_count]

    def run_epoch(self):
        args = self._args_for(self.B)
        left = self.n_full
        self._fuse_snapshot = None
        if args.fuse_kernels and getattr(self, "_fuses", False) and not self.multi:
            # what the epoch starts from (620 KB at C3), should a partner workgroup not answer (end_epoch)
            self._fuse_snapshot = [t.clone() for t in self._epoch_state()]
        use_graph = self.ppo.use_graphs and (not self.multi or self.xchg is not None)   # RCCL calls are not captured
        chunk = self.graph_chunk if self.n_full < 8 * self.graph_chunk else 4 * self.graph_chunk   # long epochs: fewer, longer graphs
        if left > 0 and self._c_loop(args, left):
            left = 0
        while left > 0:
            if use_graph and left >= chunk:
                g = self._graphs.get(chunk)
                if g is None:
                    s = torch.cuda.Stream()
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        for _ in range(chunk):
                            self._one(args)               # warm-up pass: these mini-batches are real
                    torch.cuda.current_stream().wait_stream(s)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        for _ in range(chunk):
                            self._one(args)               # capture only
                    self._graphs[chunk] = g
                else:
                    g.replay()
                left -= chunk
            else:
                self._one(args)
                left -= 1
        if self.tail:
            self._one(self._args_for(self.tail))

    def _fused_failure(self):
        """After a host synchronisation: '' or why the epoch's single launches did not complete."""
        if not getattr(self, "_fused_used", False):
            return ""
        self._fused_used = False
        word = self._split_space[:4].view(torch.int32)
        if int(word.item()) == 0:
            return ""
        word.zero_()
        return "icm_fused_kernel: a wait for the partner workgroup's records ran out of time (another process on this GPU?)"

    def end_epoch(self):
        """-> numpy [sum of icm_loss over mini-batches, mini-batch count] (summed over ranks)."""
        if getattr(self, "_fused_used", False):
            torch.cuda.current_stream().synchronize()
            why = self._fused_failure()
            if why:
                if self._fuse_snapshot is None:
                    raise _lib.PpoafError(why + ".  Set FusedIcmUpdate.fuse_kernels = False to use three launches.")
                import sys
                print(f"[ppo_and_friends_amd] {why}; restoring the epoch's starting state and continuing with three launches per mini-batch",
                      file=sys.stderr, flush=True)
                self._fuse_disabled = why
                for t, keep in zip(self._epoch_state(), self._fuse_snapshot):
                    t.copy_(keep)
                self.cursor.zero_()
                self.totals.zero_()
                self._args = {"sig": self._args.get("sig")}
                self._graphs.clear()
                self.run_epoch()
        return _reduce_totals(self, self.totals)


# ======================================================================================
# K15: fused multi-agent-transformer update
# ======================================================================================
_MAT_PARAM_NAMES = (
    ["actor.action_encoder.0.weight", "actor.ln.weight", "actor.ln.bias"]
    + [f"actor.blocks.0.ln{i}.{p}" for i in (1, 2, 3) for p in ("weight", "bias")]
    + [f"actor.blocks.0.attn{i}.{n}.{p}" for i in (1, 2) for n in ("key_net", "query_net", "value_net", "proj")
       for p in ("weight", "bias")]
    + [f"actor.blocks.0.mlp.{i}.{p}" for i in (0, 2) for p in ("weight", "bias")]
    + [f"actor.head.{i}.{p}" for i in (0, 2, 3) for p in ("weight", "bias")]
    + [f"critic.obs_encoder.{i}.{p}" for i in (0, 1) for p in ("weight", "bias")]
    + ["critic.ln.weight", "critic.ln.bias"]
    + [f"critic.blocks.0.ln{i}.{p}" for i in (1, 2) for p in ("weight", "bias")]
    + [f"critic.blocks.0.attn.{n}.{p}" for n in ("key_net", "query_net", "value_net", "proj") for p in ("weight", "bias")]
    + [f"critic.blocks.0.mlp.{i}.{p}" for i in (0, 2) for p in ("weight", "bias")]
    + [f"critic.head.{i}.{p}" for i in (0, 2, 3) for p in ("weight", "bias")])


def _describe_mat(pol):
    """Topology + offset table of a MATPolicy's actor_critic for K15, or (None, reason)."""
    from .networks.multi_agent_transformer import MATActorCritic
    ac = getattr(pol, "actor_critic", None)
    if not isinstance(ac, MATActorCritic):
        return None, "not a MATActorCritic"
    if pol.action_dtype != "discrete":
        return None, "the fused MAT update covers Discrete actions"
    D = ac.actor.embedding_size
    if D != 64 or ac.critic.embedding_size != 64:
        return None, f"embedding {D} (the fused kernel is built for 64)"
    if len(ac.actor.blocks) != 1 or len(ac.critic.blocks) != 1:
        return None, "more than one block"
    atts = [ac.actor.blocks[0].attn1, ac.actor.blocks[0].attn2, ac.critic.blocks[0].attn]
    if any(a.num_heads != 1 for a in atts):
        return None, "more than one attention head"
    gelus = [m for m in ac.modules() if isinstance(m, nn.GELU)]
    others = [m for m in ac.modules() if isinstance(m, (nn.ReLU, nn.Tanh, nn.LeakyReLU, nn.Sigmoid, nn.ELU))]
    if others or any(getattr(m, "approximate", "none") != "none" for m in gelus):
        return None, "activations other than exact GELU"
    named = list(ac.named_parameters())
    if [n for n, _ in named] != _MAT_PARAM_NAMES:
        return None, "parameter list differs from the default MATActorCritic"
    NA, O, A = ac.actor.action_pred_size, ac.critic.in_size, ac.actor.num_agents
    if not (1 <= NA <= 8 and 1 <= O <= 32 and 1 <= A <= 16):
        return None, f"sizes (actions {NA}, obs {O}, agents {A}) outside the fused kernel's limits (8, 32, 16)"
    if ac.actor.action_encoder[0].in_features != NA + 1 or ac.actor.action_encoder[0].bias is not None:
        return None, "action encoder is not the Discrete (start token + one-hot, no bias) form"
    base = ac.flat_params.data_ptr()
    offs, off = [], 0
    for _, p in named:
        if (p.data_ptr() - base) // 4 != off:
            return None, "parameter layout differs from the kernel's table"
        offs.append(off)
        off += (p.numel() + 3) // 4 * 4
    if off != ac.flat_params.numel():
        return None, "bucket holds parameters the fused kernel does not know about"
    return dict(obs_dim=O, num_agents=A, num_actions=NA, embedding=64, offsets=offs, bucket_total=off), ""


class FusedMatUpdate(FusedPolicyUpdate):
    """
    Host driver of K15 (csrc/mat_update.hip).  Same epoch protocol as FusedPolicyUpdate: records of
    every mini-batch up front, then per mini-batch fwd_bwd -> reduce -> [all-reduce] -> K11 clip + Adam,
    `graph_chunk` mini-batches per hipGraph on a single rank.
    """

    @staticmethod
    def unsupported_reason(pol, batch_size):
        if not pol.agent_grouping:
            return "not an agent-grouped policy"
        _, why = _describe_mat(pol)
        if not why and batch_size < 2:
            return "batch size < 2"
        return why

    def __init__(self, ppo, policy_id):
        self.ppo, self.policy_id = ppo, policy_id
        pol = self.pol = ppo.policies[policy_id]
        dev = pol.device
        self.topo, _ = _describe_mat(pol)
        self.world = mpi_utils.get_num_procs()
        self.multi = mpi_utils.distributed_path()
        self.B = ppo.batch_size
        self.per_tile = 16 // self.topo["num_agents"]
        self.n_wg = (self.B + self.per_tile - 1) // self.per_tile
        total = self.topo["bucket_total"]
        self.slabs = torch.zeros(self.n_wg, total, dtype=torch.float32, device=dev)
        self.cursor = torch.zeros(1, dtype=torch.int64, device=dev)
        self.vn_mean = torch.zeros(2, dtype=torch.float32, device=dev)
        self.vn_var = torch.ones(2, dtype=torch.float32, device=dev)
        self.vn_count = torch.full((2,), 1e-4, dtype=torch.float64, device=dev)
        self.loss_partials = torch.zeros(self.n_wg, 8, dtype=torch.float32, device=dev)
        self.totals = torch.zeros(9, dtype=torch.float64, device=dev)
        self._lib = _lib.load()
        self.records = self.adv_records = self.perm = None
        self._graphs, self._args = {}, {}
        self.xchg, self.xchg_reason = (peer_exchange.open_exchange(total, dev) if self.multi else (None, "single rank"))
        # split-wgrad chain (csrc/mat_update.hip: mat_update_wgrad_kernel): PPOAF_SPLIT_WGRAD = auto | 1 (default) | 0.  The reduce
        # entry point keeps its contract (slabs / panels -> gradient bucket), so every path above it -- graphs, K17, the
        # RCCL loops -- is the same with either form.
        mode = "0" if os.environ.get("PPOAF_SPLIT_WGRAD", "auto") == "0" else "1"
        self.split = mode == "1"
        self._split_space = None
        self._norm_partials = {}

    def _make_args(self, B):
        pol, ppo, buf = self.pol, self.ppo, self.pol.buffer
        a = _lib.MatUpdateArgs()
        t = self.topo
        a.obs_dim, a.num_agents, a.num_actions, a.embedding = t["obs_dim"], t["num_agents"], t["num_actions"], 64
        for i, o in enumerate(t["offsets"]):
            a.offsets[i] = o
        a.bucket_total = t["bucket_total"]
        ac = pol.actor_critic
        a.params, a.grads, a.slabs = ac.flat_params.data_ptr(), ac.flat_grads.data_ptr(), self.slabs.data_ptr()
        t = self.tables                        # per-epoch inputs in shuffled order (begin_epoch): no index -> data dependent load
        a.critic_obs, a.raw_actions = t["critic_obs"].data_ptr(), t["raw_actions"].data_ptr()
        a.advantages, a.old_log_probs = t["advantages"].data_ptr(), t["log_probs"].data_ptr()
        a.rewards_to_go, a.values = t["rewards_to_go"].data_ptr(), buf.values.data_ptr()
        a.inputs_in_batch_order = 1
        a.perm, a.row_map, a.n_rows = self.perm.data_ptr(), buf.row_map.data_ptr(), buf.num_transitions
        a.cursor, a.B, a.batch_stride = self.cursor.data_ptr(), B, self.B
        a.normalize_values, a.n_ranks = int(bool(ppo.normalize_values)), self.world
        a.normalize_adv, a.use_huber = int(bool(ppo.normalize_adv)), int(bool(pol.use_huber_loss))
        a.vn_mean, a.vn_var, a.vn_count = self.vn_mean.data_ptr(), self.vn_var.data_ptr(), self.vn_count.data_ptr()
        a.vn_records = self.records.data_ptr() if self.records is not None else None
        a.adv_records = self.adv_records.data_ptr() if self.adv_records is not None else None
        a.surr_clip, a.entropy_weight = float(pol.surr_clip), float(pol.entropy_weight())
        a.kl_loss_weight, a.huber_delta = float(pol.kl_loss_weight), 10.0
        a.loss_partials, a.totals = self.loss_partials.data_ptr(), self.totals.data_ptr()
        opt = pol.actor_critic_optim
        a.norm_scratch, a.step_count = opt.norm_scratch.data_ptr(), opt.step_count.data_ptr()
        # the reduce launch also advances the step count and yields the local ||g||^2 (replaced by K17's norm of the
        # summed gradient when ranks exchange); only the RCCL path runs the separate K11 norm pass
        a.fuse_norm = int(not self.multi or self.xchg is not None)
        a.mb_offset, a.cursor_advance = 0, 1
        a.split_workspace, a.split_workspace_bytes = None, 0
        if self.split:
            if self._split_space is None:                # sized once, for the full batch size (a tail mini-batch needs less)
                need = C.c_int64(0)
                _lib.check(self._lib.ppoaf_mat_update_split_workspace_bytes(C.byref(a), C.byref(need)), "mat_update_split_workspace_bytes")
                self._split_space = torch.zeros(int(need.value), dtype=torch.uint8, device=pol.device)
            a.split_workspace, a.split_workspace_bytes = self._split_space.data_ptr(), self._split_space.numel()
        n = int(self._lib.ppoaf_mat_update_norm_partials(C.byref(a)))
        if n < 0:
            _lib.check(n, "mat_update_norm_partials")
        self._norm_partials[B] = n
        if opt.norm_scratch.numel() < 2 + n:             # one squared-norm partial per workgroup of the reduce launch
            opt.norm_scratch = torch.zeros(2 + n, dtype=torch.float64, device=pol.device)
            a.norm_scratch = opt.norm_scratch.data_ptr()
        return a

    @property
    def norm_partials(self):
        """How ppoaf_adam_step_prenormed finds ||g||^2: the reduce launch's per-workgroup partials (single rank), or
        norm_scratch[0] as K17's exchange left it (0)."""
        if self.xchg is not None:
            return 0
        return self._norm_partials.get(self.B) or (self.topo["bucket_total"] // 4 + 255) // 256

    def _eager_multi_rank(self, args, n):
        """The RCCL fallback (no K17 exchange): fwd_bwd -> reduce -> all-reduce -> K11 clip + Adam per mini-batch, issued from
        C (ppoaf_mat_update_chain_allreduce, <= 256 mini-batches per call) when the library owns an RCCL communicator."""
        pol = self.pol
        opt, ac, clip = pol.actor_critic_optim, pol.actor_critic, pol.gradient_clip
        comm = None if mpi_utils._needs_staging(ac.flat_grads) else FusedPolicyUpdate._rccl_comm(self)
        if comm is None:
            for _ in range(n):
                self._one(args)
            return
        st, ref = K.stream(), C.byref(args)
        while n > 0:
            k = min(n, 256)
            _lib.check(self._lib.ppoaf_mat_update_chain_allreduce(
                ref, comm, k, opt.exp_avg.data_ptr(), opt.exp_avg_sq.data_ptr(), opt.lr.data_ptr(), opt.betas[0], opt.betas[1],
                opt.eps, 1.0 / self.world, float(clip) if clip is not None else 0.0, opt.grad_norm.data_ptr(), st),
                "mat_update_chain_allreduce")
            n -= k

    def begin_epoch(self, perm):
        pol, ppo, buf = self.pol, self.ppo, self.pol.buffer
        ds = pol.dataset
        N = perm.numel()
        if self.perm is None or self.perm.numel() != N:
            self.perm = torch.empty(N, dtype=torch.int64, device=pol.device)
            self._graphs.clear()
        self.perm.copy_(perm)
        nb = (N + self.B - 1) // self.B
        # K4 over the whole epoch: every input field of the update in shuffled order (one launch)
        flat = lambda x: x.view((buf.num_transitions,) + tuple(x.shape[2:]))
        fields = dict(critic_obs=buf.critic_observations, raw_actions=buf.raw_actions, advantages=buf.advantages,
                      log_probs=buf.log_probs, rewards_to_go=buf.rewards_to_go)
        t = getattr(self, "tables", None)
        if t is None or t["advantages"].shape[0] != N:
            t = self.tables = {k: torch.empty((N,) + tuple(v.shape[2:]), dtype=v.dtype, device=v.device) for k, v in fields.items()}
            self._graphs.clear()
            self._args = {}
        K.minibatch_gather([(flat(v), t[k]) for k, v in fields.items()], self.perm, buf.row_map)

        def keep(name, rec):
            cur = getattr(self, name)
            if cur is None or cur.shape != rec.shape:
                setattr(self, name, torch.empty_like(rec))
                self._graphs.clear()
            getattr(self, name).copy_(rec)

        if ppo.normalize_values:
            rec = ppo._epoch_records(self.policy_id, ds, self.perm, self.B)             # [R, nb, 3]
            keep("records", rec.permute(1, 0, 2).contiguous())
            rs = ppo.value_normalizers[self.policy_id].running_stats
            self.vn_mean[0:1].copy_(rs.mean_t); self.vn_var[0:1].copy_(rs.var_t); self.vn_count[0:1].copy_(rs.count_t)
        if ppo.normalize_adv:
            keep("adv_records", ppo._epoch_records(self.policy_id, ds, self.perm, self.B, field="advantages",
                                                   gather=False)[0].contiguous())
        self.cursor.zero_()
        self.totals.zero_()
        sig = (buf.critic_observations.data_ptr(), self.tables["critic_obs"].data_ptr(), buf.num_transitions, self.perm.data_ptr(),
               None if self.records is None else self.records.data_ptr(),
               None if self.adv_records is None else self.adv_records.data_ptr(),
               float(pol.entropy_weight()), float(pol.surr_clip), float(pol.kl_loss_weight),
               bool(pol.use_huber_loss), pol.gradient_clip, bool(ppo.normalize_adv), bool(ppo.normalize_values))
        if self._args.get("sig") != sig:
            self._args = {"sig": sig}
            self._graphs.clear()
        self.n_full, self.tail = N // self.B, N % self.B
        self.n_done = 0

    # ---- fused tail of K15's split-wgrad chain (csrc/mat_update.hip: mat_update_wgrad_adam_kernel): weight gradients, the
    # clip norm from tagged records and clip + Adam in ONE launch (single rank) -- two launches per mini-batch
    def tail_reason(self):
        import os
        if os.environ.get("PPOAF_FUSED_TAIL", "1") == "0":
            return "off (PPOAF_FUSED_TAIL=0)"
        if getattr(self, "_tail_disabled", ""):
            return "disabled after a failed launch: " + self._tail_disabled
        if not self.split:
            return "the slab form runs (" + getattr(self, "split_reason", "PPOAF_SPLIT_WGRAD=0") + ")"
        if self.multi:
            return "N > 1: the gradient exchange sits between the weight gradients and the optimiser step"
        return ""

    def _epoch_state(self):
        pol = self.pol
        opt, ac = pol.actor_critic_optim, pol.actor_critic
        return [ac.flat_params, opt.exp_avg, opt.exp_avg_sq, opt.step_count, self.vn_mean, self.vn_var, self.vn_count, pol.buffer.values]

    def _mat_tail_ctl(self, args):
        ctl = getattr(self, "_tail_ctl", None)
        if ctl is None:
            need = C.c_int64(0)
            _lib.check(self._lib.ppoaf_mat_update_tail_ctl_bytes(C.byref(args), C.byref(need)), "mat_update_tail_ctl_bytes")
            ctl = self._tail_ctl = torch.zeros((int(need.value) + 63) // 64 * 16, dtype=torch.int32, device=self.pol.device)
        FusedPolicyUpdate.tail_launches += 1
        return ctl.data_ptr()

    def _one(self, args):
        lib, st, ref = self._lib, K.stream(), C.byref(args)
        rc = lib.ppoaf_mat_update_fwd_bwd(ref, st)
        if rc == 0 and self.tail_reason() == "":
            opt, clip = self.pol.actor_critic_optim, self.pol.gradient_clip
            rc = lib.ppoaf_mat_update_wgrad_adam(
                ref, self._mat_tail_ctl(args), opt.exp_avg.data_ptr(), opt.exp_avg_sq.data_ptr(), opt.lr.data_ptr(), opt.betas[0],
                opt.betas[1], opt.eps, 1.0, float(clip) if clip is not None else 0.0, opt.grad_norm.data_ptr(),
                self.tail_wait_seconds, st)
            self._tail_used = True
            if rc != 0:
                _lib.check(rc, "mat_update_wgrad_adam")
            return
        if rc == 0:
            rc = lib.ppoaf_mat_update_reduce(ref, st)
        if rc != 0:
            _lib.check(rc, "mat_update")
        if self.multi and self.xchg is None:
            mpi_utils.allreduce_sum_(self.pol.policy_grads)
            self.pol.optimizer_step(1.0 / self.world)
            return
        opt, ac = self.pol.actor_critic_optim, self.pol.actor_critic
        clip = self.pol.gradient_clip
        if self.xchg is not None:
            self.xchg.allreduce(ac.flat_grads, ac.flat_grads, norm_scale=1.0 / self.world, norm_out=opt.norm_scratch, stream=st)
        rc = lib.ppoaf_adam_step_prenormed(
            ac.flat_params.data_ptr(), ac.flat_grads.data_ptr(), opt.exp_avg.data_ptr(), opt.exp_avg_sq.data_ptr(),
            ac.flat_params.numel(), opt.step_count.data_ptr(), opt.lr.data_ptr(), opt.betas[0], opt.betas[1], opt.eps,
            1.0 / self.world, float(clip) if clip is not None else 0.0, opt.norm_scratch.data_ptr(), self.norm_partials,
            opt.grad_norm.data_ptr(), st)
        if rc != 0:
            _lib.check(rc, "adam_step_prenormed")
