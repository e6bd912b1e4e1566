"""
TEST INFRASTRUCTURE (see oracle/__init__.py) -- CPU restatement ("port") of the
reference's whole hot path with the reference's own loop structure, used
  (a) as the end-to-end parity checker for rollout -> GAE -> mini-batch update, and
  (b) as bench.py's `cpu_baseline` (kind "port"): the reference's Python files
      cannot travel to the GPU box, so this port is what is timed on its host cores.

Loop structure kept on purpose (it is where the reference spends its time,
SURVEY.md §6): per-env `add_info` appends with `.item()` calls
(policies/ppo_policy.py:638-651 -> utils/episode_info.py:355-368), a Python
reverse scan per finished episode (:254-262, :289-293), list -> ndarray -> tensor
dataset build (:815-887), torch DataLoader(shuffle=True) over per-sample
13-tuples + default collate (ppo.py:2181-2184, episode_info.py:939-952), the
value normaliser on every mini-batch (ppo.py:2299-2303), per-network backward /
clip_grad_norm_ / Adam(eps=1e-5) (ppo_policy.py:1032-1055).

PINNED end to end by the g12_* fixtures: the unmodified reference's own PPO object driven for whole iterations
over a table-driven env (tests/golden/make_golden_update.py); this port reproduces its datasets, first-mini-batch
losses and raw gradients, per-epoch statistics, final weights and value-normaliser state for the C2 / C3 / C4
layer shapes, the tanh-Gaussian head, ICM and the filter stack (tests/test_oracle_update_golden.py).  The buffer
half is additionally pinned bit-exact through oracle/episode_info_oracle.py (g1-g3).
"""
import time

import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Categorical
from torch.utils.data import DataLoader, Dataset

from . import episode_info_oracle as eo
from . import ppo_loss_oracle as lo
from .running_stats_oracle import RunningMeanStd


def make_mlp(in_size, out_size, hidden=128, depth=3, out_gain=None, activation=None):
    """networks/utils.py:120-191 + init_layer :53-80 (orthogonal, gain sqrt(2) / out_gain, zero bias)."""
    act = nn.ReLU() if activation is None else activation

    def lin(i, o, gain=np.sqrt(2)):
        layer = nn.Linear(i, o)
        nn.init.orthogonal_(layer.weight, gain)
        nn.init.constant_(layer.bias, 0.0)
        return layer

    inner = []
    for _ in range(depth - 1):
        inner += [lin(hidden, hidden), act]
    last = lin(hidden, out_size) if out_gain is None else lin(hidden, out_size, out_gain)
    return nn.Sequential(lin(in_size, hidden), act, nn.Sequential(*inner), last)


class _Episode:
    """The lists of EpisodeInfo (episode_info.py:205-221) and its end_episode."""

    def __init__(self, gamma, lambd, clip, use_gae=True):
        self.gamma, self.lambd, self.clip, self.use_gae = gamma, lambd, clip, use_gae
        self.observations, self.next_observations, self.critic_observations = [], [], []
        self.actions, self.raw_actions, self.values, self.log_probs, self.rewards = [], [], [], [], []

    def add_info(self, observation, next_observation, raw_action, action, value, log_prob, reward,
                 critic_observation):
        self.observations.append(observation)
        self.next_observations.append(next_observation)
        self.actions.append(action)
        self.raw_actions.append(raw_action)
        self.values.append(value)
        self.log_probs.append(log_prob)
        self.rewards.append(reward)
        self.critic_observations.append(critic_observation)

    def end_episode(self, ending_value, ending_reward, rtg_accum="float64"):
        self.ending_value = ending_value
        self.length = len(self.rewards)
        clip = self.clip
        if clip == "dynamic":                      # ppo_policy.py:1104-1106: the episode's own reward range
            clip = (min(self.rewards), max(self.rewards))
        self.advantages, self.rewards_to_go = eo.end_episode(
            self.rewards, self.values, ending_value, ending_reward, self.gamma, self.lambd,
            clip, self.use_gae, rtg_accum)
        self.values = np.array(self.values).astype(np.float32)


class _ListDataset(Dataset):
    """PPODataset.build + __getitem__ (episode_info.py:745-952), discrete actions."""

    def __init__(self, episodes, action_dtype=torch.long):
        obs, nobs, cobs, act, ract, rtg, lp, adv, val = [], [], [], [], [], [], [], [], []
        for ep in episodes:
            obs.extend(ep.observations); nobs.extend(ep.next_observations)
            cobs.extend(ep.critic_observations); act.extend(ep.actions); ract.extend(ep.raw_actions)
            rtg.extend(ep.rewards_to_go); lp.extend(ep.log_probs); adv.extend(ep.advantages)
            val.extend(ep.values)
        self.episodes = episodes
        self.observations = torch.tensor(np.array(obs), dtype=torch.float32)
        self.next_observations = torch.tensor(np.array(nobs), dtype=torch.float32)
        self.critic_observations = torch.tensor(np.array(cobs), dtype=torch.float32)
        self.actions = torch.tensor(np.array(act), dtype=action_dtype)           # episode_info.py:889-906
        self.raw_actions = torch.tensor(np.array(ract), dtype=action_dtype)
        self.rewards_to_go = torch.tensor(np.array(rtg), dtype=torch.float32)
        self.log_probs = torch.tensor(lp, dtype=torch.float32)
        self.advantages = torch.tensor(np.array(adv), dtype=torch.float32)
        self.values = torch.tensor(np.array(val), dtype=torch.float32)
        if self.actions.dim() <= 1:
            self.actions = self.actions.unsqueeze(-1)
            self.raw_actions = self.raw_actions.unsqueeze(-1)
        self.empty = np.zeros(len(self.observations)).astype(np.uint8)

    def __len__(self):
        return len(self.observations)

    def __getitem__(self, idx):
        return (self.critic_observations[idx], self.observations[idx], self.next_observations[idx],
                self.raw_actions[idx], self.actions[idx], self.advantages[idx], self.log_probs[idx],
                self.rewards_to_go[idx], self.empty[idx], self.empty[idx], self.empty[idx],
                self.empty[idx], idx)


class ThreadComm:
    """
    The three collectives the reference's update path uses (mpi4py `comm.allgather`, `comm.allreduce(x, MPI.SUM)`,
    `comm.barrier`) for R "ranks" that are threads of one process: every call is a rendezvous that hands each rank the
    rank-ordered list of everybody's objects; sums fold that list in rank order (parts[0] + parts[1] + ...), as the
    pickle-based collectives of mpi4py apply the Python `+`.  oracle/cpu_ddppo.GlooComm is the same surface for R
    processes over torch.distributed/gloo.
    """

    class Shared:
        def __init__(self, size):
            import threading
            self.size, self.slots, self.barrier = size, [None] * size, threading.Barrier(size)

    def __init__(self, shared, rank):
        self.shared, self.rank, self.size = shared, rank, shared.size

    def allgather(self, x):
        sh = self.shared
        sh.slots[self.rank] = x
        sh.barrier.wait()
        out = list(sh.slots)
        sh.barrier.wait()
        return out

    def allreduce_sum(self, x):
        parts = self.allgather(x)
        out = parts[0]
        for p in parts[1:]:
            out = out + p
        return out

    def barrier(self):
        self.shared.barrier.wait()


def avg_gradients(params, comm):
    """mpi_avg_gradients (utils/mpi_utils.py:89-111): one all-reduce(SUM) / num_procs per parameter TENSOR, written back
    in place; parameters without a gradient are skipped."""
    if comm is None or comm.size == 1:
        return
    for p in params:
        if p.grad is None:
            continue
        g = p.grad.numpy()                                   # CPU tensor: a view, as `.cpu().numpy()` is in the reference
        g[...] = comm.allreduce_sum(g.copy()) / comm.size


class CpuPPO:
    """One rank of the reference's PPO on CPU for a Discrete-action MLP policy and a table-driven env."""

    def __init__(self, obs_dim, n_actions, hidden=128, depth=3, lr=3e-4, gamma=0.99, lambd=0.95,
                 bootstrap_clip=(-100.0, 100.0), surr_clip=0.2, entropy_weight=0.01,
                 gradient_clip=0.5, batch_size=256, normalize_adv=True, normalize_values=True,
                 seed=0, rtg_accum="float64", critic_obs_dim=None, critic_hidden=None,
                 enable_icm=False, icm_lr=3e-4, icm_beta=0.8, intr_reward_weight=1.0, activation=None,
                 continuous=False, act_low=-1.0, act_high=1.0):
        """continuous=True: Box action space of n_actions dims, tanh-Gaussian head with a learned log_std
        (networks/distributions.py:441-694; log_std is a parameter of the ACTOR module, wrappers.py:30-31)."""
        torch.manual_seed(seed)
        critic_obs_dim = obs_dim if critic_obs_dim is None else critic_obs_dim
        critic_hidden = hidden if critic_hidden is None else critic_hidden
        self.actor = make_mlp(obs_dim, n_actions, hidden, depth, out_gain=0.01, activation=activation)     # ppo_policy.py:433-439
        self.critic = make_mlp(critic_obs_dim, 1, critic_hidden, depth, out_gain=1.0, activation=activation)   # :441-446
        self.continuous = continuous
        self.actor_params = list(self.actor.parameters())
        if continuous:                                 # distributions.py:491-492 (std_offset 0.5), :476-483 (bounds as arrays)
            self.log_std = nn.Parameter(torch.as_tensor(-0.5 * np.ones(n_actions, dtype=np.float32)))
            self.actor_params.append(self.log_std)
            self.act_low = np.broadcast_to(np.asarray(act_low, dtype=np.float32), (n_actions,)).copy()
            self.act_high = np.broadcast_to(np.asarray(act_high, dtype=np.float32), (n_actions,)).copy()
        self.actor_optim = torch.optim.Adam(self.actor_params, lr=lr, eps=1e-5)
        self.critic_optim = torch.optim.Adam(self.critic.parameters(), lr=lr, eps=1e-5)
        self.gamma, self.lambd, self.clip = gamma, lambd, bootstrap_clip
        self.surr_clip, self.entropy_weight, self.gradient_clip = surr_clip, entropy_weight, gradient_clip
        self.batch_size = batch_size
        self.normalize_adv, self.normalize_values = normalize_adv, normalize_values
        self.value_stats = RunningMeanStd()
        self.rtg_accum = rtg_accum
        self.loader_generator = torch.Generator().manual_seed(seed)
        self.enable_icm, self.icm_beta, self.intr_reward_weight = enable_icm, icm_beta, intr_reward_weight
        self.intrinsic_score_avg = 0.0                 # status_dict[...]["intrinsic score avg"], ppo.py:518
        if enable_icm:                                 # ppo_policy.py:461-468, 341-343
            from .icm_oracle import ICM
            self.icm = ICM(obs_dim, n_actions, discrete=not continuous)
            self.icm_optim = torch.optim.Adam(self.icm.parameters(), lr=icm_lr, eps=1e-5)

    # ----- value normaliser (utils/misc.py:84-128)
    def _denorm(self, v):
        mean = torch.tensor(self.value_stats.mean, dtype=torch.float32)
        var = torch.tensor(self.value_stats.variance, dtype=torch.float32)
        return mean + v * torch.sqrt(var + torch.tensor([1e-8]))

    def _norm_update(self, x, comm=None):
        data = x.detach().cpu().numpy()
        if comm is not None and comm.size > 1:               # utils/stats.py:47-50: the raw data of ALL ranks, concatenated
            self.value_stats.update(None, gathered=comm.allgather(data))
        else:
            self.value_stats.update(data)
        mean = torch.tensor(self.value_stats.mean, dtype=torch.float32)
        var = torch.tensor(self.value_stats.variance, dtype=torch.float32)
        return (x - mean) / torch.sqrt(var + torch.tensor([1e-8]))

    def values_of(self, obs):
        with torch.no_grad():
            v = self.critic(obs).reshape(-1)
        return self._denorm(v) if self.normalize_values else v

    # ----- rollout (ppo.py:1646-1983) on pre-generated observation / reward tables
    def rollout(self, obs_table, reward_table, actions=None, term_table=None, max_ts_per_ep=None,
                critic_obs_table=None, comm=None):
        """
        obs_table [T+1,E,O], reward_table [T,E] numpy.  actions (optional [T,E])
        replays a recorded rollout instead of sampling.  term_table (optional bool
        [T,E]) marks terminal steps: those episodes end with (0, 0) (ppo.py:1804-1819);
        episodes reaching max_ts_per_ep, and every open episode at the last step,
        end with the critic bootstrap of the next observation (ppo.py:1863-1938).
        Episodes enter the dataset in completion order.
        """
        T, E = reward_table.shape
        nxt_all = obs_table[1:]
        if critic_obs_table is None:
            critic_obs_table = obs_table          # single agent: the critic sees the actor's observation
        new_ep = lambda: _Episode(self.gamma, self.lambd, self.clip)
        episodes = [new_ep() for _ in range(E)]
        finished = []
        ep_ts = np.zeros(E, dtype=np.int64)
        episode_lengths = np.zeros(E, dtype=np.int64)      # reset on termination only (ppo.py:1849)
        total_episodes, total_intr = 0.0, 0.0
        ism = self.intrinsic_score_avg
        for t in range(T):
            ep_ts += 1
            episode_lengths += 1
            obs = obs_table[t]
            t_obs = torch.tensor(obs, dtype=torch.float32)
            if self.continuous:                        # ppo_policy.py:758-794 with the Gaussian head
                with torch.no_grad():
                    mean = self.actor(t_obs)
                    gd = lo.gaussian_dist(mean, self.log_std)
                    raw = gd.sample() if actions is None else torch.as_tensor(actions[t], dtype=torch.float32)
                    a = lo.gaussian_refine(raw, self.act_low, self.act_high)
                    log_prob = lo.gaussian_tanh_logp(mean, self.log_std, raw).unsqueeze(-1)
                a_np, raw_np = a.numpy(), raw.numpy()
            else:
                with torch.no_grad():
                    probs = torch.softmax(self.actor(t_obs), dim=-1)
                dist = Categorical(probs)
                if actions is None:
                    a = dist.sample()
                else:
                    a = torch.as_tensor(actions[t], dtype=torch.long)
                log_prob = torch.unsqueeze(dist.log_prob(a), dim=-1)
                a_np = raw_np = a.unsqueeze(-1).numpy()
            cobs = critic_obs_table[t]
            value = self.values_of(torch.tensor(cobs, dtype=torch.float32)).unsqueeze(-1)
            nxt = obs_table[t + 1]
            rew = reward_table[t].reshape(E, 1).astype(np.float64)
            intr = np.zeros((E, 1), dtype=np.float32)
            if self.enable_icm:                        # ppo.py:1719-1723 -> ppo_policy.py:954-1007
                with torch.no_grad():
                    ir, _, _ = self.icm(t_obs, torch.tensor(nxt_all[t], dtype=torch.float32),
                                        a if self.continuous else a.unsqueeze(1))
                intr = ir.numpy().reshape(E, 1) * self.intr_reward_weight
                rew = rew + intr                       # float64 + float32 (ppo.py:1283)
                total_intr += float(intr.sum())
            for e in range(E):                         # ppo_policy.py:638-651
                episodes[e].add_info(
                    critic_observation=cobs[e], observation=obs[e], next_observation=nxt[e],
                    raw_action=raw_np[e], action=a_np[e], value=value[e].item(),
                    log_prob=log_prob[e], reward=rew[e].item())
            where_term = np.where(term_table[t])[0] if term_table is not None else np.array([], dtype=np.int64)
            for e in where_term:                       # ppo.py:1810-1819
                episodes[e].end_episode(0.0, 0.0, self.rtg_accum)
                finished.append(episodes[e])
                episodes[e] = new_ep()
                ep_ts[e] = 0
                episode_lengths[e] = 0
                total_episodes += 1
            if t == T - 1:                             # ppo.py:1870-1877
                where_maxed = np.arange(E)
            elif max_ts_per_ep is not None:
                where_maxed = np.where(ep_ts >= max_ts_per_ep)[0]
            else:
                where_maxed = np.array([], dtype=np.int64)
            where_maxed = np.setdiff1d(where_maxed, where_term)
            if where_maxed.size > 0:
                next_value = self.values_of(torch.tensor(critic_obs_table[t + 1], dtype=torch.float32))
                for e in where_maxed:                  # ppo.py:1932-1938 (each env its own value: quirk Q1 fixed)
                    nv = nr = next_value[e].item()
                    if self.enable_icm:
                        # ppo.py:1926-1930 "surprise" (per env: quirk Q2 fixed).  Quirk Q12: next_reward is
                        # get_detached_dict(next_value) (ppo.py:1115-1141): on the CPU `.detach().cpu().numpy()`
                        # SHARES the tensor's memory, so the in-place `+=` lands in next_value as well -- the
                        # ending VALUE of the GAE carries the surprise too (pinned by fixture g12_c2_icm)
                        nr = float(np.float32(nr) + (intr[e, 0] - np.float32(ism)))
                        # (a reference running on a CUDA device copies in `.cpu()`: the ending value keeps V(next obs))
                        nv = nr if getattr(self, "reference_device", "cpu") == "cpu" else nv
                    episodes[e].end_episode(nv, nr, self.rtg_accum)
                    finished.append(episodes[e])
                    episodes[e] = new_ep()
                    ep_ts[e] = 0
        if self.enable_icm:                            # ppo.py:1940-1963, 2074-2080
            combined = float(episode_lengths.sum())
            ts_before = max(T * E - combined, 0.0)
            cur_total = total_episodes if total_episodes != 0 else 1.0
            avg_ep_len = combined / E if ts_before == 0 else ts_before / cur_total
            total_episodes += float((episode_lengths / avg_ep_len).sum())
            if comm is not None and comm.size > 1:     # ppo.py:1991, 2076-2077: episodes and intrinsic rewards of every rank
                total_episodes, total_intr = comm.allreduce_sum(total_episodes), comm.allreduce_sum(total_intr)
            self.intrinsic_score_avg = total_intr / (total_episodes / E)
        self.dataset = _ListDataset(finished, torch.float32 if self.continuous else torch.long)
        return self.dataset

    def _loader(self, perm):
        """DataLoader(shuffle=True) (ppo.py:2181-2184); `perm` replays a recorded shuffle instead of drawing one."""
        if perm is not None:
            return DataLoader(self.dataset, batch_size=self.batch_size, sampler=[int(i) for i in perm])
        return DataLoader(self.dataset, batch_size=self.batch_size, shuffle=True, generator=self.loader_generator)

    def icm_train_epoch(self, perm=None, comm=None):
        """ppo.py:2487-2567; `comm` (R > 1): mpi_avg_gradients(icm_model) :2559, all-reduced totals :2565-2567."""
        loader = self._loader(perm)
        total, n = 0.0, 0
        for batch in loader:
            _, obs, next_obs, _, actions, _, _, _, _, _, _, _, _ = batch
            if len(actions.shape) < 2:
                actions = actions.unsqueeze(1)
            _, inv_loss, f_loss = self.icm(obs, next_obs, actions)
            icm_loss = (1.0 - self.icm_beta) * f_loss + self.icm_beta * inv_loss
            total += icm_loss.item()
            self.icm_optim.zero_grad()
            icm_loss.backward()
            avg_gradients(self.icm.parameters(), comm)
            self.icm_optim.step()
            n += 1
        if comm is not None and comm.size > 1:
            n, total = comm.allreduce_sum(n), comm.allreduce_sum(total)
        return total / max(n, 1)

    # ----- one epoch (ppo.py:2274-2485)
    def train_epoch(self, perm=None, comm=None):
        """`comm` (R > 1 ranks, see ThreadComm): the reference's DD-PPO exchanges -- the value normaliser sees the
        rewards-to-go of every rank's mini-batch (utils/stats.py:47-50), every parameter tensor's gradient is averaged
        over the ranks after each backward (utils/mpi_utils.py:89-111, ppo_policy.py:1035,1048), a barrier per
        mini-batch (ppo.py:2468), and the epoch's five totals are summed over the ranks (ppo.py:2471-2475)."""
        loader = self._loader(perm)
        tot = dict(actor=0.0, critic=0.0, entropy=0.0, kl=0.0, n=0)
        for batch in loader:
            critic_obs, obs, _, raw_actions, _, advantages, log_probs, rewards_tg, _, _, _, _, idxs = batch
            if self.normalize_values:
                rewards_tg = self._norm_update(rewards_tg.flatten(), comm).reshape(rewards_tg.shape)
            if obs.shape[0] == 1:
                continue
            values = self.critic(critic_obs).squeeze()
            if self.continuous:                        # ppo_policy.py:930-952: entropy := -log_prob of the MEAN (distributions.py:694)
                mean = self.actor(obs)
                cur_lp = lo.gaussian_tanh_logp(mean, self.log_std, raw_actions)
                entropy = -lo.gaussian_tanh_logp(mean, self.log_std, mean)
            else:
                probs = torch.softmax(self.actor(obs), dim=-1)
                dist = Categorical(probs)
                cur_lp = torch.unsqueeze(dist.log_prob(raw_actions.flatten()), dim=-1)
                entropy = dist.entropy()
            self.dataset.values[idxs] = values.detach()
            r = lo.ppo_minibatch_losses(cur_lp, log_probs, advantages, entropy, values, rewards_tg,
                                        self.normalize_adv, self.surr_clip, self.entropy_weight)
            if getattr(self, "trace", None) is not None:     # tests: per-mini-batch losses + raw (unclipped) gradients
                ga = torch.autograd.grad(r["actor_loss"], self.actor_params, retain_graph=True)
                gc = torch.autograd.grad(r["critic_loss"], list(self.critic.parameters()), retain_graph=True)
                self.trace.append(dict(actor=r["actor"], critic=r["critic"], kl=r["kl"], entropy=r["entropy"],
                                       adv_mean=r["adv_mean"], adv_std=r["adv_std"],
                                       actor_grad=torch.cat([x.reshape(-1) for x in ga]).numpy(),
                                       critic_grad=torch.cat([x.reshape(-1) for x in gc]).numpy()))
            self.actor_optim.zero_grad()
            r["actor_loss"].backward()
            avg_gradients(self.actor_params, comm)
            if getattr(self, "trace", None) is not None and comm is not None:
                self.trace[-1]["actor_avg_grad"] = torch.cat([p.grad.reshape(-1) for p in self.actor_params]).numpy().copy()
            nn.utils.clip_grad_norm_(self.actor_params, self.gradient_clip)
            self.actor_optim.step()
            self.critic_optim.zero_grad()
            r["critic_loss"].backward()
            avg_gradients(self.critic.parameters(), comm)
            if getattr(self, "trace", None) is not None and comm is not None:
                self.trace[-1]["critic_avg_grad"] = torch.cat([p.grad.reshape(-1) for p in self.critic.parameters()]).numpy().copy()
            nn.utils.clip_grad_norm_(self.critic.parameters(), self.gradient_clip)
            self.critic_optim.step()
            if comm is not None:
                comm.barrier()                                   # ppo.py:2468
            tot["actor"] += r["surr"]; tot["critic"] += r["critic"]
            tot["entropy"] += r["entropy"]; tot["kl"] += r["kl"]; tot["n"] += 1
        if comm is not None and comm.size > 1:                   # ppo.py:2471-2475
            for k in ("n", "entropy", "actor", "critic", "kl"):
                tot[k] = comm.allreduce_sum(tot[k])
        n = max(tot["n"], 1)
        return {"actor loss": tot["actor"] / n, "critic loss": tot["critic"] / n,
                "weighted entropy": tot["entropy"] * self.entropy_weight / n, "kl avg": tot["kl"] / n}


def train_on_rollout(cpu, epochs_per_iter, target_kl=100.0, perms=None, icm_perms=None, on_epoch=None, comm=None):
    """
    The epoch loop of PPO.learn with its KL early stop (ppo.py:2201-2232): per epoch one `_ppo_batch_train` pass, then --
    with ICM -- one `_icm_batch_train` pass (:2213-2214, BEFORE the test, so the stopping epoch still trains the ICM),
    then `if status["kl avg"] > target_kl: break` (:2222-2232; strict `>`).  `perms` / `icm_perms` replay recorded
    shuffles (one per epoch actually run).  With `comm` (R > 1 ranks) the statistics are the all-reduced ones, so every
    rank takes the same decision (the `comm.barrier()` of :2221 carries no data).  Returns the list of per-epoch
    statistics dicts (+ "icm loss").
    """
    out = []
    for epoch_idx in range(epochs_per_iter):
        kw = {} if comm is None else {"comm": comm}
        stats = cpu.train_epoch(perm=None if perms is None else perms[epoch_idx], **kw)
        if getattr(cpu, "enable_icm", False):
            stats["icm loss"] = cpu.icm_train_epoch(perm=None if icm_perms is None else icm_perms[epoch_idx], **kw)
        out.append(stats)
        if on_epoch is not None:
            on_epoch(epoch_idx, stats)
        if stats["kl avg"] > target_kl:
            break
    return out


def run_ranks(fns):
    """Run one callable per rank, each on its own thread with a ThreadComm; returns their results in rank order
    (an exception on any rank is re-raised here)."""
    import threading
    shared = ThreadComm.Shared(len(fns))
    out, err = [None] * len(fns), []

    def body(r):
        try:
            out[r] = fns[r](ThreadComm(shared, r))
        except BaseException as e:           # noqa: BLE001 -- release the other ranks, then re-raise below
            err.append(e)
            shared.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(len(fns))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if err:
        first = [e for e in err if not isinstance(e, threading.BrokenBarrierError)] or err
        raise first[0]
    return out


def ddppo_train_epoch(ranks, perms=None):
    """
    One epoch of the reference's DD-PPO over R in-process "ranks" (list of CpuPPO with identical weights, each with its
    own dataset and shuffle generator): every rank runs CpuPPO.train_epoch on its own thread with a ThreadComm, i.e. the
    exchanges of utils/stats.py:47-54, utils/mpi_utils.py:89-111 and ppo.py:2468-2475 at the points where the reference
    makes them.  Pinned by the R = 2 reference fixtures (tests/test_oracle_update_golden.py).  Returns the all-reduced
    epoch statistics (identical on every rank).
    """
    stats = run_ranks([(lambda comm, r=r, i=i: r.train_epoch(perm=None if perms is None else perms[i], comm=comm))
                       for i, r in enumerate(ranks)])
    return stats[0]


def time_iteration(E, T, obs_dim=4, n_actions=2, epochs=10, batch_size=256, seed=1234, threads=None):
    """cpu_baseline leg: one PPO iteration (rollout + GAE/build + `epochs` epochs) -> env-steps/s."""
    if threads is not None:
        torch.set_num_threads(threads)
    rng = np.random.default_rng(seed)
    obs_table = rng.standard_normal((T + 1, E, obs_dim), dtype=np.float32)
    rew_table = np.ones((T, E), dtype=np.float32)
    ppo = CpuPPO(obs_dim, n_actions, batch_size=batch_size, seed=seed)
    t0 = time.perf_counter()
    ppo.rollout(obs_table, rew_table)
    t1 = time.perf_counter()
    for _ in range(epochs):
        ppo.train_epoch()
    t2 = time.perf_counter()
    return dict(env_steps=E * T, rollout_s=t1 - t0, update_s=t2 - t1,
                env_steps_per_s=E * T / (t2 - t0), threads=torch.get_num_threads())


def pick_threads(candidates=(1, 4, 8, 16), T=128):
    """
    The reference leaves torch's intra-op threads at cores / ranks
    (utils/mpi_utils.py:37-48); on a many-core host that is far slower than a
    handful of threads for these 128-wide layers.  To keep the baseline honest
    the timed sample uses the fastest of a few settings, found on a tiny probe.
    """
    best, best_rate = None, -1.0
    for th in candidates:
        if th > (torch.get_num_threads() if best is None else 10 ** 9) and best is None:
            pass
        r = time_iteration(16, T, epochs=1, threads=th)
        if r["env_steps_per_s"] > best_rate:
            best, best_rate = th, r["env_steps_per_s"]
    return best
