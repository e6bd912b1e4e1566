"""
TEST INFRASTRUCTURE (see oracle/__init__.py) -- torch-CPU restatement of the reference's
multi-agent transformer path.

  SelfAttention / Encoding / Decoding blocks <- networks/attention.py:13-257        PINNED (golden g5)
  MATActor / MATCritic / MATActorCritic      <- networks/actor_critic/multi_agent_transformer.py:22-373
  evaluate_actions / tokened action block    <- policies/mat_policy.py:308-439
  shared-episode dataset + update            <- utils/episode_info.py:485-644,990-1084 (layout PINNED: g3),
                                                ppo.py:2274-2485, mat_policy.py:677-699

The policy loop is PINNED by fixture g12_c5_mat (the unmodified reference's PPO object with MATPolicy at the C5
shapes: rollout log-probs / values, shared dataset incl. quirk Q14, first-mini-batch losses + gradients, epochs,
final weights; tests/test_oracle_update_golden.py).  The ICM-for-MAT branches stay restated from text.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.distributions import Categorical
from torch.utils.data import DataLoader, Dataset

from . import episode_info_oracle as eo
from . import ppo_loss_oracle as lo
from .running_stats_oracle import RunningMeanStd

RELU_GAIN = nn.init.calculate_gain('relu')


def _lin(i, o, gain=np.sqrt(2), bias=True):
    layer = nn.Linear(i, o, bias=bias)
    nn.init.orthogonal_(layer.weight, gain)
    if bias:
        nn.init.constant_(layer.bias, 0.0)
    return layer


class SelfAttention(nn.Module):
    def __init__(self, D, H, num_agents, internal_init=0.01, out_init=0.01, masked=False):
        super().__init__()
        self.masked, self.num_heads = masked, H
        self.key_net = _lin(D, D, internal_init); self.query_net = _lin(D, D, internal_init)
        self.value_net = _lin(D, D, internal_init); self.proj = _lin(D, D, out_init)
        self.register_buffer("mask", torch.tril(torch.ones(num_agents + 1, num_agents + 1)).view(
            1, 1, num_agents + 1, num_agents + 1))

    def forward(self, key, value, query):
        B, L, D = query.size()
        H = self.num_heads
        k = self.key_net(key).view(B, L, H, D // H).transpose(1, 2)
        q = self.query_net(query).view(B, L, H, D // H).transpose(1, 2)
        v = self.value_net(value).view(B, L, H, D // H).transpose(1, 2)
        att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(k.size(-1)))
        if self.masked:
            att = att.masked_fill(self.mask[:, :, :L, :L] == 0, float('-inf'))
        y = F.softmax(att, dim=-1) @ v
        return self.proj(y.transpose(1, 2).contiguous().view(B, L, D))


class EncodingBlock(nn.Module):
    def __init__(self, D, H, A):
        super().__init__()
        self.ln1, self.ln2 = nn.LayerNorm(D), nn.LayerNorm(D)
        self.attn = SelfAttention(D, H, A, masked=False)
        self.mlp = nn.Sequential(_lin(D, D, RELU_GAIN), nn.GELU(), _lin(D, D, 0.01))

    def forward(self, x):
        x = self.ln1(x + self.attn(x, x, x))
        return self.ln2(x + self.mlp(x))


class DecodingBlock(nn.Module):
    def __init__(self, D, H, A):
        super().__init__()
        self.ln1, self.ln2, self.ln3 = nn.LayerNorm(D), nn.LayerNorm(D), nn.LayerNorm(D)
        self.attn1 = SelfAttention(D, H, A, masked=True)
        self.attn2 = SelfAttention(D, H, A, masked=True)
        self.mlp = nn.Sequential(_lin(D, D, RELU_GAIN), nn.GELU(), _lin(D, D, 0.01))

    def forward(self, x, rep_enc):
        x = self.ln1(x + self.attn1(x, x, x))
        x = self.ln2(rep_enc + self.attn2(key=x, value=x, query=rep_enc))
        return self.ln3(x + self.mlp(x))


class MATActor(nn.Module):
    def __init__(self, n_actions, A, D=64, blocks=1, H=1):
        super().__init__()
        self.action_encoder = nn.Sequential(_lin(n_actions + 1, D, RELU_GAIN, bias=False), nn.GELU())
        self.ln = nn.LayerNorm(D)
        self.blocks = nn.Sequential(*[DecodingBlock(D, H, A) for _ in range(blocks)])
        self.head = nn.Sequential(_lin(D, D, RELU_GAIN), nn.GELU(), nn.LayerNorm(D), _lin(D, n_actions, 0.01))

    def forward(self, actions, encoded_obs):
        x = self.ln(self.action_encoder(actions))
        for b in self.blocks:
            x = b(x, encoded_obs)
        return F.softmax(self.head(x), dim=-1)            # output_func (distributions.py:1043-1045)


class MATCritic(nn.Module):
    def __init__(self, obs_dim, A, D=64, blocks=1, H=1):
        super().__init__()
        self.obs_encoder = nn.Sequential(nn.LayerNorm(obs_dim), _lin(obs_dim, D, RELU_GAIN), nn.GELU())
        self.ln = nn.LayerNorm(D)
        self.blocks = nn.Sequential(*[EncodingBlock(D, H, A) for _ in range(blocks)])
        self.head = nn.Sequential(_lin(D, D, RELU_GAIN), nn.GELU(), nn.LayerNorm(D), _lin(D, 1, 0.01))

    def forward(self, obs):
        enc = self.blocks(self.ln(self.obs_encoder(obs)))
        return enc, self.head(enc)


class MATActorCritic(nn.Module):
    def __init__(self, obs_dim, n_actions, A, D=64, blocks=1, H=1):
        super().__init__()
        self.actor = MATActor(n_actions, A, D, blocks, H)
        self.critic = MATCritic(obs_dim, A, D, blocks, H)

    def forward(self, obs, action_block):
        enc, values = self.critic(obs)
        return values, self.actor(action_block, enc)


class _SharedDataset(Dataset):
    """PPOSharedEpisodeDataset items: [A, .] per index (episode_info.py:1058-1084)."""

    def __init__(self, obs, actions, adv, logp, rtg, values, next_obs=None):
        self.obs, self.actions, self.adv, self.logp, self.rtg, self.values = obs, actions, adv, logp, rtg, values
        self.next_obs = next_obs

    def __len__(self):
        return self.obs.shape[0]

    def __getitem__(self, i):
        return self.obs[i], self.actions[i], self.adv[i], self.logp[i], self.rtg[i], i


class CpuMATPPO:
    """One rank of the reference's MAT training on CPU (discrete actions, critic view 'local')."""

    def __init__(self, obs_dim, n_actions, A, lr=3e-4, gamma=0.99, lambd=0.95, bootstrap_clip=(-100.0, 100.0),
                 surr_clip=0.2, entropy_weight=0.01, gradient_clip=0.5, batch_size=256, seed=0,
                 enable_icm=False, agent_shared_icm=False, icm_lr=3e-4, icm_beta=0.8, intr_reward_weight=1.0):
        torch.manual_seed(seed)
        self.A, self.n_actions = A, n_actions
        self.enable_icm, self.agent_shared_icm = enable_icm, agent_shared_icm
        self.icm_beta, self.intr_reward_weight = icm_beta, intr_reward_weight
        self.intrinsic_score_avg = 0.0
        if enable_icm:                                    # mat_policy.py:132-176, 224-227
            from .icm_oracle import ICM
            if agent_shared_icm:                          # spaces spanning the group: obs A*O, MultiDiscrete([n] * A)
                self.icm = ICM(obs_dim * A, n_actions * A, discrete=True, nvec=[n_actions] * A)
            else:
                self.icm = ICM(obs_dim, n_actions, discrete=True)
            self.icm_optim = torch.optim.Adam(self.icm.parameters(), lr=icm_lr, eps=1e-5)
        self.ac = MATActorCritic(obs_dim, n_actions, A)
        self.optim = torch.optim.Adam(self.ac.parameters(), lr=lr, eps=1e-5)       # mat_policy.py:221-222
        self.gamma, self.lambd, self.clip = gamma, lambd, bootstrap_clip
        self.surr_clip, self.entropy_weight, self.gradient_clip = surr_clip, entropy_weight, gradient_clip
        self.batch_size = batch_size
        self.value_stats = RunningMeanStd()
        self.loader_generator = torch.Generator().manual_seed(seed)

    def _denorm(self, v):
        mean = torch.tensor(self.value_stats.mean, dtype=torch.float32)
        var = torch.tensor(self.value_stats.variance, dtype=torch.float32)
        return mean + v * torch.sqrt(var + torch.tensor([1e-8]))

    def action_block(self, actions):
        """mat_policy.py:308-344 + 399-403: start token, one-hot of the previous agents' actions."""
        B = actions.shape[0]
        blk = torch.zeros(B, self.A, self.n_actions + 1)
        blk[:, 0, 0] = 1
        blk[:, 1:, 1:] = F.one_hot(actions, num_classes=self.n_actions)[:, :-1, :]
        return blk

    def evaluate(self, obs, actions):
        """mat_policy.py:378-439 -> values [B,A,1], log_probs [B,A,1], entropy [B,A,1]."""
        B = obs.shape[0]
        values, probs = self.ac(obs, self.action_block(actions))
        dist = Categorical(probs.reshape(-1, self.n_actions))
        logp = dist.log_prob(actions.reshape(-1)).reshape(B, self.A, 1)
        ent = dist.entropy().reshape(B, self.A, 1)
        return values, logp, ent

    def intrinsic_rewards(self, obs, nxt, actions, slot_order):
        """
        ppo.py:1219-1288 for one step: obs / nxt [E,A,O], actions [E,A] in the policy's slot order ->
        float32 [E,A].  Shared form (mat_policy.py:1012-1090): the rows are built in the ORIGINAL agent
        order (icm_agent_ids) -- original agent k sits in slot argsort(slot_order)[k] -- and every agent of
        an env receives the env's one value.
        """
        E, A = actions.shape
        o, n, a = (torch.tensor(x) for x in (obs, nxt, actions))
        with torch.no_grad():
            if self.agent_shared_icm:
                inv = torch.as_tensor(np.argsort(slot_order))
                ir, _, _ = self.icm(o[:, inv].reshape(E, -1).float(), n[:, inv].reshape(E, -1).float(), a[:, inv].long())
                ir = ir.reshape(E, 1).repeat(1, A)
            else:
                ir, _, _ = self.icm(o.reshape(E * A, -1).float(), n.reshape(E * A, -1).float(), a.reshape(E * A, 1).long())
                ir = ir.reshape(E, A)
        return ir.numpy() * np.float32(self.intr_reward_weight)

    def rollout(self, obs_table, reward_table, actions, slot_order=None, dataset_slot_of=None):
        """
        dataset_slot_of (quirk Q14, pinned by fixture g12_c5_mat): PPO.rollout creates the dataset BEFORE it reshuffles
        the policy's agents (ppo.py:1546-1547 vs 1643-1644; PPOSharedEpisodeDataset keeps the agent_ids array it was
        given, episode_info.py:1012), so the dataset's agent axis is in the PREVIOUS rollout's slot order while the
        networks saw this rollout's.  dataset_slot_of[j] = the rollout slot whose agent sits in dataset slot j.

        obs_table [T+1,E,A,O], reward_table [T,E,A], actions [T,E,A] (recorded).  Fixed-length: every env's
        shared episode closes at the last step with the critic bootstrap; dataset rows are env-major
        (episode_info.py:584-637), each row [A, .].  slot_order[j] = original index of the agent in slot j
        (needed by the agent-shared ICM only).
        """
        T, E, A = reward_table.shape
        intr = None
        if self.enable_icm:
            slot_order = np.arange(A) if slot_order is None else np.asarray(slot_order)
            intr = np.stack([self.intrinsic_rewards(obs_table[t], obs_table[t + 1], actions[t], slot_order)
                             for t in range(T)])                              # [T,E,A]
            reward_table = reward_table.astype(np.float64) + intr             # ppo.py:1283 (float64 + float32)
        with torch.no_grad():
            vals, logps = [], []
            for t in range(T):
                o = torch.tensor(obs_table[t], dtype=torch.float32)
                a = torch.tensor(actions[t], dtype=torch.long)
                v, lp, _ = self.evaluate(o, a)                  # teacher-forced == autoregressive log-probs
                vals.append(self._denorm(v.squeeze(-1)).numpy()); logps.append(lp.squeeze(-1).numpy())
            _, nv = self.ac.critic(torch.tensor(obs_table[T], dtype=torch.float32))
            next_value = self._denorm(nv.squeeze(-1)).numpy()
        vals, logps = np.stack(vals), np.stack(logps)           # [T,E,A]
        adv = np.zeros((E * T, A), dtype=np.float32); rtg = np.zeros((E * T, A), dtype=np.float32)
        for e in range(E):
            for a in range(A):
                nv = nr = float(next_value[e, a])
                if self.enable_icm:                      # ppo.py:1926-1930 "surprise" (per env: quirk Q2 fixed)
                    # quirk Q12: the in-place `+=` on the numpy view lands in next_value too (see cpu_ppo_loop.py)
                    nv = nr = float(np.float32(nr) + (intr[T - 1, e, a] - np.float32(self.intrinsic_score_avg)))
                ad, rg = eo.end_episode(reward_table[:, e, a], vals[:, e, a], nv,
                                        nr, self.gamma, self.lambd, self.clip, True)
                adv[e * T:(e + 1) * T, a] = ad; rtg[e * T:(e + 1) * T, a] = rg
        flat = lambda x: np.concatenate([x[:, e] for e in range(E)], axis=0)
        if dataset_slot_of is not None:
            k = np.asarray(dataset_slot_of)
            obs_table, actions, vals, logps = obs_table[:, :, k], actions[:, :, k], vals[:, :, k], logps[:, :, k]
            adv, rtg = adv[:, k], rtg[:, k]
        self.dataset = _SharedDataset(torch.tensor(flat(obs_table[:-1])), torch.tensor(flat(actions)).long(),
                                      torch.tensor(adv), torch.tensor(flat(logps)), torch.tensor(rtg),
                                      torch.tensor(flat(vals)),
                                      torch.tensor(flat(obs_table[1:])) if self.enable_icm else None)
        if self.enable_icm:
            # ppo.py:1940-1963, 2074-2080 with no terminations: every env contributes exactly one (maxed) episode,
            # total_episodes / env_batch_size = 1, and the per-policy score sums the agents' rewards
            self.intrinsic_score_avg = float(intr.sum())
        return self.dataset

    def icm_train_epoch(self, agent_idxs=None):
        """
        ppo.py:2487-2567 on the shared dataset.  agent_idxs = the policy's (in-place shuffled) index vector,
        applied as `x[:, agent_idxs]` before the agents are laid side by side (case 2); without agent sharing
        every (row, agent) pair is one sample (case 3).
        """
        ds = self.dataset
        loader = DataLoader(ds, batch_size=self.batch_size, shuffle=True, generator=self.loader_generator)
        total, n = 0.0, 0
        for obs, act, _, _, _, idx in loader:
            obs, nxt = obs.float(), ds.next_obs[idx].float()
            B = obs.shape[0]
            if self.agent_shared_icm:
                ai = torch.as_tensor(np.asarray(agent_idxs))
                obs, nxt = obs[:, ai].reshape(B, -1), nxt[:, ai].reshape(B, -1)
                act = act.reshape(B, self.A, 1)[:, ai].reshape(B, -1)
            else:
                obs, nxt, act = obs.reshape(B * self.A, -1), nxt.reshape(B * self.A, -1), act.reshape(B * self.A, -1)
            _, inv_loss, f_loss = self.icm(obs, nxt, act)
            icm_loss = (1.0 - self.icm_beta) * f_loss + self.icm_beta * inv_loss
            total += icm_loss.item()
            self.icm_optim.zero_grad()
            icm_loss.backward()
            self.icm_optim.step()
            n += 1
        return total / max(n, 1)

    def train_epoch(self, perm=None):
        """ppo.py:2274-2485 with MATPolicy.evaluate / update_weights (one optimiser, summed loss).
        perm replays a recorded shuffle; self.trace (a list) collects per-mini-batch losses and raw gradients."""
        if perm is not None:
            loader = DataLoader(self.dataset, batch_size=self.batch_size, sampler=[int(i) for i in perm])
        else:
            loader = DataLoader(self.dataset, batch_size=self.batch_size, shuffle=True, generator=self.loader_generator)
        tot = dict(actor=0.0, critic=0.0, entropy=0.0, kl=0.0, n=0)
        for obs, actions, adv, logp_old, rtg, idxs in loader:
            shape = rtg.shape
            self.value_stats.update(rtg.flatten().numpy())
            mean = torch.tensor(self.value_stats.mean, dtype=torch.float32)
            var = torch.tensor(self.value_stats.variance, dtype=torch.float32)
            rtg = ((rtg.flatten() - mean) / torch.sqrt(var + torch.tensor([1e-8]))).reshape(shape)
            if obs.shape[0] == 1:
                continue
            values, cur_lp, entropy = self.evaluate(obs, actions)
            self.dataset.values[idxs] = values.squeeze(-1).detach()
            r = lo.ppo_minibatch_losses(cur_lp, logp_old, adv, entropy, values, rtg, True, self.surr_clip,
                                        self.entropy_weight, use_huber=True)      # MATPolicy: use_huber_loss=True
            if getattr(self, "trace", None) is not None:
                ga = torch.autograd.grad(r["actor_loss"], list(self.ac.actor.parameters()), retain_graph=True)
                gc = torch.autograd.grad(r["critic_loss"], list(self.ac.critic.parameters()), retain_graph=True)
                self.trace.append(dict(actor=r["actor"], critic=r["critic"],
                                       actor_grad=torch.cat([x.reshape(-1) for x in ga]).numpy(),
                                       critic_grad=torch.cat([x.reshape(-1) for x in gc]).numpy()))
            self.optim.zero_grad()
            (r["actor_loss"] + r["critic_loss"]).backward()                       # mat_policy.py:677-699
            nn.utils.clip_grad_norm_(self.ac.parameters(), self.gradient_clip)
            self.optim.step()
            tot["actor"] += r["surr"]; tot["critic"] += r["critic"]
            tot["entropy"] += r["entropy"]; tot["kl"] += r["kl"]; tot["n"] += 1
        n = max(tot["n"], 1)
        return {"actor loss": tot["actor"] / n, "critic loss": tot["critic"] / n,
                "weighted entropy": tot["entropy"] * self.entropy_weight / n, "kl avg": tot["kl"] / n}
