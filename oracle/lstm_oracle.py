"""
TEST INFRASTRUCTURE (see oracle/__init__.py) -- torch-CPU restatement of the reference's LSTM policy
path, with the reference's list-based loop structure, on torch primitives (nn.LSTM, LayerNorm, Categorical, Adam).
PINNED by fixtures g12_lstm_term / g12_lstm_cut recorded from the unmodified reference's PPO object with
LSTMNetwork actor / critic: logged hidden states, windows, masks, epochs with hand-over + write-back, final weights
(tests/test_oracle_update_golden.py).

  LSTMNet                 <- LSTMNetwork.forward            networks/ppo_networks/lstm.py:13-127,
                             PPOLSTMNetwork                 networks/ppo_networks/base.py:136-185
  CpuLSTMPPO.rollout      <- PPO.rollout                    ppo.py:1646-1983 with
                             add_episode_info hidden states policies/ppo_policy.py:593-651
  SequenceDataset         <- PPODataset.build / __getitem__ utils/episode_info.py:745-987
                             (terminal_sequence_masks :775-809, windows :954-987)
  CpuLSTMPPO.train_epoch  <- PPO._ppo_batch_train           ppo.py:2274-2485 incl. the hidden-state
                             hand-over :2312-2319 and write-back :2450-2466
"""
import numpy as np
import torch
import torch.nn as nn
from torch.distributions import Categorical
from torch.utils.data import DataLoader, Dataset

from . import ppo_loss_oracle as lo
from .cpu_ppo_loop import CpuPPO, _Episode, make_mlp


class LSTMNet(nn.Module):
    """lstm.py:13-127 (stateful: hidden_state persists between calls, reset on a batch-size change)."""

    def __init__(self, in_size, out_size, out_gain, lstm_hidden=128, layers=1, ff_hidden=128, ff_depth=1):
        super().__init__()
        self.layers, self.lstm_hidden = layers, lstm_hidden
        self.lstm = nn.LSTM(in_size, lstm_hidden, layers)
        self.layer_norm = nn.LayerNorm(lstm_hidden)
        self.ff_layers = nn.Module()
        self.ff_layers.sequential_net = make_mlp(lstm_hidden, out_size, ff_hidden, ff_depth, out_gain=out_gain)
        self.hidden_state = None

    def zero_state(self, batch):
        return (torch.zeros(self.layers, batch, self.lstm_hidden), torch.zeros(self.layers, batch, self.lstm_hidden))

    def reset_hidden_state(self, batch):
        self.hidden_state = self.zero_state(batch)

    def forward(self, x):
        out = x.unsqueeze(0) if x.dim() == 2 else torch.transpose(x, 0, 1)        # :103-107
        if self.hidden_state is None or self.hidden_state[0].shape[1] != out.shape[1]:
            self.reset_hidden_state(out.shape[1])                                 # :109-113
        _, self.hidden_state = self.lstm(out, self.hidden_state)
        out = torch.relu(self.layer_norm(self.hidden_state[0][-1]))               # :115-121
        return self.ff_layers.sequential_net(out)


class _SeqEpisode(_Episode):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.actor_hidden, self.actor_cell, self.critic_hidden, self.critic_cell = [], [], [], []
        self.terminal = False


class SequenceDataset(Dataset):
    """PPODataset with sequence_length S (episode_info.py:745-987)."""

    def __init__(self, episodes, S):
        cat = lambda name: [x for ep in episodes for x in getattr(ep, name)]
        self.S = S
        self.observations = torch.tensor(np.array(cat("observations")), dtype=torch.float32)
        self.critic_observations = torch.tensor(np.array(cat("critic_observations")), dtype=torch.float32)
        self.raw_actions = torch.tensor(np.array(cat("raw_actions")), dtype=torch.long)
        self.rewards_to_go = torch.tensor(np.array([x for ep in episodes for x in ep.rewards_to_go]), dtype=torch.float32)
        self.advantages = torch.tensor(np.array([x for ep in episodes for x in ep.advantages]), dtype=torch.float32)
        self.log_probs = torch.stack(cat("log_probs")).to(torch.float32)
        self.values = torch.tensor(np.concatenate([ep.values for ep in episodes]), dtype=torch.float32)
        hid = lambda name: torch.transpose(torch.tensor(np.concatenate(cat(name), axis=1), dtype=torch.float32), 0, 1)
        self.actor_hidden, self.actor_cell = hid("actor_hidden"), hid("actor_cell")          # [N, layers, H], :841-857
        self.critic_hidden, self.critic_cell = hid("critic_hidden"), hid("critic_cell")
        self.episodes = episodes
        N = len(self.observations)
        term = np.zeros(N, dtype=bool)                                                      # :775-789
        cur = 0
        for ep in episodes:
            cur += ep.length
            if ep.terminal:
                term[cur - 1] = True
        self.masks = []
        for ts in range(N - (S - 1)):                                                       # :791-809
            mask = np.zeros(S, dtype=bool)
            k = 0
            for m in range(ts, ts + S):
                if term[m]:
                    mask[k + 1:] = True
                    break
                k += 1
            self.masks.append(mask)

    def __len__(self):
        return len(self.observations) - (self.S - 1)

    def __getitem__(self, idx):
        if self.S == 1:
            i = idx
            return (self.critic_observations[i], self.observations[i], self.raw_actions[i], self.advantages[i],
                    self.log_probs[i], self.rewards_to_go[i], self.actor_hidden[i], self.critic_hidden[i],
                    self.actor_cell[i], self.critic_cell[i], i)
        idx += self.S - 1                                                                   # :960-962
        start, stop = idx - (self.S - 1), idx + 1
        glob = self.critic_observations[start:stop].clone()
        obs = self.observations[start:stop].clone()
        obs[torch.as_tensor(self.masks[start])] = 0.0                                       # :976-978
        return (glob, obs, self.raw_actions[idx], self.advantages[idx], self.log_probs[idx], self.rewards_to_go[idx],
                self.actor_hidden[idx], self.critic_hidden[idx], self.actor_cell[idx], self.critic_cell[idx], idx)


class CpuLSTMPPO(CpuPPO):
    """One rank of the reference's PPO with LSTM actor / critic (Discrete actions, table-driven env)."""

    def __init__(self, obs_dim, n_actions, sequence_length=10, lstm_hidden=128, layers=1, ff_hidden=128, ff_depth=1,
                 lr=3e-4, **kw):
        super().__init__(obs_dim, n_actions, lr=lr, **kw)
        self.S = sequence_length
        self.actor = LSTMNet(obs_dim, n_actions, 0.01, lstm_hidden, layers, ff_hidden, ff_depth)
        self.critic = LSTMNet(obs_dim, 1, 1.0, lstm_hidden, layers, ff_hidden, ff_depth)
        self.actor_optim = torch.optim.Adam(self.actor.parameters(), lr=lr, eps=1e-5)
        self.critic_optim = torch.optim.Adam(self.critic.parameters(), lr=lr, eps=1e-5)

    def rollout(self, obs_table, reward_table, actions, term_table=None, max_ts_per_ep=None):
        T, E = reward_table.shape
        self.actor.reset_hidden_state(1)                       # ppo_policy.py:512-519
        self.critic.reset_hidden_state(1)
        new_ep = lambda: _SeqEpisode(self.gamma, self.lambd, self.clip)
        episodes = [new_ep() for _ in range(E)]
        finished = []
        ep_ts = np.zeros(E, dtype=np.int64)
        for t in range(T):
            ep_ts += 1
            obs = obs_table[t]
            t_obs = torch.tensor(obs, dtype=torch.float32)
            with torch.no_grad():
                probs = torch.softmax(self.actor(t_obs), dim=-1)
            dist = Categorical(probs)
            a = torch.as_tensor(actions[t], dtype=torch.long)
            log_prob = torch.unsqueeze(dist.log_prob(a), dim=-1)
            a_np = a.unsqueeze(-1).numpy()
            value = self.values_of(t_obs).unsqueeze(-1)
            nxt = obs_table[t + 1]
            rew = reward_table[t].reshape(E, 1).astype(np.float64)
            where_term = np.where(term_table[t])[0] if term_table is not None else np.array([], dtype=np.int64)
            where_not_term = np.setdiff1d(np.arange(E), where_term)
            # ppo_policy.py:598-627: the states AFTER this step, zeroed for the terminated envs
            states = [x.clone() for x in (*self.actor.hidden_state, *self.critic.hidden_state)]
            for s in states:
                s[:, where_term, :] = 0.0
            ah, ac, ch, cc = states
            for e in range(E):
                episodes[e].add_info(critic_observation=obs[e], observation=obs[e], next_observation=nxt[e],
                                     raw_action=a_np[e], action=a_np[e], value=value[e].item(),
                                     log_prob=log_prob[e], reward=rew[e].item())
                episodes[e].actor_hidden.append(ah[:, [e], :].numpy()); episodes[e].actor_cell.append(ac[:, [e], :].numpy())
                episodes[e].critic_hidden.append(ch[:, [e], :].numpy()); episodes[e].critic_cell.append(cc[:, [e], :].numpy())
            for e in where_term:
                episodes[e].end_episode(0.0, 0.0, self.rtg_accum)
                episodes[e].terminal = True
                finished.append(episodes[e])
                episodes[e] = new_ep()
            ep_ts[where_term] = 0                              # ppo.py:1851, before the check below
            ep_max_reached = max_ts_per_ep is not None and (ep_ts == max_ts_per_ep).any() and where_not_term.size > 0
            last = t == T - 1
            if ep_max_reached or last:                         # ppo.py:1863-1881
                where_maxed = np.arange(E) if last else np.where(ep_ts >= max_ts_per_ep)[0]
                where_maxed = np.setdiff1d(where_maxed, where_term)
                next_value = self.values_of(torch.tensor(nxt, dtype=torch.float32))   # steps the critic's LSTM
                for e in where_maxed:
                    episodes[e].end_episode(next_value[e].item(), next_value[e].item(), self.rtg_accum)
                    finished.append(episodes[e])
                    episodes[e] = new_ep()
                    ep_ts[e] = 0
        self.dataset = SequenceDataset(finished, self.S)
        return self.dataset

    def train_epoch(self, perm=None):
        loader = self._loader(perm)
        tot = dict(actor=0.0, critic=0.0, entropy=0.0, kl=0.0, n=0)
        ds = self.dataset
        for batch in loader:
            critic_obs, obs, raw_actions, advantages, log_probs, rewards_tg, a_h, c_h, a_c, c_c, idxs = batch
            if self.normalize_values:
                rewards_tg = self._norm_update(rewards_tg.flatten()).reshape(rewards_tg.shape)
            if obs.shape[0] == 1:
                continue
            self.actor.hidden_state = (torch.transpose(a_h, 0, 1).contiguous(), torch.transpose(a_c, 0, 1).contiguous())
            self.critic.hidden_state = (torch.transpose(c_h, 0, 1).contiguous(), torch.transpose(c_c, 0, 1).contiguous())
            values = self.critic(critic_obs).squeeze()
            probs = torch.softmax(self.actor(obs), dim=-1)
            dist = Categorical(probs)
            cur_lp = torch.unsqueeze(dist.log_prob(raw_actions.flatten()), dim=-1)
            entropy = dist.entropy()
            ds.values[idxs] = values.detach()
            r = lo.ppo_minibatch_losses(cur_lp, log_probs, advantages, entropy, values, rewards_tg,
                                        self.normalize_adv, self.surr_clip, self.entropy_weight)
            self.actor_optim.zero_grad()
            r["actor_loss"].backward(retain_graph=True)
            nn.utils.clip_grad_norm_(self.actor.parameters(), self.gradient_clip)
            self.actor_optim.step()
            self.critic_optim.zero_grad()
            r["critic_loss"].backward(retain_graph=True)
            nn.utils.clip_grad_norm_(self.critic.parameters(), self.gradient_clip)
            self.critic_optim.step()
            ds.actor_hidden[idxs] = torch.transpose(self.actor.hidden_state[0].detach().clone(), 0, 1)      # :2450-2466
            ds.critic_hidden[idxs] = torch.transpose(self.critic.hidden_state[0].detach().clone(), 0, 1)
            ds.actor_cell[idxs] = torch.transpose(self.actor.hidden_state[1].detach().clone(), 0, 1)
            ds.critic_cell[idxs] = torch.transpose(self.critic.hidden_state[1].detach().clone(), 0, 1)
            tot["actor"] += r["surr"]; tot["critic"] += r["critic"]
            tot["entropy"] += r["entropy"]; tot["kl"] += r["kl"]; tot["n"] += 1
        n = max(tot["n"], 1)
        return {"actor loss": tot["actor"] / n, "critic loss": tot["critic"] / n,
                "weighted entropy": tot["entropy"] * self.entropy_weight / n, "kl avg": tot["kl"] / n}
