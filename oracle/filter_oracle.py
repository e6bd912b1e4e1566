"""
TEST INFRASTRUCTURE (see oracle/__init__.py) -- numpy restatement of the observation / reward
filter wrappers of the reference, on the batched-env array contract of this build
(rows are agent-major columns c = a * E + e; see environments/synthetic.py).

Follows /root/reference/environments/filter_wrappers.py:
  ObservationNormalizer  :113-340   per-agent RunningMeanStd over the env batch, then
                                    (x - mean) / sqrt(var + 1e-8)                      (:155-258)
  RewardNormalizer       :342-521   discounted running reward per env; stats updated INSIDE the
                                    per-env loop with the half-updated vector (quirk Q3, :412-418);
                                    reward / sqrt(var + 1e-8); running reward zeroed where done
  GenericClipper / ObservationClipper / RewardClipper  :523-719   np.clip(x, lo, hi)
  wiring order           wrapper_utils.py:81-111: obs normaliser -> obs clipper -> reward normaliser
                                    -> reward clipper; "terminal observation" entries stay unfiltered.

PINNED by fixtures recorded from the unmodified wrapper classes: g13_filters (the stack of wrapper_utils.py:81-111
stand-alone: 2 agents, terminations, two passes; per-step observations / critic observations / rewards and the
final running statistics) and g12_c3_full (the stack inside whole PPO iterations); RunningMeanStd underneath by g4.

`gathered` (a list of per-rank inputs) reproduces the comm.allgather of the raw data inside
RunningMeanStd.update (stats.py:47-50) for the R > 1 tests.
"""
import numpy as np

from .running_stats_oracle import RunningMeanStd


class ObservationNormalizerOracle:
    """filter_wrappers.py:113-340 for one observation kind (actor or critic) of A agents."""

    def __init__(self, num_agents, obs_dim, update_stats=True, epsilon=1e-8):
        self.stats = [RunningMeanStd(shape=(obs_dim,)) for _ in range(num_agents)]
        self.update_stats = update_stats
        self.epsilon = epsilon

    def filter(self, obs, gathered=None):
        """obs [A*E, O] float32 (agent-major); gathered: list over ranks of such arrays."""
        A = len(self.stats)
        o = np.asarray(obs, dtype=np.float32).reshape(A, -1, obs.shape[-1])
        out = np.empty_like(o)
        for a in range(A):
            if self.update_stats:                                               # :218-221
                if gathered is None:
                    self.stats[a].update(o[a])
                else:
                    self.stats[a].update(None, gathered=[
                        np.asarray(g, dtype=np.float32).reshape(A, -1, obs.shape[-1])[a] for g in gathered])
            out[a] = (o[a] - self.stats[a].mean) / np.sqrt(self.stats[a].variance + self.epsilon)   # :266-268
        return out.reshape(obs.shape)


class RewardNormalizerOracle:
    """filter_wrappers.py:342-521 for A agents x E envs."""

    def __init__(self, num_agents, num_envs, update_stats=True, epsilon=1e-8, gamma=0.99):
        self.stats = [RunningMeanStd(shape=()) for _ in range(num_agents)]
        self.running_reward = [np.zeros(num_envs) for _ in range(num_agents)]    # float64, :385-387
        self.update_stats = update_stats
        self.epsilon = epsilon
        self.gamma = gamma
        self.E = num_envs

    def filter(self, reward, done):
        """reward [A*E] float32, done [A*E] bool -> normalised reward (single rank)."""
        return reward_filter_ranks([self], [reward], [done])[0]


def reward_filter_ranks(oracles, rewards, dones):
    """
    One RewardNormalizer.step on every rank of a job (oracles[r] with its reward / done arrays).
    All ranks run the same E loop iterations, and each RunningMeanStd.update inside the loop
    gathers every rank's half-updated running-reward vector (stats.py:47-50).
    """
    A, E = len(oracles[0].stats), oracles[0].E
    rs = [np.asarray(r, dtype=np.float32).reshape(A, E) for r in rewards]
    ds = [np.asarray(d, dtype=bool).reshape(A, E) for d in dones]
    outs = [np.empty_like(r) for r in rs]
    for a in range(A):
        if oracles[0].update_stats:
            for e in range(E):                                                  # :412-418 (Q3)
                for o, r in zip(oracles, rs):
                    o.running_reward[a][e] = o.running_reward[a][e] * o.gamma + r[a][e]
                vecs = [o.running_reward[a] for o in oracles]
                for o in oracles:
                    o.stats[a].update(None, gathered=vecs)
        for o, r, d, out in zip(oracles, rs, ds, outs):
            o.running_reward[a][np.where(d[a])[0]] = 0.0                        # :420-424
            x = r[a].copy()
            x /= np.sqrt(o.stats[a].variance + o.epsilon)                       # :455-458
            out[a] = x
    return [out.reshape(np.shape(r)) for out, r in zip(outs, rewards)]


def clip(x, clip_range):
    """GenericClipper._clip, filter_wrappers.py:583-597."""
    return np.clip(x, clip_range[0], clip_range[1])


class FilteredEnvOracle:
    """
    wrapper_utils.py:81-111 over the batched-env contract
        obs, critic_obs, reward, terminated, truncated, terminal_obs = env.step(action)
    (terminal_obs is the "terminal observation" info entry: never filtered).
    """

    def __init__(self, num_agents, num_envs, obs_dim, critic_obs_dim, normalize_obs=True,
                 normalize_rewards=True, obs_clip=None, reward_clip=None, gamma=0.99,
                 update_stats=True):
        self.obs_norm = self.cobs_norm = self.rew_norm = None
        if normalize_obs:
            self.obs_norm = ObservationNormalizerOracle(num_agents, obs_dim, update_stats)
            self.cobs_norm = ObservationNormalizerOracle(num_agents, critic_obs_dim, update_stats)
        if normalize_rewards:
            self.rew_norm = RewardNormalizerOracle(num_agents, num_envs, update_stats, gamma=gamma)
        self.obs_clip = obs_clip
        self.reward_clip = reward_clip

    def filter_obs(self, obs, critic_obs):
        """ObservationFilter.reset/step: local then critic (:60-66)."""
        obs = np.asarray(obs, dtype=np.float32)
        critic_obs = np.asarray(critic_obs, dtype=np.float32)
        if self.obs_norm is not None:
            obs = self.obs_norm.filter(obs)
            critic_obs = self.cobs_norm.filter(critic_obs)
        if self.obs_clip is not None:
            obs, critic_obs = clip(obs, self.obs_clip), clip(critic_obs, self.obs_clip)
        return obs, critic_obs

    def filter_step(self, obs, critic_obs, reward, terminated, truncated):
        obs, critic_obs = self.filter_obs(obs, critic_obs)
        reward = np.asarray(reward, dtype=np.float32)
        if self.rew_norm is not None:
            reward = self.rew_norm.filter(reward, np.logical_or(terminated, truncated))
        if self.reward_clip is not None:
            reward = clip(reward, self.reward_clip)
        return obs, critic_obs, reward
