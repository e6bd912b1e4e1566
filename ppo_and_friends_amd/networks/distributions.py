"""
Action distributions backed by the HIP kernels of csrc/distributions.hip.

Same class / method names as the reference's networks/distributions.py
(CategoricalDistribution :199-269, GaussianDistribution :441-694), but instead
of moving the actor output to the CPU and building torch.distributions objects
(policies/ppo_policy.py:770,930) the "distribution" here is a light handle on
the device logits / means, and sampling, log-probs and entropy (forward and
backward) are single fused launches.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import kernels as K
from ..spaces import get_space_dtype_str


class _CategoricalEval(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, actions):
        logits = logits.contiguous()
        logp, ent, probs = K.categorical_eval_fwd(logits, actions)
        ctx.save_for_backward(probs, actions)
        return logp, ent

    @staticmethod
    def backward(ctx, g_logp, g_ent):
        probs, actions = ctx.saved_tensors
        g_logp = None if g_logp is None else g_logp.contiguous()
        g_ent = None if g_ent is None else g_ent.contiguous()
        return K.categorical_eval_bwd(probs, actions, g_logp, g_ent), None


class _GaussianTanhEval(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, log_std, x, min_std):
        mean = mean.contiguous()
        logp, ent = K.gaussian_tanh_eval_fwd(mean, log_std, x, min_std)
        ctx.save_for_backward(mean, log_std, x)
        ctx.min_std = min_std
        return logp, ent

    @staticmethod
    def backward(ctx, g_logp, g_ent):
        mean, log_std, x = ctx.saved_tensors
        g_logp = None if g_logp is None else g_logp.contiguous()
        g_ent = None if g_ent is None else g_ent.contiguous()
        d_mean, d_log_std = K.gaussian_tanh_eval_bwd(mean, log_std, x, g_logp, g_ent, ctx.min_std)
        return d_mean, d_log_std, None, None


class _PhiloxStream:
    """Counter-based RNG bookkeeping: (seed, running offset) instead of generator state."""

    def __init__(self, seed=0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.offset = 0

    def take(self, n):
        o = self.offset
        self.offset += int(n)
        return self.seed, o


class CategoricalDistribution:
    """networks/distributions.py:199-269 on device logits (the softmax output_func :1043-1045 is fused in)."""

    def __init__(self, seed=0, **kw_args):
        self.rng = _PhiloxStream(seed)

    def get_distribution(self, logits):
        return logits

    def sample_distribution(self, logits):
        """-> (action, raw_action, log_prob), each [n,1] (reference returns the sample twice, :66-90)."""
        seed, off = self.rng.take(logits.shape[0])
        a, lp = K.categorical_sample(logits.contiguous(), seed, off)
        a = a.unsqueeze(-1)
        return a, a, lp.unsqueeze(-1)

    def get_log_probs_and_entropy(self, logits, actions):
        logp, ent = _CategoricalEval.apply(logits, actions.flatten().contiguous())
        return logp.unsqueeze(-1), ent

    def refine_prediction(self, logits):
        return torch.argmax(logits, dim=-1)          # :262-263 (argmax of probs == argmax of logits)


class GaussianDistribution(nn.Module):
    """networks/distributions.py:441-694: learned log_std (init -std_offset), softplus, min_std floor, tanh squash."""

    def __init__(self, act_dim, std_offset=0.5, min_std=0.01, distribution_min=-1.0,
                 distribution_max=1.0, seed=0, **kw_args):
        super().__init__()
        self.min_std = float(min_std)
        # bounds per action dimension, as the reference keeps them (:476-483); scalars broadcast
        self.dist_min = np.broadcast_to(np.asarray(distribution_min, dtype=np.float32).reshape(-1), (act_dim,)).copy() \
            if np.size(distribution_min) in (1, act_dim) else np.asarray(distribution_min, dtype=np.float32).reshape(-1)
        self.dist_max = np.broadcast_to(np.asarray(distribution_max, dtype=np.float32).reshape(-1), (act_dim,)).copy() \
            if np.size(distribution_max) in (1, act_dim) else np.asarray(distribution_max, dtype=np.float32).reshape(-1)
        if self.dist_min.size != act_dim or self.dist_max.size != act_dim:
            raise ValueError(f"distribution_min / distribution_max must be scalars or hold {act_dim} values")
        # :606: the rescale runs for every dimension as soon as ANY bound differs from the unit box
        self.rescale = bool((self.dist_min != -1.0).any() or (self.dist_max != 1.0).any())
        # device copies for the kernels; not part of the state_dict (the reference's holds log_std only)
        self.register_buffer("dist_min_t", torch.from_numpy(self.dist_min.copy()), persistent=False)
        self.register_buffer("dist_max_t", torch.from_numpy(self.dist_max.copy()), persistent=False)
        self.log_std = nn.Parameter(torch.as_tensor(-std_offset * np.ones(act_dim, dtype=np.float32)))
        self.rng = _PhiloxStream(seed)

    def bound_tensors(self):
        """(lo, hi) float32[D] device tensors for the kernels, or (None, None) for the unit box."""
        return (self.dist_min_t, self.dist_max_t) if self.rescale else (None, None)

    def get_distribution(self, action_mean):
        return action_mean

    def sample_distribution(self, mean):
        seed, off = self.rng.take(mean.shape[0])
        lo, hi = self.bound_tensors()
        raw, act, lp = K.gaussian_tanh_sample(mean.contiguous(), self.log_std.detach(), seed, off,
                                              self.min_std, lo, hi)
        return act, raw, lp.unsqueeze(-1)

    def get_log_probs_and_entropy(self, mean, raw_actions):
        logp, ent = _GaussianTanhEval.apply(mean, self.log_std, raw_actions.contiguous(), self.min_std)
        return logp.unsqueeze(-1), ent

    def refine_prediction(self, mean):
        s = torch.tanh(mean)
        if self.rescale:                                                       # :580-581, :604-609
            s = ((s + 1.0) / 2.0) * (self.dist_max_t - self.dist_min_t) + self.dist_min_t
        return s


class MultiCategoricalDistribution:
    """
    networks/distributions.py:272-438 for MultiDiscrete action spaces: one categorical per action dimension over
    its slice of the actor output (the per-slice softmax of the output function, :1046-1056, is fused in as for
    the single categorical); log-probs and entropies are summed over the dimensions.  Each slice runs the K6
    categorical kernels.
    """

    def __init__(self, nvec, seed=0, **kw_args):
        self.nvec = [int(n) for n in nvec]
        self.rng = _PhiloxStream(seed)

    def _slices(self, logits):
        start = 0
        for n in self.nvec:
            yield logits[:, start:start + n].contiguous()
            start += n

    def get_distribution(self, logits):
        return logits

    def sample_distribution(self, logits):
        acts, lps = [], []
        for sl in self._slices(logits):
            seed, off = self.rng.take(sl.shape[0])
            a, lp = K.categorical_sample(sl, seed, off)
            acts.append(a); lps.append(lp)
        a = torch.stack(acts, dim=1)
        return a, a, torch.stack(lps, dim=-1).sum(dim=-1, keepdim=True)

    def get_log_probs_and_entropy(self, logits, actions):
        actions = actions.reshape(logits.shape[0], -1)
        lps, ents = [], []
        for d, sl in enumerate(self._slices(logits)):
            lp, ent = _CategoricalEval.apply(sl, actions[:, d].contiguous())
            lps.append(lp); ents.append(ent)
        return torch.stack(lps, dim=-1).sum(dim=-1, keepdim=True), torch.stack(ents, dim=-1).sum(dim=-1)

    def refine_prediction(self, logits):
        return torch.stack([torch.argmax(sl, dim=-1) for sl in self._slices(logits)], dim=-1)


class BernoulliDistribution:
    """
    networks/distributions.py:134-196 for MultiBinary action spaces: independent Bernoulli(sigmoid(logit)) per
    bit (the sigmoid is the actor's output function), log-probs summed over the bits, entropy likewise
    (torch.distributions.Bernoulli semantics incl. its probability clamp), elementwise torch-ROCm ops.
    """

    def __init__(self, seed=0, **kw_args):
        self.rng = _PhiloxStream(seed)

    def get_distribution(self, logits):
        return logits

    @staticmethod
    def _dist(logits):
        # validate_args=False: the argument check synchronises with the host, which a hipGraph capture forbids
        return torch.distributions.Bernoulli(probs=torch.sigmoid(logits), validate_args=False)

    def sample_distribution(self, logits):
        seed, off = self.rng.take(logits.numel())
        g = torch.Generator(device=logits.device).manual_seed((int(seed) * 1000003 + int(off)) % (1 << 62))
        a = (torch.rand(logits.shape, device=logits.device, generator=g) < torch.sigmoid(logits)).float()
        lp = self._dist(logits).log_prob(a).sum(dim=-1, keepdim=True)
        return a, a, lp

    def get_log_probs_and_entropy(self, logits, actions):
        d = self._dist(logits)
        return d.log_prob(actions.reshape(logits.shape).float()).sum(dim=-1, keepdim=True), d.entropy().sum(dim=-1)

    def refine_prediction(self, logits):
        return (torch.sigmoid(logits) >= 0.5).float()


def get_actor_distribution(action_space, seed=0, **kw_args):
    """networks/distributions.py:984-1115 for Discrete, MultiDiscrete, MultiBinary and Box action spaces."""
    dtype = get_space_dtype_str(action_space)
    if dtype == "discrete":
        return CategoricalDistribution(seed=seed)
    if dtype == "multi-discrete":
        return MultiCategoricalDistribution(action_space.nvec, seed=seed)
    if dtype == "multi-binary":
        return BernoulliDistribution(seed=seed)
    if dtype == "continuous":
        # :1058-1107: bounds given through the actor's kw_args win; else the action space's, which must be finite
        kw_args = dict(kw_args)
        for key, side in (("distribution_min", "low"), ("distribution_max", "high")):
            if kw_args.get(key) is None:
                bound = np.asarray(getattr(action_space, side), dtype=np.float32)
                if np.isinf(bound).any():
                    raise ValueError(f"the action space's {side} bound {bound} is infinite: the Gaussian {key} must be "
                                     f"finite -- set it through the actor kw_args (actor_kw_args['{key}'] = k)")
                kw_args[key] = bound
        return GaussianDistribution(int(np.prod(action_space.shape)), seed=seed, **kw_args)
    raise NotImplementedError(f"action space dtype {dtype} is outside this build's hot path "
                              "(Discrete and Box only; SURVEY.md §2.1 #4)")
