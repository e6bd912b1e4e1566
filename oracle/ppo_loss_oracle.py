"""
TEST INFRASTRUCTURE (see oracle/__init__.py) -- torch-CPU float32 restatement of
the reference's mini-batch loss, distributions and optimiser tail.

PINNED by fixtures recorded from the unmodified reference (tests/golden/make_golden_update.py; the reference
imports here with a metadata-only `gymnasium` stand-in): g8_distributions (log-probs, entropies, their gradients,
refine_*), g12_* (first-mini-batch losses + raw gradients, per-epoch statistics, weights after the optimiser
steps) -- tests/test_oracle_update_golden.py.  The restatement follows the reference text line by line on torch's
own primitives (the same third-party code the reference calls): torch.distributions.Categorical / Normal,
nn.MSELoss / nn.HuberLoss, nn.utils.clip_grad_norm_, torch.optim.Adam.

  ppo_minibatch_losses   <- PPO._ppo_batch_train              ppo.py:2325-2438
  categorical_*          <- CategoricalDistribution           networks/distributions.py:199-269
                            softmax output_func               :1043-1045
                            PPODistribution.get_entropy       :93-111
  gaussian_*             <- GaussianDistribution              networks/distributions.py:441-694
  update_weights         <- PPOPolicy.update_weights          policies/ppo_policy.py:1012-1055
"""
import torch
import torch.nn as nn
from torch.distributions import Categorical
from torch.distributions.normal import Normal


def ppo_minibatch_losses(cur_logp, old_logp, adv, entropy, values, rtg, normalize_adv=True,
                         surr_clip=0.2, entropy_weight=0.01, kl_loss_weight=0.0,
                         use_huber=False):
    """Returns dict of python floats + the two loss tensors (graph attached)."""
    advantages = adv
    adv_mean = adv_std = None
    if normalize_adv:                                   # ppo.py:2325-2333
        adv_std = advantages.std()
        adv_mean = advantages.mean()
        advantages = (advantages - adv_mean) / (adv_std + 1e-8)
    cur_logp = cur_logp.flatten(); old_logp = old_logp.flatten()
    advantages = advantages.flatten(); entropy = entropy.flatten()
    values = values.flatten(); rtg = rtg.flatten()
    ratios = torch.exp(cur_logp - old_logp)             # :2352
    surr1 = ratios * advantages
    surr2 = torch.clamp(ratios, 1 - surr_clip, 1 + surr_clip) * advantages
    current_kl = (old_logp - cur_logp).mean().item()    # :2358
    bad = bool(torch.isnan(ratios).any() or torch.isinf(ratios).any())
    actor_loss = (-torch.min(surr1, surr2)).mean()      # :2392
    surr_loss = actor_loss.item()
    ent_mean = 0.0
    if entropy_weight != 0.0:                           # :2395-2398
        ent_mean = entropy.mean().item()
        actor_loss = actor_loss - entropy_weight * entropy.mean()
    if kl_loss_weight > 0.0:                            # :2403-2405 (python float: no gradient)
        actor_loss = actor_loss + kl_loss_weight * current_kl
    if use_huber:                                       # :2416-2419
        critic_loss = nn.HuberLoss(delta=10.0)(values, rtg)
    else:
        critic_loss = nn.MSELoss()(values, rtg)
    return dict(surr=surr_loss, actor=actor_loss.item(), critic=critic_loss.item(),
                entropy=ent_mean, kl=current_kl, bad=bad,
                adv_mean=None if adv_mean is None else adv_mean.item(),
                adv_std=None if adv_std is None else adv_std.item(),
                actor_loss=actor_loss, critic_loss=critic_loss)


def categorical_from_logits(logits):
    """Actor head for Discrete spaces: softmax output_func then Categorical(probs) (:1043-1045, :217)."""
    probs = torch.softmax(logits, dim=-1)
    return probs, Categorical(probs)


def categorical_logp_entropy(logits, actions):
    probs, dist = categorical_from_logits(logits)
    logp = torch.unsqueeze(dist.log_prob(actions.flatten()), dim=-1)      # :240
    ent = dist.entropy()
    if len(ent.shape) <= 1:                                               # :106-111
        ent = torch.unsqueeze(ent, dim=-1)
    return logp.flatten(), ent.sum(dim=-1), probs


def gaussian_dist(mean, log_std, min_std=0.01):
    std = nn.functional.softplus(log_std)                                 # :514
    std = torch.max(std, torch.tensor([min_std], dtype=torch.float32))    # :515
    return Normal(mean, std)


def gaussian_tanh_logp(mean, log_std, x, min_std=0.01, epsilon=1e-6):
    dist = gaussian_dist(mean, log_std, min_std)
    nlp = dist.log_prob(x)                                                # :551
    nlp = torch.clamp(nlp, -100, 100).sum(dim=-1)
    tanh_prime = 1.0 - torch.pow(torch.tanh(x), 2)
    tanh_prime = torch.clamp(tanh_prime, epsilon, None)
    s_log = torch.log(tanh_prime).sum(dim=-1)
    return nlp - s_log                                                    # :558


def gaussian_refine(sample, lo=-1.0, hi=1.0):
    """lo / hi: scalars or per-dimension arrays (dist_min / dist_max are numpy arrays in the reference, :476-483)."""
    import numpy as np
    lo = np.atleast_1d(np.asarray(lo, dtype=np.float32))
    hi = np.atleast_1d(np.asarray(hi, dtype=np.float32))
    s = torch.tanh(sample)                                                # :604
    if (lo != -1.0).any() or (hi != 1.0).any():                           # :606
        s = ((s + 1.0) / 2.0) * (hi - lo) + lo                            # :580-581 (torch tensor x numpy array)
    return s


def clip_adam_reference(params, grads, steps, lr=3e-4, eps=1e-5, max_norm=0.5, grad_scale=1.0):
    """
    Runs `steps` optimiser steps on a list of parameter tensors with the given
    per-step gradient lists; mirrors ppo_policy.py:1032-1042 (+ the /num_procs
    of mpi_utils.py:86 as grad_scale).  Returns the final params and the norms.
    """
    ps = [torch.nn.Parameter(p.clone()) for p in params]
    opt = torch.optim.Adam(ps, lr=lr, eps=eps)
    norms = []
    for s in range(steps):
        opt.zero_grad()
        for p, g in zip(ps, grads[s]):
            p.grad = (g * grad_scale).clone()
        if max_norm is not None:
            norms.append(float(nn.utils.clip_grad_norm_(ps, max_norm)))
        opt.step()
    return [p.detach() for p in ps], norms
