"""
Intrinsic Curiosity Module -- stand-in for networks/ppo_networks/icm.py:22-430 and
networks/encoders.py:9-56 of the reference (same sub-module names, hence the same state_dict
keys: obs_encoder.enc_1..4, inv_model.sequential_net.*, forward_model.sequential_net.*).

The MLPs run on torch-ROCm (north_star: network forward/backward may be PyTorch-ROCm); the
element-wise tail of ICM.forward -- forward-model squared error, intrinsic reward, 0.5*mean -- is
the fused HIP kernel K8 (csrc/icm.hip), forward and backward.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as t_functional

from .. import kernels as K
from ..spaces import (get_action_prediction_shape, get_flattened_space_length,
                      get_space_dtype_str, get_space_shape)
from .feed_forward import PPONetwork, create_sequential_network, init_layer


class LinearObservationEncoder(nn.Module):
    """networks/encoders.py:9-56: obs -> 3 hidden layers -> encoded_dim (last gain = out_init)."""

    def __init__(self, obs_shape, encoded_dim, out_init, hidden_size, activation=None, **kw_args):
        super().__init__()
        self.activation = nn.ReLU() if activation is None else activation
        obs_size = int(np.prod(obs_shape))
        self.enc_1 = init_layer(nn.Linear(obs_size, hidden_size))
        self.enc_2 = init_layer(nn.Linear(hidden_size, hidden_size))
        self.enc_3 = init_layer(nn.Linear(hidden_size, hidden_size))
        self.enc_4 = init_layer(nn.Linear(hidden_size, encoded_dim), gain=out_init)

    def forward(self, obs):
        x = obs.flatten(start_dim=1)
        x = self.activation(self.enc_1(x))
        x = self.activation(self.enc_2(x))
        x = self.activation(self.enc_3(x))
        return self.enc_4(x)


class LinearInverseModel(nn.Module):
    """icm.py:22-114: (enc_1, enc_2) -> predicted action (softmax output for discrete actions)."""

    def __init__(self, in_size, out_size, out_init, hidden_size, hidden_depth, action_dtype,
                 activation=None, **kw_args):
        super().__init__()
        self.action_dtype = action_dtype
        self.activation = nn.ReLU() if activation is None else activation
        self.sequential_net = create_sequential_network(in_size, out_size, hidden_size, hidden_depth,
                                                        self.activation, out_init)

    def forward(self, enc_obs_1, enc_obs_2):
        out = self.sequential_net(torch.cat((enc_obs_1, enc_obs_2), dim=1))
        if self.action_dtype in ("discrete", "multi-discrete"):
            # icm.py:76-77: ONE softmax over the whole output row, also when it holds several dims' classes
            out = t_functional.softmax(out, dim=-1)
        return out


class LinearForwardModel(nn.Module):
    """icm.py:117-211: (enc_1, action [one-hot for discrete]) -> predicted enc_2."""

    def __init__(self, in_size, out_size, out_init, hidden_size, hidden_depth, action_dtype, n_classes,
                 activation=None, action_nvec=None, **kw_args):
        super().__init__()
        self.action_dtype = action_dtype
        self.n_classes = n_classes
        self.action_nvec = None if action_nvec is None else [int(n) for n in action_nvec]
        self.activation = nn.ReLU() if activation is None else activation
        self.sequential_net = create_sequential_network(in_size, out_size, hidden_size, hidden_depth,
                                                        self.activation, out_init)

    def forward(self, enc_obs_1, actions):
        if self.action_dtype == "discrete":
            actions = t_functional.one_hot(actions, num_classes=self.n_classes).float().flatten(start_dim=1)
        elif self.action_dtype == "multi-discrete":
            # icm.py:198-211 literally: the loop slices the action COLUMNS by class offsets
            # (start:stop walks nvec), so with nvec = [n] * A the first slices take all the columns and the
            # rest are empty -- for equal class counts that equals one_hot(actions, n) flattened
            one_hots, start = [], 0
            for dim in self.action_nvec:
                stop = start + dim
                one_hots.append(t_functional.one_hot(actions[:, start:stop], num_classes=dim).float().flatten(start_dim=1))
                start = stop
            actions = torch.cat(one_hots, dim=1)
        return self.sequential_net(torch.cat((enc_obs_1, actions), dim=1))


class _ForwardLoss(torch.autograd.Function):
    """K8: (pred, enc2) -> (intrinsic_reward [n], f_loss scalar); gradient of f_loss only, as used."""

    @staticmethod
    def forward(ctx, pred, enc2, reward_scale):
        pred, enc2 = pred.contiguous(), enc2.contiguous()
        intr, f_loss = K.icm_forward_loss_fwd(pred, enc2, reward_scale)
        ctx.save_for_backward(pred, enc2)
        ctx.mark_non_differentiable(intr)
        return intr, f_loss.reshape(())

    @staticmethod
    def backward(ctx, g_intr, g_loss):
        pred, enc2 = ctx.saved_tensors
        d_pred, d_enc2 = K.icm_forward_loss_bwd(pred, enc2, g_loss.contiguous().float())
        return d_pred, d_enc2, None


class ICM(PPONetwork):
    """icm.py:214-430.  forward(obs_1, obs_2, actions) -> (intrinsic_reward, inv_loss, f_loss)."""

    def __init__(self, obs_space, action_space, out_init=1.0, obs_encoder=LinearObservationEncoder,
                 reward_scale=0.01, activation=None, encoded_obs_dim=128, encoder_hidden_size=128,
                 inverse_hidden_size=128, inverse_hidden_depth=2, forward_hidden_size=128,
                 forward_hidden_depth=2, **kw_args):
        kw_args.pop("in_shape", None); kw_args.pop("out_shape", None)
        super().__init__(in_shape=get_space_shape(obs_space), out_shape=(1,), **kw_args)
        self.reward_scale = reward_scale
        self.action_dtype = get_space_dtype_str(action_space)
        if self.action_dtype not in ("discrete", "multi-discrete", "continuous"):
            raise NotImplementedError(f"ICM for {self.action_dtype} actions is outside the hot-path scope")
        self.obs_space, self.action_space = obs_space, action_space
        self.action_nvec = [int(n) for n in action_space.nvec] if hasattr(action_space, "nvec") else None
        act_size = get_action_prediction_shape(action_space)[0]
        self.activation = nn.ReLU() if activation is None else activation
        self.ce_loss = nn.CrossEntropyLoss(reduction="mean")
        if encoded_obs_dim > 0:
            self.obs_encoder = obs_encoder(get_space_shape(obs_space), encoded_obs_dim, out_init,
                                           encoder_hidden_size, activation=self.activation)
        else:
            self.obs_encoder = nn.Identity()
            encoded_obs_dim = get_flattened_space_length(obs_space)
        self.inv_model = LinearInverseModel(encoded_obs_dim * 2, act_size, out_init, inverse_hidden_size,
                                            inverse_hidden_depth, self.action_dtype, activation=self.activation)
        self.forward_model = LinearForwardModel(encoded_obs_dim + act_size, encoded_obs_dim, out_init,
                                                forward_hidden_size, forward_hidden_depth, self.action_dtype,
                                                act_size, activation=self.activation, action_nvec=self.action_nvec)

    def forward(self, obs_1, obs_2, actions):
        enc_obs_1 = self.obs_encoder(obs_1)
        enc_obs_2 = self.obs_encoder(obs_2)
        action_pred = self.inv_model(enc_obs_1, enc_obs_2)
        if self.action_dtype == "discrete":
            # the reference feeds its softmax output to CrossEntropyLoss (a second log-softmax): kept
            actions = actions.reshape(actions.shape[0], -1)[:, 0]
            inv_loss = self.ce_loss(action_pred, actions)
            fwd_actions = actions
        elif self.action_dtype == "multi-discrete":
            # icm.py:400-412: one cross entropy per action dim on its slice of the (jointly soft-maxed) row, summed
            actions = actions.reshape(actions.shape[0], -1)
            inv_loss, start = 0, 0
            for idx, dim in enumerate(self.action_nvec):
                stop = start + dim
                inv_loss = inv_loss + self.ce_loss(action_pred[:, start:stop], actions[:, idx])
                start = stop
            fwd_actions = actions
        else:
            fwd_actions = actions.reshape(action_pred.shape)
            inv_loss = ((action_pred - fwd_actions) ** 2).mean()
        obs_2_pred = self.forward_model(enc_obs_1, fwd_actions)
        intrinsic_reward, f_loss = _ForwardLoss.apply(obs_2_pred, enc_obs_2, self.reward_scale)
        return intrinsic_reward, inv_loss, f_loss
