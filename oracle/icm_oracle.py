"""
TEST INFRASTRUCTURE (see oracle/__init__.py) -- torch-CPU restatement of the reference's ICM.
PINNED by fixtures recorded from the unmodified reference: g10_icm (forward outputs + every parameter gradient,
discrete and continuous) and g12_c2_icm / g12_c3_full (rollout rewards, training epochs, final weights).

  ObsEncoder   <- LinearObservationEncoder   networks/encoders.py:9-56
  InverseModel <- LinearInverseModel         networks/ppo_networks/icm.py:22-114
  ForwardModel <- LinearForwardModel         :117-211
  ICM.forward  <- ICM.forward                :365-430
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .cpu_ppo_loop import make_mlp


def _lin(i, o, gain=np.sqrt(2)):
    layer = nn.Linear(i, o)
    nn.init.orthogonal_(layer.weight, gain)
    nn.init.constant_(layer.bias, 0.0)
    return layer


class ObsEncoder(nn.Module):
    def __init__(self, obs_size, encoded_dim=128, out_init=1.0, hidden=128):
        super().__init__()
        self.enc_1 = _lin(obs_size, hidden); self.enc_2 = _lin(hidden, hidden)
        self.enc_3 = _lin(hidden, hidden); self.enc_4 = _lin(hidden, encoded_dim, out_init)

    def forward(self, obs):
        x = torch.relu(self.enc_1(obs.flatten(start_dim=1)))
        x = torch.relu(self.enc_2(x))
        x = torch.relu(self.enc_3(x))
        return self.enc_4(x)


class ICM(nn.Module):
    def __init__(self, obs_size, act_size, discrete, reward_scale=0.01, out_init=1.0, enc=128, hidden=128, depth=2,
                 nvec=None, enc_hidden=None, inv_depth=None, fwd_depth=None):
        """nvec: the MultiDiscrete class counts of an agent-shared ICM (act_size = sum(nvec)); icm.py:322-324.
        enc_hidden / inv_depth / fwd_depth: encoder_hidden_size, inverse_hidden_depth, forward_hidden_depth when they
        differ from `hidden` / `depth` (icm.py:228-240)."""
        super().__init__()
        self.discrete, self.act_size, self.reward_scale = discrete, act_size, reward_scale
        self.nvec = None if nvec is None else [int(n) for n in nvec]
        self.obs_encoder = ObsEncoder(obs_size, enc, out_init, hidden if enc_hidden is None else enc_hidden)
        self.inv_model = nn.Module()
        self.inv_model.sequential_net = make_mlp(2 * enc, act_size, hidden, depth if inv_depth is None else inv_depth,
                                                 out_gain=out_init)
        self.forward_model = nn.Module()
        self.forward_model.sequential_net = make_mlp(enc + act_size, enc, hidden, depth if fwd_depth is None else fwd_depth,
                                                     out_gain=out_init)

    def forward(self, obs_1, obs_2, actions):
        e1, e2 = self.obs_encoder(obs_1), self.obs_encoder(obs_2)
        pred = self.inv_model.sequential_net(torch.cat((e1, e2), dim=1))
        if self.nvec is not None:                                             # "multi-discrete"
            pred = F.softmax(pred, dim=-1)                                    # icm.py:76-77: one softmax over the row
            inv_loss, start = 0, 0
            for idx, dim in enumerate(self.nvec):                             # :400-412
                inv_loss = inv_loss + nn.CrossEntropyLoss(reduction="mean")(pred[:, start:start + dim], actions[:, idx:idx + 1].flatten())
                start += dim
            hots, start = [], 0
            for dim in self.nvec:                                             # :198-211 (slices action COLUMNS by class offsets)
                hots.append(F.one_hot(actions[:, start:start + dim], num_classes=dim).float().flatten(start_dim=1))
                start += dim
            fa = torch.cat(hots, dim=1)
        elif self.discrete:
            pred = F.softmax(pred, dim=-1)                                    # icm.py:84-85
            inv_loss = nn.CrossEntropyLoss(reduction="mean")(pred, actions.squeeze(1))   # :413
            fa = F.one_hot(actions, num_classes=self.act_size).float().flatten(start_dim=1)   # :189-191
        else:
            actions = actions.reshape(pred.shape)
            inv_loss = nn.MSELoss(reduction="none")(pred, actions).mean()     # :417-419
            fa = actions
        obs_2_pred = self.forward_model.sequential_net(torch.cat((e1, fa), dim=1))
        f = nn.MSELoss(reduction="none")(obs_2_pred, e2)                      # :425
        return (self.reward_scale / 2.0) * f.sum(dim=-1), inv_loss, 0.5 * f.mean()   # :427-428
