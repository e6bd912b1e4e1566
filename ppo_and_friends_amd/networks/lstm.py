"""
LSTM actor / critic network -- stand-in for networks/ppo_networks/lstm.py:13-127 and the
PPOLSTMNetwork base (networks/ppo_networks/base.py:136-185) of the reference, with the same
sub-module names (lstm, layer_norm, ff_layers.sequential_net.*), hence the same state_dict keys.

The recurrent forward / backward is torch-ROCm (`nn.LSTM` -> MIOpen), as north_star allows for
the network itself; everything around it (rollout buffer with the per-step hidden states, window
gather, losses, optimiser) is this package's device path.

Semantics kept from the reference:
  * the network is STATEFUL: `hidden_state` persists between forward calls and is replaced only
    when the batch size changes (lstm.py:109-113) -- a rollout steps it once per env step, an
    update assigns it from the dataset before every mini-batch (ppo.py:2312-2319);
  * a 2-D input [B, O] is one time step for B sequences, a 3-D input [B, S, O] a window of S
    steps (lstm.py:103-107); the output is computed from the LAST layer's final hidden state ->
    LayerNorm -> activation -> feed-forward head (lstm.py:115-127).
"""
import torch
import torch.nn as nn

from .feed_forward import FeedForwardNetwork, PPONetwork


class PPOLSTMNetwork(PPONetwork):
    """base.py:136-185."""

    def get_zero_hidden_state(self, batch_size, device):
        hidden = torch.zeros(self.num_lstm_layers, batch_size, self.lstm_hidden_size, dtype=torch.float32,
                             device=device)
        return hidden, torch.zeros_like(hidden)

    def reset_hidden_state(self, batch_size, device):
        self.hidden_state = self.get_zero_hidden_state(batch_size, device)


class LSTMNetwork(PPOLSTMNetwork):

    def __init__(self, in_shape, out_shape, sequence_length=10, out_init=None, activation=None,
                 lstm_hidden_size=128, num_lstm_layers=1, ff_hidden_size=128, ff_hidden_depth=1, **kw_args):
        super().__init__(in_shape=in_shape, out_shape=out_shape, **kw_args)
        self.sequence_length = int(sequence_length)
        self.activation = nn.ReLU() if activation is None else activation
        self.lstm_hidden_size = int(lstm_hidden_size)
        self.num_lstm_layers = int(num_lstm_layers)
        self.lstm = nn.LSTM(self.in_size, self.lstm_hidden_size, self.num_lstm_layers)
        for name, param in self.lstm.named_parameters():          # networks/utils.py:83-111
            if "weight" in name:
                nn.init.orthogonal_(param, 2 ** 0.5)
            elif "bias" in name:
                nn.init.constant_(param, 0.0)
        self.layer_norm = nn.LayerNorm(self.lstm_hidden_size)
        self.hidden_state = None
        ff_kw_args = dict(kw_args)
        ff_kw_args["name"] = self.name + "_lstm_ff"
        self.ff_layers = FeedForwardNetwork(in_shape=self.lstm_hidden_size, out_shape=self.out_shape,
                                            hidden_size=ff_hidden_size, hidden_depth=ff_hidden_depth,
                                            activation=self.activation, is_embedded=False, out_init=out_init,
                                            **ff_kw_args)

    def forward_logits(self, _input):
        """Everything before output_func (what the HIP distribution kernels consume)."""
        out = _input.unsqueeze(0) if _input.dim() == 2 else torch.transpose(_input, 0, 1)
        batch_size = out.shape[1]
        if self.hidden_state is None or self.hidden_state[0].shape[1] != batch_size:
            self.reset_hidden_state(batch_size, out.device)
        h, c = self.hidden_state
        _, self.hidden_state = self.lstm(out.contiguous(), (h.contiguous(), c.contiguous()))
        out = self.hidden_state[0][-1]
        out = self.activation(self.layer_norm(out))
        return self.ff_layers.forward_logits(out)

    def forward(self, _input):
        return self._shape_output(self.output_func(self.forward_logits(_input)))
