"""
Actor / critic MLPs with flat parameter and gradient buckets.

Architecture and initialisation follow the reference so that state_dicts are
interchangeable (networks/ppo_networks/feed_forward.py:14-86,
networks/utils.py:53-80,120-191): Linear(in,h0) -> act -> Sequential(Linear,act,...)
-> Linear(h,out); orthogonal weights (gain sqrt(2), output gain `out_init`),
zero biases; state_dict keys `sequential_net.0`, `sequential_net.2.<i>`,
`sequential_net.3`.

MI355X-first difference: every parameter is a view into ONE contiguous float32
bucket (`flat_params`) and every gradient a view into `flat_grads`, so the
DD-PPO exchange is one RCCL all-reduce and clip+Adam one fused HIP kernel pair
instead of per-tensor Python loops (policies/ppo_policy.py:1032-1055).
"""
import numpy as np
import torch
import torch.nn as nn

from ..utils.mpi_utils import get_rank


def init_layer(layer, gain=np.sqrt(2), bias_const=0.0):
    """networks/utils.py:53-80."""
    nn.init.orthogonal_(layer.weight, gain)
    if layer.bias is not None:
        nn.init.constant_(layer.bias, bias_const)
    return layer


def create_sequential_network(in_size, out_size, hidden_size, hidden_depth, activation, out_init=None):
    """networks/utils.py:120-191 (same module tree, hence same state_dict keys)."""
    if not isinstance(hidden_size, list):
        if (hidden_size == 0) != (hidden_depth == 0):
            raise ValueError("if either hidden_size or hidden_depth is 0, both must be 0 "
                             f"(got {hidden_size}, {hidden_depth})")
        hidden_size = [hidden_size] * hidden_depth
    else:
        hidden_depth = len(hidden_size)
    out_gain = {} if out_init is None else {"gain": out_init}
    layers = []
    if len(hidden_size) != 0:
        layers.append(init_layer(nn.Linear(in_size, hidden_size[0])))
        layers.append(activation)
        inner = []
        for i in range(hidden_depth - 1):
            inner.append(init_layer(nn.Linear(hidden_size[i], hidden_size[i + 1])))
            inner.append(activation)
        layers.append(nn.Sequential(*inner))
        layers.append(init_layer(nn.Linear(hidden_size[-1], out_size), **out_gain))
    else:
        layers.append(init_layer(nn.Linear(in_size, out_size), **out_gain))
    return nn.Sequential(*layers)


class FlatBucketModule(nn.Module):
    """nn.Module whose parameters / gradients live in two flat float32 buckets."""

    def __init__(self):
        super().__init__()
        self.flat_params = None
        self.flat_grads = None

    def bucket_size(self):
        """Elements of the flat bucket (every tensor padded to a multiple of 4 floats = 16 B)."""
        return sum((p.numel() + 3) // 4 * 4 for p in self.parameters())

    def flatten_parameters_(self, device=None, storage=None):
        """
        (Re)build the buckets on `device`; parameters become views into flat_params.
        `storage` = (params_slice, grads_slice) places them inside a larger bucket
        (actor and critic adjacent -> one all-reduce for both).
        """
        params = [p for p in self.parameters()]
        device = params[0].device if device is None else torch.device(device)
        n = sum(p.numel() for p in params)
        # keep every tensor's offset 16-byte aligned so float4 kernels can take sub-ranges
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        if storage is None:
            flat = torch.zeros(total, dtype=torch.float32, device=device)
            grads = torch.zeros(total, dtype=torch.float32, device=device)
        else:
            flat, grads = storage
            assert flat.numel() == total and grads.numel() == total and flat.device == device
        for p, o in zip(params, offs):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.grad = None
            p.data = flat[o:o + p.numel()].view(p.shape)
            p.grad = grads[o:o + p.numel()].view(p.shape)
        for b in self.buffers():
            b.data = b.data.to(device)
        self.flat_params, self.flat_grads = flat, grads
        self.num_params = n
        return self

    def to(self, *args, **kwargs):
        device = None
        for a in args:
            if isinstance(a, (str, torch.device)):
                device = torch.device(a)
        device = kwargs.get("device", device)
        if device is None:
            return super().to(*args, **kwargs)
        return self.flatten_parameters_(device)

    def zero_grad(self, set_to_none=False):
        # gradients must stay views of the bucket: never drop them
        if self.flat_grads is not None:
            self.flat_grads.zero_()
        else:
            super().zero_grad(set_to_none=False)


class PPONetwork(FlatBucketModule):
    """networks/ppo_networks/base.py:17-127 (name, save/load by `<name>_<rank>.model`)."""

    def __init__(self, in_shape, out_shape, name="ppo-network", test_mode=False, **kw_args):
        super().__init__()
        self.in_shape = (in_shape,) if isinstance(in_shape, int) else tuple(in_shape)
        self.out_shape = (out_shape,) if isinstance(out_shape, int) else tuple(out_shape)
        self.in_size = int(np.prod(self.in_shape))
        self.out_size = int(np.prod(self.out_shape))
        self.output_func = lambda x: x
        self.name = name
        self.test_mode = test_mode

    def _shape_output(self, output):
        return output.reshape((output.shape[0],) + self.out_shape)

    def save(self, path):
        import os
        if self.test_mode:
            return
        torch.save({k: v.detach().cpu().clone() for k, v in self.state_dict().items()},
                   os.path.join(path, "{}_{}.model".format(self.name, get_rank())))

    def load(self, path):
        import os
        f = os.path.join(path, "{}_{}.model".format(self.name, 0 if self.test_mode else get_rank()))
        if not os.path.exists(f):
            f = os.path.join(path, "{}_0.model".format(self.name))
        sd = torch.load(f, map_location="cpu")
        with torch.no_grad():
            for k, v in self.state_dict().items():
                v.copy_(sd[k])


class FeedForwardNetwork(PPONetwork):
    """networks/ppo_networks/feed_forward.py:14-86: defaults hidden 128 x depth 3, ReLU."""

    def __init__(self, in_shape, out_shape, out_init=None, activation=None, hidden_size=128,
                 hidden_depth=3, is_embedded=False, **kw_args):
        super().__init__(in_shape=in_shape, out_shape=out_shape, **kw_args)
        self.is_embedded = is_embedded
        self.activation = nn.ReLU() if activation is None else activation
        self.sequential_net = create_sequential_network(
            in_size=self.in_size, out_size=self.out_size, hidden_size=hidden_size,
            hidden_depth=hidden_depth, activation=self.activation, out_init=out_init)

    def forward_logits(self, _input):
        """The Sequential alone (pre output_func): what the HIP distribution kernels consume."""
        return self.sequential_net(_input.flatten(start_dim=1))

    def forward(self, _input):
        out = self.forward_logits(_input)
        if self.is_embedded:
            return self.activation(out)
        return self._shape_output(self.output_func(out))

    def layer_dims(self):
        """[(in, out), ...] of the Linear layers in order (for the fused MLP kernels)."""
        return [(m.in_features, m.out_features) for m in self.sequential_net.modules()
                if isinstance(m, nn.Linear)]
