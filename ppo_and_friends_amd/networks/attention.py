"""
Self-attention blocks of the multi-agent transformer -- stand-in for networks/attention.py:13-257
(same module / parameter names: key_net, query_net, value_net, proj, ln1.., attn.., mlp.N).

The q/k/v/proj and MLP linears, LayerNorm and GELU run on torch-ROCm; the attention core
(scores, mask, softmax, att @ v) is the f32-MFMA kernel K9 (csrc/mat_attention.hip), forward and
backward, with floor(16/L) agent sequences packed per MFMA tile.
"""
import torch
import torch.nn as nn

from .. import kernels as K
from .feed_forward import init_layer


class _AttentionCore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, masked):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        y, probs = K.mat_attention_fwd(q, k, v, masked)
        ctx.save_for_backward(q, k, v, probs)
        return y

    @staticmethod
    def backward(ctx, dy):
        q, k, v, probs = ctx.saved_tensors
        dq, dk, dv = K.mat_attention_bwd(q, k, v, probs, dy.contiguous())
        return dq, dk, dv, None


class SelfAttention(nn.Module):
    """attention.py:13-108."""

    def __init__(self, embedding_size, num_heads, num_agents, internal_init=0.01, out_init=0.01, masked=False):
        super().__init__()
        assert embedding_size % num_heads == 0
        self.masked = masked
        self.num_heads = num_heads
        self.key_net = init_layer(nn.Linear(embedding_size, embedding_size), gain=internal_init)
        self.query_net = init_layer(nn.Linear(embedding_size, embedding_size), gain=internal_init)
        self.value_net = init_layer(nn.Linear(embedding_size, embedding_size), gain=internal_init)
        self.proj = init_layer(nn.Linear(embedding_size, embedding_size), out_init)
        self.register_buffer("mask", torch.tril(torch.ones(num_agents + 1, num_agents + 1)).view(
            1, 1, num_agents + 1, num_agents + 1))

    def forward(self, key, value, query):
        B, L, D = query.size()
        H, hs = self.num_heads, D // self.num_heads
        split = lambda t: t.view(B, L, H, hs).transpose(1, 2).reshape(B * H, L, hs)
        k, q, v = split(self.key_net(key)), split(self.query_net(query)), split(self.value_net(value))
        y = _AttentionCore.apply(q, k, v, self.masked)                      # K9
        y = y.view(B, H, L, hs).transpose(1, 2).contiguous().view(B, L, D)
        return self.proj(y)


class SelfAttentionEncodingBlock(nn.Module):
    """attention.py:111-172."""

    def __init__(self, embedding_size, num_heads, num_agents, activation=None,
                 internal_init=nn.init.calculate_gain('relu'), out_init=0.01,
                 self_atten_internal_init=0.01, self_atten_out_init=0.01, **kw_args):
        super().__init__()
        activation = nn.GELU() if activation is None else activation
        self.ln1 = nn.LayerNorm(embedding_size)
        self.ln2 = nn.LayerNorm(embedding_size)
        self.attn = SelfAttention(embedding_size, num_heads, num_agents, internal_init=self_atten_internal_init,
                                  out_init=self_atten_out_init, masked=False)
        self.mlp = nn.Sequential(init_layer(nn.Linear(embedding_size, embedding_size), gain=internal_init),
                                 activation,
                                 init_layer(nn.Linear(embedding_size, embedding_size), gain=out_init))

    def forward(self, x):
        x = self.ln1(x + self.attn(x, x, x))
        x = self.ln2(x + self.mlp(x))
        return x


class SelfAttentionDecodingBlock(nn.Module):
    """attention.py:175-257 (cross-attention: query = encoded observations, key/value = action stream)."""

    def __init__(self, embedding_size, num_heads, num_agents, mlp_hidden_scale=1, activation=None,
                 internal_init=nn.init.calculate_gain('relu'), out_init=0.01,
                 self_atten_internal_init=0.01, self_atten_out_init=0.01, **kw_args):
        super().__init__()
        activation = nn.GELU() if activation is None else activation
        self.ln1 = nn.LayerNorm(embedding_size)
        self.ln2 = nn.LayerNorm(embedding_size)
        self.ln3 = nn.LayerNorm(embedding_size)
        self.attn1 = SelfAttention(embedding_size, num_heads, num_agents, internal_init=self_atten_internal_init,
                                   out_init=self_atten_out_init, masked=True)
        self.attn2 = SelfAttention(embedding_size, num_heads, num_agents, internal_init=self_atten_internal_init,
                                   out_init=self_atten_out_init, masked=True)
        self.mlp = nn.Sequential(
            init_layer(nn.Linear(embedding_size, mlp_hidden_scale * embedding_size), gain=internal_init),
            activation,
            init_layer(nn.Linear(mlp_hidden_scale * embedding_size, embedding_size), gain=out_init))

    def forward(self, x, rep_enc):
        x = self.ln1(x + self.attn1(x, x, x))
        x = self.ln2(rep_enc + self.attn2(key=x, value=x, query=rep_enc))
        x = self.ln3(x + self.mlp(x))
        return x
