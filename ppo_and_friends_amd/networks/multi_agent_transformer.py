"""
Multi-Agent Transformer actor / critic -- stand-in for
networks/actor_critic/multi_agent_transformer.py:22-373 (same module names -> same state_dict keys:
actor.action_encoder.0, actor.ln, actor.blocks.N.*, actor.head.0/3, critic.obs_encoder.0/1,
critic.ln, critic.blocks.N.*, critic.head.0/3).

One difference in convention: MATActor.forward returns the head's raw output (logits for discrete
actions, means for continuous ones).  The reference applies its softmax `output_func` there and
builds Categorical(probs) on the CPU; here the HIP distribution kernels consume logits and fuse the
softmax (networks/distributions.py in this package), which is the same function of the weights.
"""
import numpy as np
import torch
import torch.nn as nn

from ..spaces import (get_action_prediction_shape, get_flattened_space_length,
                      get_space_dtype_str, get_space_shape)
from .attention import SelfAttentionDecodingBlock, SelfAttentionEncodingBlock
from .distributions import get_actor_distribution
from .feed_forward import PPONetwork, init_layer

_RELU_GAIN = nn.init.calculate_gain('relu')


class MATActor(PPONetwork):
    """multi_agent_transformer.py:22-156."""

    def __init__(self, obs_space, action_space, num_agents, embedding_size=64, num_blocks=1, num_heads=1,
                 internal_init=_RELU_GAIN, out_init=0.01, activation=None, decoder_internal_init=_RELU_GAIN,
                 decoder_out_init=0.01, decoder_activation=None, self_atten_internal_init=0.01,
                 self_atten_out_init=0.01, seed=0, **kw_args):
        kw_args.pop("name", None)
        super().__init__(name="mat_actor", in_shape=(embedding_size,),
                         out_shape=get_action_prediction_shape(action_space), **kw_args)
        activation = nn.GELU() if activation is None else activation
        decoder_activation = nn.GELU() if decoder_activation is None else decoder_activation
        self.obs_space = obs_space
        self.distribution = get_actor_distribution(action_space, seed=seed)
        self.action_dtype = get_space_dtype_str(action_space)
        if self.action_dtype not in ("discrete", "continuous"):
            raise NotImplementedError(f"MAT with {self.action_dtype} actions is outside the hot-path scope")
        self.embedding_size = embedding_size
        self.num_agents = num_agents
        self.action_pred_size = self.out_size
        self.action_dim = get_flattened_space_length(action_space)
        if self.action_dtype == "discrete":     # +1: the start token column (mat_policy.py:325-333)
            self.action_encoder = nn.Sequential(
                init_layer(nn.Linear(self.action_pred_size + 1, embedding_size, bias=False), gain=internal_init),
                activation)
        else:
            self.action_encoder = nn.Sequential(
                init_layer(nn.Linear(self.action_pred_size, embedding_size), gain=internal_init), activation)
        self.ln = nn.LayerNorm(embedding_size)
        self.blocks = nn.Sequential(*[
            SelfAttentionDecodingBlock(embedding_size, num_heads, num_agents, activation=decoder_activation,
                                       internal_init=decoder_internal_init, out_init=decoder_out_init,
                                       self_atten_internal_init=self_atten_internal_init,
                                       self_atten_out_init=self_atten_out_init)
            for _ in range(num_blocks)])
        self.head = nn.Sequential(init_layer(nn.Linear(embedding_size, embedding_size), gain=internal_init),
                                  activation, nn.LayerNorm(embedding_size),
                                  init_layer(nn.Linear(embedding_size, self.action_pred_size), gain=out_init))

    def forward(self, actions, encoded_obs):
        x = self.ln(self.action_encoder(actions))
        for block in self.blocks:
            x = block(x, encoded_obs)
        return self.head(x)


class MATCritic(PPONetwork):
    """multi_agent_transformer.py:159-315."""

    def __init__(self, obs_space, num_agents, embedding_size=64, num_blocks=1, num_heads=1, out_init=0.01,
                 internal_init=_RELU_GAIN, activation=None, encoder_internal_init=_RELU_GAIN,
                 encoder_out_init=0.01, encoder_activation=None, self_atten_internal_init=0.01,
                 self_atten_out_init=0.01, **kw_args):
        kw_args.pop("name", None); kw_args.pop("action_space", None); kw_args.pop("seed", None)
        super().__init__(name="mat_critic", in_shape=get_space_shape(obs_space), out_shape=(1,), **kw_args)
        activation = nn.GELU() if activation is None else activation
        encoder_activation = nn.GELU() if encoder_activation is None else encoder_activation
        self.obs_space = obs_space
        self.embedding_size = embedding_size
        self.num_agents = num_agents
        self.obs_encoder = nn.Sequential(nn.LayerNorm(self.in_size),
                                         init_layer(nn.Linear(self.in_size, embedding_size), gain=internal_init),
                                         activation)
        self.ln = nn.LayerNorm(embedding_size)
        self.blocks = nn.Sequential(*[
            SelfAttentionEncodingBlock(embedding_size, num_heads, num_agents, activation=encoder_activation,
                                       internal_init=encoder_internal_init, out_init=encoder_out_init,
                                       self_atten_internal_init=self_atten_internal_init,
                                       self_atten_out_init=self_atten_out_init)
            for _ in range(num_blocks)])
        self.head = nn.Sequential(init_layer(nn.Linear(embedding_size, embedding_size), gain=internal_init),
                                  activation, nn.LayerNorm(embedding_size),
                                  init_layer(nn.Linear(embedding_size, 1), gain=out_init))

    def encode_obs(self, obs):
        return self.blocks(self.ln(self.obs_encoder(obs)))

    def forward(self, obs):
        encoded_obs = self.encode_obs(obs)
        return encoded_obs, self.head(encoded_obs)


class MATActorCritic(PPONetwork):
    """multi_agent_transformer.py:318-373."""

    def __init__(self, obs_space, action_space, num_agents, name="actor_critic", **kw_args):
        super().__init__(name=name, in_shape=get_space_shape(obs_space), out_shape=(1,),
                         test_mode=kw_args.pop("test_mode", False))
        self.actor = MATActor(obs_space=obs_space, action_space=action_space, num_agents=num_agents, **kw_args)
        self.critic = MATCritic(obs_space=obs_space, action_space=action_space, num_agents=num_agents, **kw_args)

    def forward(self, obs, action_block):
        encoded_obs, values = self.critic(obs)
        return values, self.actor(action_block, encoded_obs)
