"""
generate_policy / get_single_policy_defaults with the reference's signatures
(policies/utils.py:11-108).  policy_settings[name] is the reference's 5-tuple
(policy_class, actor_obs_space, critic_obs_space, action_space, policy_args).
"""
from .ppo_policy import PPOPolicy


def generate_policy(policy_name, policy_class, actor_observation_space, critic_observation_space,
                    action_space, test_mode, envs_per_proc, **kw_args):
    from .mat_policy import MATPolicy
    if policy_class is None:
        policy_class = PPOPolicy
    if policy_class not in (PPOPolicy, MATPolicy):
        raise NotImplementedError(f"policy_class {policy_class} is of unsupported type; "
                                  "supported: PPOPolicy, MATPolicy (policies/utils.py:45-52)")
    return policy_class(name=policy_name, action_space=action_space,
                        actor_observation_space=actor_observation_space,
                        critic_observation_space=critic_observation_space,
                        test_mode=test_mode, envs_per_proc=envs_per_proc, **kw_args)


def get_single_policy_defaults(env_generator, policy_args, policy_name="single_agent",
                               agent_name="agent0", policy_type=PPOPolicy):
    env = env_generator()

    def pick(space):
        return space[agent_name] if isinstance(space, dict) else space

    critic_space = getattr(env, "critic_observation_space", env.observation_space)
    policy_settings = {policy_name: (policy_type, pick(env.observation_space), pick(critic_space),
                                     pick(env.action_space), policy_args)}
    return policy_settings, (lambda *args: policy_name)
