"""PPO: ppo_agent.PPOAgent, ppo_clip_agent.PPOClipAgent, ppo_policy.PPOPolicy,
ppo_actor_network.PPOActorNetwork, value networks (tf_agents/agents/ppo/)."""
