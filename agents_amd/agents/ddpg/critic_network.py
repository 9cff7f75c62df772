"""tf_agents/agents/ddpg/critic_network.py under its reference import path (the SAC script imports
`tf_agents.agents.ddpg.critic_network`); the implementation lives in networks/critic_network.py."""
from agents_amd.networks.critic_network import CriticNetwork  # noqa: F401
