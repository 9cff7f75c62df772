"""Agents: tf_agent.TFAgent base, dqn.dqn_agent.DqnAgent/DdqnAgent, ppo.*."""
from agents_amd.agents import tf_agent  # noqa: F401
from agents_amd.agents.tf_agent import LossInfo, TFAgent  # noqa: F401
