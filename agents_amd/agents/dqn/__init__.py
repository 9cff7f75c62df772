from agents_amd.agents.dqn import dqn_agent  # noqa: F401
