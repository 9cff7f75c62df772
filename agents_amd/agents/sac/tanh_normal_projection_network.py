"""tf_agents/agents/sac/tanh_normal_projection_network.py under its reference import path; the
implementation lives beside the actor network (networks/actor_distribution_network.py)."""
from agents_amd.networks.actor_distribution_network import (  # noqa: F401
    TanhNormalProjectionNetwork, std_clip_transform)
