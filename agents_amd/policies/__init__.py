"""Policies (tf_policy.TFPolicy and the Q-value policy family)."""
from agents_amd.policies import q_policy, tf_policy  # noqa: F401
from agents_amd.policies import q_policy as greedy_policy  # noqa: F401
from agents_amd.policies import q_policy as epsilon_greedy_policy  # noqa: F401
from agents_amd.policies import q_policy as random_tf_policy  # noqa: F401
