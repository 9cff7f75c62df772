"""Policies (tf_policy.TFPolicy, the Q-value policy family, and the reference's module names for
the wrappers the train_eval scripts import)."""
from agents_amd.policies import q_policy, tf_policy  # noqa: F401
from agents_amd.policies import q_policy as epsilon_greedy_policy  # noqa: F401
from agents_amd.policies import greedy_policy, random_tf_policy  # noqa: F401
