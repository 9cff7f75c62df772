from agents_amd.environments import random_tf_environment, tf_environment  # noqa: F401
