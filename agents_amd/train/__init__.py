from agents_amd.train import learner, ppo_learner  # noqa: F401
from agents_amd.train.utils import strategy_utils  # noqa: F401
