from agents_amd.train import learner  # noqa: F401
from agents_amd.train.utils import strategy_utils  # noqa: F401
