from agents_amd.train.utils import strategy_utils  # noqa: F401
