"""PolicyStep(action, state, info); tf_agents/trajectories/policy_step.py:31-78."""
import collections


class PolicyStep(collections.namedtuple("PolicyStep", ("action", "state", "info"))):
    __slots__ = ()

    def __new__(cls, action=(), state=(), info=()):
        return super().__new__(cls, action, state, info)

    def replace(self, **kwargs):
        return self._replace(**kwargs)


class CommonFields:
    LOG_PROBABILITY = "log_probability"


def get_log_probability(info):
    if isinstance(info, dict):
        return info[CommonFields.LOG_PROBABILITY]
    return getattr(info, CommonFields.LOG_PROBABILITY)
