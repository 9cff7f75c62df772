"""Driver base: holds the environment, the policy and the three observer lists that concrete drivers
feed (role of tf_agents/drivers/driver.py:28-85; `run` is left to DynamicStepDriver /
DynamicEpisodeDriver)."""
import abc


def _as_list(callables):
    return list(callables) if callables else []


class Driver(abc.ABC):
    """`observers` receive every Trajectory, `transition_observers` every
    (time_step, policy_step, next_time_step) triple, `info_observers` the environment info."""

    _ROLES = ("observers", "transition_observers", "info_observers")

    def __init__(self, env, policy, observers=None, transition_observers=None,
                 info_observers=None):
        self._env, self._policy = env, policy
        given = (observers, transition_observers, info_observers)
        for role, callables in zip(self._ROLES, given):
            setattr(self, "_" + role, _as_list(callables))

    env = property(lambda self: self._env)
    policy = property(lambda self: self._policy)
    observers = property(lambda self: self._observers)
    transition_observers = property(lambda self: self._transition_observers)
    info_observers = property(lambda self: self._info_observers)

    @abc.abstractmethod
    def run(self, *args, **kwargs):
        """Steps the environment with the policy and notifies the observers."""
