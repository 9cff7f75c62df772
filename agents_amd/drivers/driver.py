"""Driver base class (tf_agents/drivers/driver.py:28-85)."""
import abc


class Driver(abc.ABC):
    def __init__(self, env, policy, observers=None, transition_observers=None,
                 info_observers=None):
        self._env = env
        self._policy = policy
        self._observers = observers or []
        self._transition_observers = transition_observers or []
        self._info_observers = info_observers or []

    @property
    def env(self):
        return self._env

    @property
    def policy(self):
        return self._policy

    @property
    def transition_observers(self):
        return self._transition_observers

    @property
    def observers(self):
        return self._observers

    @property
    def info_observers(self):
        return self._info_observers

    @abc.abstractmethod
    def run(self):
        ...
