"""TFEnvironment contract (tf_agents/environments/tf_environment.py:37-58): batched
`current_time_step()`, `reset()`, `step(action)` returning TimeStep nests of device tensors;
stepping an env whose current step is LAST ignores the action and resets it."""
import abc

from agents_amd.trajectories import time_step as ts


class TFEnvironment(abc.ABC):
    def __init__(self, time_step_spec=None, action_spec=None, batch_size=1):
        self._time_step_spec = time_step_spec
        self._action_spec = action_spec
        self._batch_size = batch_size

    def time_step_spec(self):
        return self._time_step_spec

    def action_spec(self):
        return self._action_spec

    def observation_spec(self):
        return self._time_step_spec.observation

    def reward_spec(self):
        return self._time_step_spec.reward

    @property
    def batched(self):
        return True

    @property
    def batch_size(self):
        return self._batch_size

    def current_time_step(self):
        return self._current_time_step()

    def reset(self):
        return self._reset()

    def step(self, action):
        return self._step(action)

    @abc.abstractmethod
    def _current_time_step(self):
        ...

    @abc.abstractmethod
    def _reset(self):
        ...

    @abc.abstractmethod
    def _step(self, action):
        ...
