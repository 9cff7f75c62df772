"""Step-interval trigger used by the Learner (role of tf_agents/train/interval_trigger.py:24-74).

Behaviour contract taken from the reference:
  * a non-positive interval disables the trigger altogether;
  * the callback fires once the observed value is at least `interval` past the value at which it
    last fired (the first reference point is `start`);
  * `force_trigger=True` fires regardless of the distance, except when nothing has advanced since
    the last firing;
  * `reset()` returns the reference point to the constructor's `start`, `set_start(v)` moves it.
"""


class IntervalTrigger:
    def __init__(self, interval, fn, start=0):
        self._every = interval
        # a bound method of the trigger itself (the subclasses in triggers.py) is kept as the plain
        # function: trigger -> bound method -> trigger would be a reference cycle that keeps the
        # checkpointed agent alive until the cyclic collector runs
        self._callback_is_own = getattr(fn, "__self__", None) is self
        self._callback = fn.__func__ if self._callback_is_own else fn
        self._start0 = start
        self._fired_at = start

    @property
    def enabled(self):
        return self._every > 0

    def _due(self, value, forced):
        advanced = value != self._fired_at
        far_enough = value - self._fired_at >= self._every
        return far_enough or (forced and advanced)

    def __call__(self, value, force_trigger=False):
        if self.enabled and self._due(value, force_trigger):
            self._fired_at = value
            if self._callback_is_own:
                self._callback(self)
            else:
                self._callback()

    def reset(self):
        self._fired_at = self._start0

    def set_start(self, start):
        self._fired_at = start
