"""IntervalTrigger: calls `fn` when the step value has advanced by at least `interval` since the
last trigger.  Same behaviour as tf_agents/train/interval_trigger.py:24-74 (interval <= 0 never
triggers; `force_trigger` fires unless the value equals the last trigger value)."""


class IntervalTrigger:
    def __init__(self, interval, fn, start=0):
        self._interval = interval
        self._original_start_value = start
        self._last_trigger_value = start
        self._fn = fn

    def __call__(self, value, force_trigger=False):
        if self._interval <= 0:
            return
        if (force_trigger and value != self._last_trigger_value) or \
                (value >= self._last_trigger_value + self._interval):
            self._last_trigger_value = value
            self._fn()

    def reset(self):
        self._last_trigger_value = self._original_start_value

    def set_start(self, start):
        self._last_trigger_value = start
