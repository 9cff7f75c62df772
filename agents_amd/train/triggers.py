"""Learner triggers (hot-path subset of tf_agents/train/triggers.py:40-264).

  StepPerSecondLogTrigger  (:233-264)  logs train steps/s every `interval` steps
  CheckpointTrigger                    saves a `common.Checkpointer` every `interval` steps -- the
                                       reference's Learner owns this internally
                                       (learner.py:206-243); exposed so scripts can checkpoint the
                                       replay buffer / environment next to the agent
PolicySavedModelTrigger / ReverbCheckpointTrigger export TF SavedModels / Reverb tables and are out
of scope (the on-disk format here is torch.save, not TF checkpoints).
"""
import time

from agents_amd.train import interval_trigger


class StepPerSecondLogTrigger(interval_trigger.IntervalTrigger):
    def __init__(self, train_step, interval, log_fn=print):
        self._train_step = train_step
        self._log_fn = log_fn
        self._t0 = time.perf_counter()
        self._s0 = int(train_step)
        self.last_steps_per_sec = None
        super().__init__(interval, self._log)

    def _log(self):
        now, s = time.perf_counter(), int(self._train_step)
        if now > self._t0 and s > self._s0:
            self.last_steps_per_sec = (s - self._s0) / (now - self._t0)
            self._log_fn("Step: %d, %.3f steps/sec" % (s, self.last_steps_per_sec))
        self._t0, self._s0 = now, s


class CheckpointTrigger(interval_trigger.IntervalTrigger):
    def __init__(self, checkpointer, train_step, interval):
        self._checkpointer = checkpointer
        self._train_step = train_step
        super().__init__(interval, self._save, start=int(train_step))

    def _save(self):
        self._checkpointer.save(int(self._train_step))
