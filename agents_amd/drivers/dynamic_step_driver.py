"""DynamicStepDriver: steps a batched environment with a policy until `num_steps` non-boundary
transitions have been collected, feeding every trajectory to the observers.

Same contract as tf_agents/drivers/dynamic_step_driver.py:48-224:
  loop while sum(counter) < num_steps (:113)
    action_step = policy.action(time_step, policy_state)                      (:134)
    next_time_step = env.step(action_step.action)                             (:136)
    traj = trajectory.from_transition(time_step, action_step, next_time_step) (:147)
    observers(traj); transition_observers((time_step, action_step, next))     (:150-163)
    counter += ~traj.is_boundary()                                            (:170)
so with B envs it "may take more steps than num_steps but never less" (:57-60), and boundary rows
(step_type == LAST) are shown to the observers but not counted.

The reference's tf.while_loop evaluates its condition in-graph; here the counter lives on the
device (aa_count_steps) and the host reads it only when the loop could terminate: an iteration
adds at most B to the counter, so the first ceil(num_steps / B) iterations are enqueued without any
device->host synchronisation.
"""
import torch

from agents_amd import _lib
from agents_amd.drivers import driver
from agents_amd.trajectories import trajectory
from agents_amd.utils import nest_utils


def is_bandit_env(env):
    return False


# A/B aid: False keeps the step counter a launch of its own in the eager loop
COUNT_IN_ADD = True


class DynamicStepDriver(driver.Driver):
    def __init__(self, env, policy, observers=None, transition_observers=None, num_steps=1):
        super().__init__(env, policy, observers, transition_observers)
        self._num_steps = num_steps
        self._total = None
        self._counter = None

    def _count(self, step_type):
        """counter[b] += (step_type != LAST); returns nothing (device-side)."""
        lib = _lib.load()
        st = step_type if step_type.dim() > 0 else step_type.reshape(1)
        if st.dtype != torch.int32:
            st = st.to(torch.int32)
        st = st.contiguous()
        _lib.require_cuda(st)
        B = st.numel()
        if self._total is None or self._total.device != st.device:
            self._total = torch.zeros((1,), dtype=torch.int64, device=st.device)
        if self._counter is None or self._counter.numel() != B or \
                self._counter.device != st.device:
            self._counter = torch.zeros((B,), dtype=torch.int32, device=st.device)
        with torch.cuda.device(st.device):
            _lib.check(lib.aa_count_steps(st.data_ptr(), B, self._counter.data_ptr(),
                                          self._total.data_ptr(), None, _lib.stream_ptr()),
                       "aa_count_steps")
        return B

    def run(self, time_step=None, policy_state=None, maximum_iterations=None):
        """Returns (final time_step, final policy_state)."""
        if time_step is None:
            time_step = self.env.current_time_step()
        if policy_state is None:
            policy_state = self.policy.get_initial_state(self.env.batch_size)
        if self._total is not None:
            self._total.zero_()
            self._counter.zero_()
        iterations = 0
        upper = 0  # host-known upper bound of the device counter
        while maximum_iterations is None or iterations < maximum_iterations:
            if upper >= self._num_steps:
                # the loop might be done: read the device counter (the only sync in the loop)
                if self._total is None or int(self._total.item()) >= self._num_steps:
                    break
            action_step = self.policy.action(time_step, policy_state)
            policy_state = action_step.state
            next_time_step = self.env.step(action_step.action)
            traj = trajectory.from_transition(time_step, action_step, next_time_step)
            # the step count rides in the replay buffer's add_batch launch when the first observer
            # is one that offers it (csrc/replay.hip: aa_rb_scatter_rows_count; same counter values,
            # one launch less per loop body -- the eager loop is bound by its launches)
            fused = self._counting_observer(traj.step_type) if COUNT_IN_ADD else None
            for k, observer in enumerate(self._observers):
                if k == 0 and fused is not None:
                    fused.add_batch_counting(traj, traj.step_type, self._counter, self._total, None)
                else:
                    observer(traj)
            for observer in self._transition_observers:
                observer((time_step, action_step, next_time_step))
            upper += traj.step_type.numel() if fused is not None else self._count(traj.step_type)
            time_step = next_time_step
            iterations += 1
        return time_step, policy_state

    def _counting_observer(self, step_type):
        """The replay buffer behind observers[0] if its add can also run this loop's counter on
        `step_type` (int32 [B] on the device), with the counter tensors made ready; else None."""
        if not self._observers:
            return None
        obs0 = self._observers[0]
        owner = getattr(obs0, "__self__", None)
        if getattr(obs0, "__name__", "") != "add_batch" or \
                not getattr(owner, "supports_counting_add", lambda: False)() or \
                step_type.dim() != 1 or step_type.dtype != torch.int32 or not step_type.is_cuda or \
                not step_type.is_contiguous():
            return None
        B = step_type.numel()
        if self._total is None or self._total.device != step_type.device:
            self._total = torch.zeros((1,), dtype=torch.int64, device=step_type.device)
        if self._counter is None or self._counter.numel() != B or \
                self._counter.device != step_type.device:
            self._counter = torch.zeros((B,), dtype=torch.int32, device=step_type.device)
        return owner
