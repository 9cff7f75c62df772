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
            for observer in self._observers:
                observer(traj)
            for observer in self._transition_observers:
                observer((time_step, action_step, next_time_step))
            upper += self._count(traj.step_type)
            time_step = next_time_step
            iterations += 1
        return time_step, policy_state
