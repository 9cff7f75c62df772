"""DynamicEpisodeDriver: steps a batched environment until `num_episodes` episode boundaries have
been seen (summed over the batch), feeding every trajectory to the observers.

Same contract as tf_agents/drivers/dynamic_episode_driver.py:45-259:
  loop while sum(counter) < num_episodes                                   (:113-129)
    action_step = policy.action(time_step, policy_state); next = env.step(action)
    traj = from_transition(...); observers(traj)
    counter += traj.is_boundary()            (step_type == LAST)            (:176)
  run(time_step=None -> env.reset(), policy_state=None, num_episodes=None,
      maximum_iterations=None)                                              (:181-259)
The boundary count lives on the device: every iteration adds `step_type != LAST` to a device
counter (aa_count_steps), so boundaries = iterations * B - counter.  The host reads the counter only
when the loop could be over, and after a read knows how many more iterations are needed at least
((num_episodes - seen + B - 1) // B), which it enqueues without synchronising.
"""
import torch

from agents_amd import _lib
from agents_amd.drivers import driver
from agents_amd.trajectories import trajectory


class DynamicEpisodeDriver(driver.Driver):
    def __init__(self, env, policy, observers=None, transition_observers=None, num_episodes=1):
        super().__init__(env, policy, observers, transition_observers)
        self._num_episodes = num_episodes
        self._total = None

    def _count_non_boundary(self, step_type):
        lib = _lib.load()
        st = step_type if step_type.dim() > 0 else step_type.reshape(1)
        if st.dtype != torch.int32:
            st = st.to(torch.int32)
        st = st.contiguous()
        _lib.require_cuda(st)
        if self._total is None or self._total.device != st.device:
            self._total = torch.zeros((1,), dtype=torch.int64, device=st.device)
        with torch.cuda.device(st.device):
            _lib.check(lib.aa_count_steps(st.data_ptr(), st.numel(), None,
                                          self._total.data_ptr(), None, _lib.stream_ptr()),
                       "aa_count_steps")
        return st.numel()

    def run(self, time_step=None, policy_state=None, num_episodes=None, maximum_iterations=None):
        """Returns (final time_step, final policy_state)."""
        if time_step is None:
            time_step = self.env.reset()
        if policy_state is None:
            policy_state = self.policy.get_initial_state(self.env.batch_size)
        num_episodes = num_episodes or self._num_episodes
        if self._total is not None:
            self._total.zero_()
        iterations = 0
        seen_rows = 0          # rows shown to the counter so far
        next_check = 0         # iteration index before which the loop cannot be finished
        B = None
        while maximum_iterations is None or iterations < maximum_iterations:
            if iterations >= next_check and B is not None:
                done = seen_rows - int(self._total.item())   # the only sync in the loop
                if done >= num_episodes:
                    break
                next_check = iterations + (num_episodes - done + B - 1) // B
            action_step = self.policy.action(time_step, policy_state)
            policy_state = action_step.state
            next_time_step = self.env.step(action_step.action)
            traj = trajectory.from_transition(time_step, action_step, next_time_step)
            for observer in self._observers:
                observer(traj)
            for observer in self._transition_observers:
                observer((time_step, action_step, next_time_step))
            n = self._count_non_boundary(traj.step_type)
            if B is None:
                B = n
                next_check = (num_episodes + B - 1) // B
            seen_rows += n
            time_step = next_time_step
            iterations += 1
        return time_step, policy_state
