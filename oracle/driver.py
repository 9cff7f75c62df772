"""numpy restatement of DynamicStepDriver's loop (TEST INFRASTRUCTURE).

tf_agents/drivers/dynamic_step_driver.py:100-224: loop while sum(counter) < num_steps:
action = policy.action(time_step, state); next = env.step(action);
traj = from_transition(...)  (tf_agents/trajectories/trajectory.py:614-647);
observers(traj); counter += (traj.step_type != LAST).
Mocks mirror tf_agents/drivers/test_utils.py:45-167 (PyEnvironmentMock, TFPolicyMock).
"""
import numpy as np

FIRST, MID, LAST = 0, 1, 2


class MockEnv:
    """state += action; episode ends when state >= final_state; LAST steps auto-reset
    (test_utils.py:45-98; py_environment.py:233-239).  Batch of 1."""

    def __init__(self, final_state=3):
        self.final_state = final_state
        self.state = 0
        self.cur = None

    def reset(self):
        self.state = 0
        self.cur = dict(step_type=FIRST, reward=0.0, discount=1.0, observation=0)
        return self.cur

    def current_time_step(self):
        return self.cur if self.cur is not None else self.reset()

    def step(self, action):
        if self.cur is None or self.cur["step_type"] == LAST:
            return self.reset()
        if action < 1 or action > 2:
            raise ValueError("action out of range")
        self.state += int(action)
        if self.state < self.final_state:
            self.cur = dict(step_type=MID, reward=1.0, discount=1.0, observation=self.state)
        else:
            self.cur = dict(step_type=LAST, reward=1.0, discount=0.0, observation=self.state)
        return self.cur


class MockPolicy:
    """Alternates actions 1,2; state resets on FIRST; info = 2*action (test_utils.py:101-167)."""

    def initial_state(self):
        return 0

    def action(self, time_step, state):
        if time_step["step_type"] == FIRST:
            state = 0
        action = state % 2 + 1
        return action, state + 1, action * 2


def run_step_driver(env, policy, num_steps, time_step=None, policy_state=None,
                    maximum_iterations=None):
    """Returns (list of trajectory dicts seen by observers, final time_step, final state)."""
    if time_step is None:
        time_step = env.current_time_step()
    if policy_state is None:
        policy_state = policy.initial_state()
    counter = 0
    seen = []
    it = 0
    while counter < num_steps and (maximum_iterations is None or it < maximum_iterations):
        action, policy_state, info = policy.action(time_step, policy_state)
        nxt = env.step(action)
        traj = dict(step_type=time_step["step_type"], observation=time_step["observation"],
                    action=action, policy_info=info, next_step_type=nxt["step_type"],
                    reward=nxt["reward"], discount=nxt["discount"])
        seen.append(traj)
        counter += int(traj["step_type"] != LAST)
        time_step = nxt
        it += 1
    return seen, time_step, policy_state
