"""GPU: DynamicStepDriver semantics on device tensors -- the reference's golden trajectory
(tf_agents/drivers/dynamic_step_driver_test.py:168-199) with the mocks of drivers/test_utils.py
re-expressed on torch device tensors, plus the synthetic env + replay + DQN collect loop."""
import numpy as np
import pytest
import torch

from agents_amd.drivers import dynamic_step_driver
from agents_amd.environments import random_tf_environment, tf_environment
from agents_amd.policies import tf_policy
from agents_amd.replay_buffers import tf_uniform_replay_buffer as rb_lib
from agents_amd.specs import tensor_spec
from agents_amd.trajectories import policy_step
from agents_amd.trajectories import time_step as ts
from oracle import driver as odriver

pytestmark = pytest.mark.gpu


class EnvMock(tf_environment.TFEnvironment):
    """state += action; episode ends at state >= 3; LAST steps reset (test_utils.py:45-98)."""

    def __init__(self, dev, final_state=3):
        obs = tensor_spec.TensorSpec((), torch.int32, "observation")
        super().__init__(ts.time_step_spec(obs),
                         tensor_spec.BoundedTensorSpec((), torch.int32, 1, 2, "action"), 1)
        self.dev, self.final_state, self.state, self.cur = dev, final_state, 0, None

    def _mk(self, step_type, reward, discount):
        return ts.TimeStep(torch.tensor([step_type], dtype=torch.int32, device=self.dev),
                           torch.tensor([reward], dtype=torch.float32, device=self.dev),
                           torch.tensor([discount], dtype=torch.float32, device=self.dev),
                           torch.tensor([self.state], dtype=torch.int32, device=self.dev))

    def _reset(self):
        self.state = 0
        self.cur = self._mk(0, 0.0, 1.0)
        return self.cur

    def _current_time_step(self):
        return self.cur if self.cur is not None else self._reset()

    def _step(self, action):
        if self.cur is None or int(self.cur.step_type) == 2:
            return self._reset()
        self.state += int(action)
        self.cur = self._mk(1, 1.0, 1.0) if self.state < self.final_state else \
            self._mk(2, 1.0, 0.0)
        return self.cur


class PolicyMock(tf_policy.TFPolicy):
    """Alternating actions 1, 2; state resets on FIRST; info = 2*action (test_utils.py:101-167)."""

    def __init__(self, env, dev):
        super().__init__(env.time_step_spec(), env.action_spec(),
                         policy_state_spec=tensor_spec.TensorSpec((), torch.int32),
                         info_spec=env.action_spec())
        self.dev = dev

    def _get_initial_state(self, batch_size):
        return torch.zeros((batch_size,), dtype=torch.int32, device=self.dev)

    def _action(self, time_step, policy_state, seed):
        policy_state = torch.where(time_step.is_first(), torch.zeros_like(policy_state),
                                   policy_state)
        action = (policy_state % 2 + 1).to(torch.int32)
        return policy_step.PolicyStep(action, policy_state + 1, action * 2)


def test_golden_trajectory(dev):
    env = EnvMock(dev)
    pol = PolicyMock(env, dev)
    rb = rb_lib.TFUniformReplayBuffer(pol.trajectory_spec, batch_size=1, max_length=20,
                                      device=dev)
    seen = []
    drv = dynamic_step_driver.DynamicStepDriver(
        env, pol, observers=[rb.add_batch, seen.append], num_steps=6)
    final_ts, _ = drv.run()
    got = rb.gather_all()
    col = lambda t: t.cpu().numpy()[0].tolist()
    assert col(got.step_type) == [0, 1, 2, 0, 1, 2, 0, 1]
    assert col(got.observation) == [0, 1, 3, 0, 1, 3, 0, 1]
    assert col(got.action) == [1, 2, 1, 1, 2, 1, 1, 2]
    assert col(got.policy_info) == [2, 4, 2, 2, 4, 2, 2, 4]
    assert col(got.next_step_type) == [1, 2, 0, 1, 2, 0, 1, 2]
    assert col(got.reward) == [1, 1, 0, 1, 1, 0, 1, 1]
    assert col(got.discount) == [1, 0, 1, 1, 0, 1, 1, 0]
    assert int(final_ts.step_type) == 2 and len(seen) == 8
    # same sequence as the numpy oracle of the loop
    oseen, _, _ = odriver.run_step_driver(odriver.MockEnv(), odriver.MockPolicy(), 6)
    assert [t["action"] for t in oseen] == col(got.action)


def test_run_twice_and_max_iterations(dev):
    env = EnvMock(dev)
    pol = PolicyMock(env, dev)
    seen = []
    drv = dynamic_step_driver.DynamicStepDriver(env, pol, observers=[seen.append], num_steps=1)
    t1, s1 = drv.run()
    t2, s2 = drv.run(t1, s1)
    assert len(seen) == 2 and int(seen[1].action) == 2
    drv2 = dynamic_step_driver.DynamicStepDriver(env, pol, observers=[seen.append],
                                                 num_steps=100)
    n0 = len(seen)
    drv2.run(maximum_iterations=5)
    assert len(seen) - n0 == 5


def test_synthetic_env_collect_into_replay(dev):
    """RandomTFEnvironment -> epsilon-greedy DQN collect policy -> add_batch; boundary rows are
    stored, counted steps exclude them, LAST is always followed by FIRST with reward 0 / discount 1."""
    from agents_amd.agents.dqn import dqn_agent
    from agents_amd.networks import q_network
    B = 64
    obs_spec = tensor_spec.BoundedTensorSpec((4,), torch.float32, -4.0, 4.0)
    aspec = tensor_spec.BoundedTensorSpec((), torch.int64, 0, 1)
    tss = ts.time_step_spec(obs_spec)
    with torch.cuda.device(dev):
        env = random_tf_environment.RandomTFEnvironment(tss, aspec, batch_size=B,
                                                        episode_end_probability=0.2, seed=4,
                                                        device=dev)
        net = q_network.QNetwork(obs_spec, aspec, fc_layer_params=(100,), seed=1)
        agent = dqn_agent.DqnAgent(tss, aspec, q_network=net, optimizer=None, epsilon_greedy=0.3)
        agent.initialize()
        rb = rb_lib.TFUniformReplayBuffer(agent.collect_data_spec, batch_size=B, max_length=64,
                                          device=dev)
        drv = dynamic_step_driver.DynamicStepDriver(env, agent.collect_policy,
                                                    observers=[rb.add_batch], num_steps=B * 20)
        drv.run()
        data = rb.gather_all()
    st = data.step_type.cpu().numpy()
    nst = data.next_step_type.cpu().numpy()
    n = st.shape[1]
    assert n >= 20 and np.all(st[:, 1:] == nst[:, :-1])
    assert (st != 2).sum() >= B * 20 and (st[:, :-1] != 2).sum() < B * 20  # "never less"
    rew, disc = data.reward.cpu().numpy(), data.discount.cpu().numpy()
    assert np.all(nst[st == 2] == 0) and np.all(rew[st == 2] == 0) and np.all(disc[st == 2] == 1)
    assert np.all(disc[nst == 2] == 0) and np.all(disc[nst == 1] == 1)
    assert (st == 2).any() and st[:, 0].tolist() == [0] * B
    obs = data.observation.cpu().numpy()
    assert obs.min() >= -4.0 and obs.max() < 4.0 and abs(obs.mean()) < 0.2
    act = data.action.cpu().numpy()
    assert set(np.unique(act)) <= {0, 1}
