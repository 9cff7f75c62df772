"""BoltzmannPolicy: samples the wrapped Q policy's Categorical at a temperature.

  BoltzmannPolicy                 tf_agents/policies/boltzmann_policy.py:30-101
      _apply_temperature :83-86   logits = dist.logits / temperature
      _distribution      :88-101  the wrapped policy's distribution with those logits
  QPolicy._distribution           tf_agents/policies/q_policy.py:150-194 (masked logits -> dtype.min)
  DqnAgent's collect policy when `boltzmann_temperature` is given
                                  tf_agents/agents/dqn/dqn_agent.py:357-360

One launch after the Q-network forward (csrc/rollout.hip: aa_boltzmann_action): division by the
temperature, mask, softmax CDF in float64, one Philox uniform per row, inverse-CDF pick.  The random
stream is ours (the reference samples with an unseeded TFP Categorical); given the uniforms the
actions are those of oracle/policy.py bit for bit.
"""
import torch

from agents_amd import _lib
from agents_amd.policies import q_policy
from agents_amd.trajectories import policy_step
from agents_amd.utils import graph, nest_utils


class BoltzmannPolicy(q_policy._DiscretePolicy):
    def __init__(self, policy, temperature=1.0, name=None, seed=None):
        if not isinstance(policy, q_policy._DiscretePolicy) or policy.q_network is None:
            raise ValueError("BoltzmannPolicy wraps a policy whose distribution is parameterized "
                             "by logits (a QPolicy)")
        super().__init__(policy.time_step_spec, policy.action_spec, q_network=policy.q_network,
                         epsilon=0.0, seed=policy._seed + 1 if seed is None else seed,
                         observation_and_action_constraint_splitter=
                         policy.observation_and_action_constraint_splitter,
                         emit_log_probability=policy.emit_log_probability, name=name)
        self._temperature = temperature
        self._wrapped_policy = policy
        self._slot = "collect"
        self._temp_dev = None
        self._temp_host = None

    @property
    def wrapped_policy(self):
        return self._wrapped_policy

    def _get_temperature_value(self):
        t = self._temperature
        return float(t() if callable(t) else t)

    def _refresh_temperature(self):
        t = self._get_temperature_value()
        if t != self._temp_host:
            self._temp_host = t
            self._temp_dev.fill_(t)

    def _distribution(self, time_step, policy_state):
        if self._temperature is None:
            return self._wrapped_policy.distribution(time_step, policy_state)
        return q_policy.distribution_of(self, time_step, self._get_temperature_value())

    def sample(self, q, mask, out=None):
        """Boltzmann actions [B] for a [B, A] Q table (one launch; advances the policy's stream)."""
        lib = _lib.load()
        _lib.require_cuda(q)
        B, dev = q.shape[0], q.device
        self._counter(dev)
        if out is None:
            out = torch.empty((B,) + tuple(self._spec.shape), dtype=self._spec.dtype, device=dev)
        if mask is not None:
            mask = mask.to(torch.int32).contiguous()
        t_ptr = None
        if callable(self._temperature):
            # a schedule: read from device memory, refreshed on the host before every launch /
            # graph replay (a by-value argument would be frozen in a captured graph)
            if self._temp_dev is None:
                self._temp_dev = torch.ones((1,), dtype=torch.float32, device=dev)
            graph.on_replay(self._refresh_temperature)
            t_ptr = self._temp_dev.data_ptr()
        _lib.check(lib.aa_boltzmann_action(
            q.data_ptr(), None if mask is None else mask.data_ptr(), B, self._num_actions,
            1.0 if t_ptr is not None else self._get_temperature_value(), t_ptr, self._seed,
            self._call_counter.data_ptr(), self._call_counter[1:].data_ptr(), self._lo,
            out.data_ptr(), 1 if self._spec.dtype == torch.int64 else 0, None, 1,
            _lib.stream_ptr()), "aa_boltzmann_action")
        return out

    def _action(self, time_step, policy_state, seed):
        obs = time_step.observation
        mask = None
        if self._observation_and_action_constraint_splitter is not None:
            obs, mask = self._observation_and_action_constraint_splitter(obs)
        batched = time_step.step_type.dim() > 0
        if not batched:
            obs = nest_utils.map_structure(lambda t: t.unsqueeze(0), obs)
            mask = None if mask is None else mask.unsqueeze(0)
        B = nest_utils.flatten(obs)[0].shape[0]
        dev = nest_utils.flatten(obs)[0].device
        graph.join_lanes(dev)
        with torch.cuda.device(dev):
            actions = self.sample(self._q_values(obs, B, dev), mask)
        if not batched:
            actions = actions.squeeze(0)
        return policy_step.PolicyStep(actions, policy_state, ())
