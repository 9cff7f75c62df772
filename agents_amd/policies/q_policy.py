"""Q-value policies: QPolicy, GreedyPolicy, EpsilonGreedyPolicy, RandomTFPolicy.

  QPolicy._distribution          tf_agents/policies/q_policy.py:150-194   (Categorical(logits=Q))
  GreedyPolicy                   tf_agents/policies/greedy_policy.py:70-89 (mode = first arg-max)
  EpsilonGreedyPolicy._action    tf_agents/policies/epsilon_greedy_policy.py:120-143
  RandomTFPolicy                 tf_agents/policies/random_tf_policy.py
The three are fused into one kernel launch after the Q-network forward: arg-max with the action
mask (masked logits -> dtype.min), a Philox uniform per env, and the epsilon mix
(csrc/rollout.hip: aa_eps_greedy_action).
"""
import numpy as np
import torch

from agents_amd import _lib
from agents_amd.policies import tf_policy
from agents_amd.trajectories import policy_step
from agents_amd.utils import graph, nest_utils


def _action_bounds(action_spec):
    spec = nest_utils.flatten(action_spec)[0]
    lo = int(np.asarray(spec.minimum).reshape(-1)[0])
    hi = int(np.asarray(spec.maximum).reshape(-1)[0])
    return spec, lo, hi


class Categorical:
    """What `policy.distribution(time_step).action` exposes of tfp Categorical: `logits` [B, A]
    and `mode()` (first arg-max, greedy_policy.py:70-89)."""

    def __init__(self, logits, policy):
        self.logits = logits
        self._policy = policy

    def mode(self):
        return self._policy.select(self.logits, None, 0.0)


def distribution_of(policy, time_step, temperature):
    """Categorical(logits = masked Q / temperature) of a discrete Q policy."""
    obs = time_step.observation
    mask = None
    if policy._observation_and_action_constraint_splitter is not None:
        obs, mask = policy._observation_and_action_constraint_splitter(obs)
    batched = time_step.step_type.dim() > 0
    if not batched:
        obs = nest_utils.map_structure(lambda t: t.unsqueeze(0), obs)
        mask = None if mask is None else mask.unsqueeze(0)
    B = nest_utils.flatten(obs)[0].shape[0]
    dev = nest_utils.flatten(obs)[0].device
    graph.join_lanes(dev)
    with torch.cuda.device(dev):
        q = policy._q_values(obs, B, dev)
        logits = torch.empty_like(q)
        if mask is not None:
            mask = mask.to(torch.int32).contiguous()
        _lib.check(_lib.load().aa_boltzmann_action(
            q.data_ptr(), None if mask is None else mask.data_ptr(), B, policy._num_actions,
            float(temperature), None, 0, None, None, policy._lo, None, 0, logits.data_ptr(), 0,
            _lib.stream_ptr()), "aa_boltzmann_action")
    return policy_step.PolicyStep(Categorical(logits, policy), (), ())


class _DiscretePolicy(tf_policy.TFPolicy):
    """Shared machinery: run (optional) Q-network, then the fused select kernel."""

    def __init__(self, time_step_spec, action_spec, q_network=None, epsilon=0.0, seed=0,
                 observation_and_action_constraint_splitter=None, emit_log_probability=False,
                 name=None):
        super().__init__(time_step_spec, action_spec, emit_log_probability=emit_log_probability,
                         observation_and_action_constraint_splitter=
                         observation_and_action_constraint_splitter, name=name)
        flat = nest_utils.flatten(action_spec)
        if len(flat) != 1:
            raise ValueError("Only scalar actions are supported now.")
        self._spec, self._lo, self._hi = _action_bounds(action_spec)
        if self._spec.shape not in [(), (1,)]:
            raise ValueError("Only scalar actions are supported now.")
        if self._spec.dtype not in (torch.int32, torch.int64):
            raise ValueError("discrete policies need an int32/int64 action spec")
        self._num_actions = self._hi - self._lo + 1
        self._q_network = q_network
        self._epsilon = epsilon
        self._seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._call_counter = None
        self._eps_dev = None     # device copy of a callable (decaying) epsilon
        self._eps_host = None
        self._zero_q = {}
        self._slot = "policy"

    def _variables(self):
        return self._q_network.variables if self._q_network is not None else []

    def _distribution(self, time_step, policy_state):
        """Categorical over the (masked) Q values (q_policy.py:150-194)."""
        return distribution_of(self, time_step, 1.0)

    @property
    def q_network(self):
        return self._q_network

    def _get_epsilon(self):
        e = self._epsilon
        return float(e() if callable(e) else e)

    def _q_values(self, observation, B, device):
        if self._q_network is None:
            z = self._zero_q.get((B, device))
            if z is None:
                z = torch.zeros((B, self._num_actions), dtype=torch.float32, device=device)
                self._zero_q[(B, device)] = z
            return z
        return self._q_network.forward(observation, slot=self._slot)

    def q_values(self, time_step):
        """[B, A] Q table for a batch of time steps (network-owned buffer)."""
        obs = time_step.observation
        if self._observation_and_action_constraint_splitter is not None:
            obs, _ = self._observation_and_action_constraint_splitter(obs)
        return self._q_values(obs, obs.shape[0], obs.device)

    def _counter(self, dev):
        if self._call_counter is None:
            # [Philox call counter, arrival count of the select kernel's workgroups]
            self._call_counter = torch.tensor([getattr(self, "_pending_counter", 0), 0],
                                              dtype=torch.int64).to(dev)
        return self._call_counter

    def select(self, q, mask, epsilon, out=None):
        """Fused masked arg-max + epsilon mix on a [B, A] Q table -> actions [B]."""
        lib = _lib.load()
        _lib.require_cuda(q)
        B = q.shape[0]
        dev = q.device
        self._counter(dev)
        if out is None:
            out = torch.empty((B,) + tuple(self._spec.shape), dtype=self._spec.dtype, device=dev)
        if mask is not None:
            mask = mask.to(torch.int32).contiguous() if mask.dtype != torch.int32 else \
                mask.contiguous()
        st = _lib.stream_ptr()
        eps_ptr = None
        if callable(self._epsilon):
            # a schedule: the kernel reads epsilon from device memory, refreshed on the host
            # before every launch / graph replay (a by-value argument would be frozen in a graph)
            if self._eps_dev is None:
                self._eps_dev = torch.zeros((1,), dtype=torch.float32, device=dev)
            graph.on_replay(self._refresh_epsilon)
            eps_ptr = self._eps_dev.data_ptr()
        advance = epsilon > 0 or eps_ptr is not None   # the stream moves only when it is consumed
        _lib.check(lib.aa_eps_greedy_action(
            q.data_ptr(), None if mask is None else mask.data_ptr(), B, self._num_actions,
            float(epsilon), eps_ptr, self._seed, self._call_counter.data_ptr(),
            self._call_counter[1:].data_ptr() if advance else None, self._lo,
            out.data_ptr(), 1 if self._spec.dtype == torch.int64 else 0, st),
            "aa_eps_greedy_action")
        return out

    def state_dict(self):
        return {"call_counter": None if self._call_counter is None
                else int(self._call_counter[0].item())}

    def load_state_dict(self, sd):
        v = sd.get("call_counter")
        if v is None:
            return
        if self._call_counter is None:
            self._pending_counter = int(v)      # applied when the counter tensor is created
        else:
            self._call_counter[0] = int(v)

    def _refresh_epsilon(self):
        e = self._get_epsilon()
        if e != self._eps_host:
            self._eps_host = e
            self._eps_dev.fill_(e)

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
            eps = self._get_epsilon()
            actions = self.select(self._q_values(obs, B, dev), mask, eps)
        if not batched:
            actions = actions.squeeze(0)
        return policy_step.PolicyStep(actions, policy_state, ())


class QPolicy(_DiscretePolicy):
    """Greedy-capable Q policy; `action()` here samples the arg-max (the reference's QPolicy
    samples Categorical(logits=Q); agents only use it wrapped in Greedy / EpsilonGreedy)."""

    def __init__(self, time_step_spec, action_spec, q_network, emit_log_probability=False,
                 observation_and_action_constraint_splitter=None, validate_action_spec=True,
                 name=None, seed=0):
        if validate_action_spec:
            _, lo, _ = _action_bounds(action_spec)
        # q_policy.py:96-101: the policy creates the network's variables from the (split)
        # observation spec if nobody has yet
        if q_network is not None and not getattr(q_network, "_built", True):
            net_spec = time_step_spec.observation
            if observation_and_action_constraint_splitter is not None:
                net_spec, _ = observation_and_action_constraint_splitter(net_spec)
            q_network.create_variables(net_spec)
        super().__init__(time_step_spec, action_spec, q_network=q_network, epsilon=0.0, seed=seed,
                         observation_and_action_constraint_splitter=
                         observation_and_action_constraint_splitter,
                         emit_log_probability=emit_log_probability, name=name)


class GreedyPolicy(_DiscretePolicy):
    def __init__(self, policy, name=None):
        super().__init__(policy.time_step_spec, policy.action_spec, q_network=policy.q_network,
                         epsilon=0.0, seed=policy._seed,
                         observation_and_action_constraint_splitter=
                         policy.observation_and_action_constraint_splitter, name=name)
        self._wrapped_policy = policy
        self._slot = "greedy"

    @property
    def wrapped_policy(self):
        return self._wrapped_policy


class EpsilonGreedyPolicy(_DiscretePolicy):
    def __init__(self, policy, epsilon, exploration_mask=None, info_fields_to_inherit_from_greedy=
                 (), name=None, seed=None):
        if exploration_mask is not None:
            raise NotImplementedError("exploration_mask is outside the hot-path scope")
        super().__init__(policy.time_step_spec, policy.action_spec, q_network=policy.q_network,
                         epsilon=epsilon, seed=policy._seed if seed is None else seed,
                         observation_and_action_constraint_splitter=
                         policy.observation_and_action_constraint_splitter, name=name)
        self._greedy_policy = GreedyPolicy(policy)
        self._slot = "collect"

    @property
    def wrapped_policy(self):
        return self._greedy_policy.wrapped_policy


class RandomTFPolicy(_DiscretePolicy):
    """Uniform random discrete actions (respecting the action mask)."""

    def __init__(self, time_step_spec, action_spec, *args, **kwargs):
        splitter = kwargs.pop("observation_and_action_constraint_splitter", None)
        seed = kwargs.pop("seed", 12345)
        super().__init__(time_step_spec, action_spec, q_network=None, epsilon=1.0, seed=seed,
                         observation_and_action_constraint_splitter=splitter,
                         name=kwargs.get("name"))
