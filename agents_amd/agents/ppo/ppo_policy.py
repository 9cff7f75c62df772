"""PPOPolicy: samples actions from the actor network's diagonal Normal and records the
distribution parameters (and, in collect mode with precomputed values, the value prediction).

  tf_agents/agents/ppo/ppo_policy.py:21-240   (ActorPolicy subclass; info = {'dist_params': ...,
                                               'value_prediction': ...})
  tf_agents/policies/actor_policy.py          (distribution -> sample, optional clip to spec)
  tf_agents/policies/greedy_policy.py:70-89   (mode of the distribution = loc)
Sampling is `aa_normal_sample` (Box-Muller on the package's Philox stream, csrc/ppo.hip).
"""
import inspect
import os

import numpy as np
import torch

from agents_amd import _lib
from agents_amd.policies import tf_policy
from agents_amd.specs import tensor_spec
from agents_amd.trajectories import policy_step
from agents_amd.utils import nest_utils

# AA_PPO_FUSE_SAMPLE=0: head, draw and counter advance of a policy step stay three launches and the
# policy info is copied out of the networks' buffers (A/B measurements; bit-identical either way)
_FUSE_SAMPLE = os.environ.get("AA_PPO_FUSE_SAMPLE", "1") != "0"


def _value_takes_out(net):
    """The value network's forward can write a caller's tensor (ValueNet of this package)."""
    try:
        return "out" in inspect.signature(net.forward).parameters
    except (TypeError, ValueError):
        return False


class PPOPolicy(tf_policy.TFPolicy):
    def __init__(self, time_step_spec, action_spec, actor_network, value_network,
                 observation_normalizer=None, clip=True, collect=True,
                 compute_value_and_advantage_in_train=False, seed=0, greedy=False, name=None):
        # ppo_policy.py:231-241: both networks see normalizer.normalize(observation)
        self._observation_normalizer = observation_normalizer
        self._actor_network = actor_network
        self._value_network = value_network
        self._collect = collect
        self._clip = clip
        self._greedy = greedy
        self._compute_value_in_train = compute_value_and_advantage_in_train
        actor_network.create_variables(time_step_spec.observation)
        value_network.create_variables(time_step_spec.observation)
        spec = nest_utils.flatten(action_spec)[0]
        self._spec = spec
        self._D = int(np.prod(spec.shape)) if len(spec.shape) else 1
        info_spec = ()
        if collect:
            pspec = tensor_spec.TensorSpec(spec.shape, torch.float32)
            info_spec = {"dist_params": {"loc": pspec, "scale": pspec}}
            if not compute_value_and_advantage_in_train:
                info_spec["value_prediction"] = tensor_spec.TensorSpec((), torch.float32)
        super().__init__(time_step_spec, action_spec, info_spec=info_spec, clip=clip, name=name)
        self._seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._call_counter = None
        self._arrival = None
        self._lo = self._hi = None
        self._value_out_ok = _value_takes_out(value_network)

    def _variables(self):
        var_list = self._actor_network.variables + self._value_network.variables
        if self._observation_normalizer is not None:       # ppo_policy.py:249-254
            var_list = var_list + list(nest_utils.flatten(self._observation_normalizer.variables))
        return var_list

    def _normalized(self, observations):
        if self._observation_normalizer is None:
            return observations
        return self._observation_normalizer.normalize(observations)

    def apply_value_network(self, observations, step_types=None, value_state=None, training=False):
        """[B, T, ...] or [N, ...] observations -> value predictions of the same outer shape."""
        obs_rank = len(self._time_step_spec.observation.shape)
        outer = tuple(observations.shape[:observations.dim() - obs_rank])
        flat = self._normalized(
            observations.reshape((-1,) + tuple(observations.shape[len(outer):])))
        v = self._value_network.forward(flat, slot=("value", training), need_grad=training)
        return v.view(outer), ()

    def get_initial_value_state(self, batch_size=None):
        return ()

    def state_dict(self):
        return {"call_counter": None if self._call_counter is None
                else int(self._call_counter.item())}

    def load_state_dict(self, sd):
        if sd.get("call_counter") is not None:
            if self._call_counter is None:
                self._call_counter = torch.zeros(
                    (1,), dtype=torch.int64, device=self._actor_network.body.flat_params.device)
            self._call_counter.fill_(int(sd["call_counter"]))

    def _action(self, time_step, policy_state, seed):
        lib = _lib.load()
        obs = time_step.observation
        batched = time_step.step_type.dim() > 0
        if not batched:
            obs = obs.unsqueeze(0)
        dev = obs.device
        with torch.cuda.device(dev):
            obs = self._normalized(obs)
            fused = _FUSE_SAMPLE and not self._greedy and \
                hasattr(self._actor_network, "forward_sample")
            if fused:
                # head + draw + clip + counter in one launch, loc / scale written where the policy
                # info wants them (csrc/ppo.hip: aa_ppo_head_forward_sample; bit-identical to the
                # launches below)
                if self._call_counter is None:
                    self._call_counter = torch.zeros((1,), dtype=torch.int64, device=dev)
                if self._arrival is None:
                    self._arrival = torch.zeros((1,), dtype=torch.int64, device=dev)
                if self._clip and self._lo is None:
                    self._lo = torch.as_tensor(np.broadcast_to(
                        np.asarray(self._spec.minimum, np.float32), (self._D,)).copy(), device=dev)
                    self._hi = torch.as_tensor(np.broadcast_to(
                        np.asarray(self._spec.maximum, np.float32), (self._D,)).copy(), device=dev)
                want_v = self._collect and not self._compute_value_in_train
                z = v = None
                if want_v and self._value_out_ok:
                    # the actor's and the value network's bodies on the same normalised
                    # observation in ONE launch (csrc/mlp_small.hip: aa_mlp_small_forward2)
                    from agents_amd.networks import sequential
                    N0 = int(obs.shape[0])
                    abody, vbody = self._actor_network.body, self._value_network.body
                    if abody._built and vbody._built and abody._fused_small_ok():
                        z = torch.empty((N0, int(abody._f_dims[len(abody._param_layers)])),
                                        dtype=torch.float32, device=dev)
                        v = torch.empty((N0,), dtype=torch.float32, device=dev)
                        if not sequential.forward_small_pair(abody, vbody, obs, z, v):
                            z = v = None
                loc, scale, action = self._actor_network.forward_sample(
                    obs, self._seed, self._call_counter, self._arrival,
                    self._lo if self._clip else None, self._hi if self._clip else None,
                    slot="policy", z=z)
                N = loc.shape[0]
                shp = (N,) + tuple(self._spec.shape)
                action = action.view(shp)
                info = ()
                if self._collect:
                    info = {"dist_params": {"loc": loc.view(shp), "scale": scale.view(shp)}}
                    if want_v and v is not None:
                        info["value_prediction"] = v
                    elif want_v:
                        v = torch.empty((N,), dtype=torch.float32, device=dev)
                        info["value_prediction"] = self._value_network.forward(
                            obs, slot="policy", out=v) if self._value_out_ok \
                            else self._value_network.forward(obs, slot="policy").clone()
                if not batched:
                    action = action.squeeze(0)
                    info = nest_utils.map_structure(lambda t: t.squeeze(0), info)
                return policy_step.PolicyStep(action, policy_state, info)
            loc, scale = self._actor_network.forward(obs, slot="policy")
            N = loc.shape[0]
            if self._greedy:
                action = loc.clone()
            else:
                if self._call_counter is None:
                    self._call_counter = torch.zeros((1,), dtype=torch.int64, device=dev)
                action = torch.empty_like(loc)
                st = _lib.stream_ptr()
                _lib.check(lib.aa_normal_sample(loc.data_ptr(), scale.data_ptr(), loc.numel(),
                                                self._seed, self._call_counter.data_ptr(),
                                                action.data_ptr(), st), "aa_normal_sample")
                _lib.check(lib.aa_counter_add(self._call_counter.data_ptr(), 1, st),
                           "aa_counter_add")
            if self._clip:
                if self._lo is None:
                    self._lo = torch.as_tensor(np.broadcast_to(
                        np.asarray(self._spec.minimum, np.float32), (self._D,)).copy(), device=dev)
                    self._hi = torch.as_tensor(np.broadcast_to(
                        np.asarray(self._spec.maximum, np.float32), (self._D,)).copy(), device=dev)
                action = torch.maximum(torch.minimum(action, self._hi), self._lo)
            action = action.view((N,) + tuple(self._spec.shape))
            info = ()
            if self._collect:
                shp = (N,) + tuple(self._spec.shape)
                info = {"dist_params": {"loc": loc.clone().view(shp),
                                        "scale": scale.clone().view(shp)}}
                if not self._compute_value_in_train:
                    info["value_prediction"] = self._value_network.forward(
                        obs, slot="policy").clone()
        if not batched:
            action = action.squeeze(0)
            info = nest_utils.map_structure(lambda t: t.squeeze(0), info)
        return policy_step.PolicyStep(action, policy_state, info)
