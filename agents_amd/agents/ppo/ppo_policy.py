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

    def _ensure_draw_state(self, dev):
        if self._call_counter is None:
            self._call_counter = torch.zeros((1,), dtype=torch.int64, device=dev)
        if self._arrival is None:
            self._arrival = torch.zeros((1,), dtype=torch.int64, device=dev)
        if self._clip and self._lo is None:
            self._lo = torch.as_tensor(np.broadcast_to(
                np.asarray(self._spec.minimum, np.float32), (self._D,)).copy(), device=dev)
            self._hi = torch.as_tensor(np.broadcast_to(
                np.asarray(self._spec.maximum, np.float32), (self._D,)).copy(), device=dev)

    def _one_launch_step(self, obs, dev):
        """The whole collect step -- observation normalisation, actor body, value body, actor
        head, Normal draw, clip, Philox counter -- as ONE launch (csrc/ppo.hip: aa_ppo_policy_step;
        every piece with the arithmetic of the launch it replaces).  (action, info) or None when
        the configuration does not qualify: PPOActorNetwork-style head on a <= 64-wide body, a
        <= 64-wide value body, float32 rank-1 observations, collect mode with value predictions,
        no normaliser or one with a single float32 leaf."""
        from agents_amd.networks import sequential
        from agents_amd.utils import tensor_normalizer as tn
        act, val = self._actor_network, self._value_network
        if not (self._collect and not self._compute_value_in_train and self._value_out_ok and
                hasattr(act, "forward_sample") and hasattr(act, "_head_params")):
            return None
        if obs.dim() != 2 or obs.dtype != torch.float32 or not obs.is_cuda or obs.stride(1) != 1:
            return None
        la, lv = sequential.small_mlp_layout(act.body), sequential.small_mlp_layout(val.body)
        if la is None or lv is None:
            return None
        nrm = self._observation_normalizer
        mean = num = den = None
        if nrm is not None:
            if not isinstance(nrm, tn.TensorNormalizer) or nrm._state is None or \
                    len(nrm._state) != 1 or \
                    nrm._flat_specs[0].dtype != torch.float32 or \
                    len(nrm._flat_specs[0].shape) != 1:
                return None
            mean, num, den = nrm._mean_var_ptrs(nrm._state[0])
        self._ensure_draw_state(dev)
        N = int(obs.shape[0])
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        loc, scale, action, v = f(N, self._D), f(N, self._D), f(N, self._D), f(N)
        d = _lib.PpoPolicyStepDesc()
        d.x, d.ldx, d.B = obs.data_ptr(), obs.stride(0), N
        d.nrm_mean, d.nrm_var_num, d.nrm_var_den = mean, num, den
        d.nrm_eps, d.nrm_clip = 1e-3, 5.0          # TensorNormalizer.normalize's defaults
        (d.params_a, d.n_layers_a, d.dims_a, d.acts_a, d.k_off_a, d.b_off_a) = la
        (d.params_b, d.n_layers_b, d.dims_b, d.acts_b, d.k_off_b, d.b_off_b) = lv
        d.value_out = v.data_ptr()
        d.std_bias = act._head_params.data_ptr()
        d.act_mean, d.act_mag = _lib.ptr(act._mean), _lib.ptr(act._mag)
        d.D = self._D
        d.loc, d.scale = loc.data_ptr(), scale.data_ptr()
        d.seed = self._seed
        d.call_counter_dev = self._call_counter.data_ptr()
        d.arrival_dev = self._arrival.data_ptr()
        d.clip_lo = self._lo.data_ptr() if self._clip else None
        d.clip_hi = self._hi.data_ptr() if self._clip else None
        d.action = action.data_ptr()
        import ctypes
        rc = _lib.load().aa_ppo_policy_step(ctypes.byref(d), _lib.stream_ptr())
        if rc in (-22, -34):       # AA_ERR_INVALID / AA_ERR_RANGE: not this kernel's shapes
            return None
        _lib.check(rc, "aa_ppo_policy_step")
        shp = (N,) + tuple(self._spec.shape)
        return action.view(shp), {"dist_params": {"loc": loc.view(shp), "scale": scale.view(shp)},
                                  "value_prediction": v}

    def _action(self, time_step, policy_state, seed):
        lib = _lib.load()
        obs = time_step.observation
        batched = time_step.step_type.dim() > 0
        if not batched:
            obs = obs.unsqueeze(0)
        dev = obs.device
        with torch.cuda.device(dev):
            if _FUSE_SAMPLE and not self._greedy:
                step = self._one_launch_step(obs, dev)
                if step is not None:
                    action, info = step
                    if not batched:
                        action = action.squeeze(0)
                        info = nest_utils.map_structure(lambda t: t.squeeze(0), info)
                    return policy_step.PolicyStep(action, policy_state, info)
            obs = self._normalized(obs)
            fused = _FUSE_SAMPLE and not self._greedy and \
                hasattr(self._actor_network, "forward_sample")
            if fused:
                # head + draw + clip + counter in one launch, loc / scale written where the policy
                # info wants them (csrc/ppo.hip: aa_ppo_head_forward_sample; bit-identical to the
                # launches below)
                self._ensure_draw_state(dev)
                loc, scale, action = self._actor_network.forward_sample(
                    obs, self._seed, self._call_counter, self._arrival,
                    self._lo if self._clip else None, self._hi if self._clip else None,
                    slot="policy")
                N = loc.shape[0]
                shp = (N,) + tuple(self._spec.shape)
                action = action.view(shp)
                info = ()
                if self._collect:
                    info = {"dist_params": {"loc": loc.view(shp), "scale": scale.view(shp)}}
                    if not self._compute_value_in_train:
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
