"""ActorDistributionNetwork with a tanh-squashed Normal projection, as SAC uses it.

Counterparts: tf_agents/networks/actor_distribution_network.py:50-190 (MLP encoder + projection
network) and tf_agents/agents/sac/tanh_normal_projection_network.py:38-143 (one Dense emitting
2*A numbers split into means and raw standard deviations; std = std_transform(raw)).
No distribution object is materialised: `forward` returns the projection output z = [mean | raw]
and the agent's kernels (csrc/sac.hip) sample / evaluate / differentiate the squashed Normal.
"""
import numpy as np
import torch

from agents_amd import _lib
from agents_amd.networks import layers as L
from agents_amd.networks import network, normal_projection_network, sequential
from agents_amd.utils import nest_utils


def std_clip_transform(stddevs=None):
    """sac_agent.std_clip_transform marker: exp(clip(raw, -20, 2)) (sac_agent.py:48-57)."""
    return "clip_exp"


class TanhNormalProjectionNetwork:
    """Configuration of the projection: `std_transform` is "exp" (the reference default tf.exp) or
    "clip_exp" (`std_clip_transform`)."""

    def __init__(self, sample_spec=None, activation_fn=None, std_transform="exp",
                 name="TanhNormalProjectionNetwork"):
        if callable(std_transform):
            std_transform = std_transform()
        if std_transform not in ("exp", "clip_exp"):
            raise NotImplementedError("std_transform must be 'exp' or 'clip_exp'")
        if activation_fn is not None:
            raise NotImplementedError("projection activation_fn is not supported")
        self.sample_spec = sample_spec
        self.std_transform = std_transform

    @property
    def std_kind(self):
        return _lib.AA_SAC_STD_EXP if self.std_transform == "exp" else _lib.AA_SAC_STD_CLIP_EXP


def _normal_projection_net(action_spec, init_action_stddev=0.35, init_means_output_factor=0.1,
                           seed_stream_class=None, seed=None):
    """The reference's default continuous projection (actor_distribution_network.py:37-57): a
    Normal with tanh-squashed means and a state-independent softplus(bias) scale."""
    import math
    return normal_projection_network.NormalProjectionNetwork(
        action_spec, init_means_output_factor=init_means_output_factor,
        std_bias_initializer_value=math.log(math.expm1(init_action_stddev)),
        mean_transform=normal_projection_network.tanh_squash_to_spec, state_dependent_std=False,
        scale_distribution=False, seed=seed)


def _activation_name(fn):
    if fn is None or isinstance(fn, str):
        return fn
    name = getattr(fn, "__name__", None)
    if name in ("relu", "tanh"):
        return name
    raise NotImplementedError(f"activation {fn!r}: pass 'relu' or 'tanh'")


def _resolve_projection(proj, spec, seed):
    if proj is _normal_projection_net:
        return proj(spec, seed=seed)
    if isinstance(proj, (TanhNormalProjectionNetwork,
                         normal_projection_network.NormalProjectionNetwork)):
        return proj
    if callable(proj):
        return proj(spec)
    return proj


class ActorDistributionNetwork(network.Network):
    """`continuous_projection_net` decides what is built (actor_distribution_network.py:50-190):
    the default Normal projection yields the PPO actor (`ppo_actor_network.TanhNormalActorNet`:
    MLP body + fused tanh-squash / softplus head, the network `PPOAgent` trains); a
    `TanhNormalProjectionNetwork` yields the SAC actor implemented by this class."""

    def __new__(cls, input_tensor_spec=None, output_tensor_spec=None, preprocessing_layers=None,
                preprocessing_combiner=None, conv_layer_params=None, fc_layer_params=(200, 100),
                dropout_layer_params=None, activation_fn="relu", kernel_initializer=None,
                seed_stream_class=None, seed=None, batch_squash=True, dtype=torch.float32,
                discrete_projection_net=None, continuous_projection_net=_normal_projection_net,
                name="ActorDistributionNetwork"):
        flat = nest_utils.flatten(output_tensor_spec)
        proj = _resolve_projection(continuous_projection_net, flat[0], seed) \
            if len(flat) == 1 else None
        if isinstance(proj, normal_projection_network.NormalProjectionNetwork):
            import math

            from agents_amd.agents.ppo import ppo_actor_network as pan
            if preprocessing_layers or preprocessing_combiner or conv_layer_params or \
                    dropout_layer_params:
                raise NotImplementedError("only fc_layer_params encoders are implemented")
            if not proj.squash_means:
                raise NotImplementedError("mean_transform=None is not implemented")
            spec, D = pan._flat_action_spec(output_tensor_spec)
            ki = kernel_initializer or L.GlorotUniform()
            act = _activation_name(activation_fn)
            layers = [L.Dense(int(n), act, kernel_initializer=ki)
                      for n in (fc_layer_params or ())]
            layers.append(L.Dense(D, None, kernel_initializer=L.VarianceScaling(
                proj.init_means_output_factor)))
            body = sequential.Sequential(layers, seed=seed, name="ActorDistributionBody")
            b = proj.std_bias_initializer_value
            init_std = math.log1p(math.exp(b))          # softplus(bias)
            return pan.TanhNormalActorNet(body, output_tensor_spec, init_std,
                                          input_spec=input_tensor_spec, name=name)
        return super().__new__(cls)

    def __init__(self, input_tensor_spec, output_tensor_spec, preprocessing_layers=None,
                 preprocessing_combiner=None, conv_layer_params=None, fc_layer_params=(200, 100),
                 dropout_layer_params=None, activation_fn="relu", kernel_initializer=None,
                 seed_stream_class=None, seed=None, batch_squash=True, dtype=torch.float32,
                 discrete_projection_net=None,
                 continuous_projection_net=_normal_projection_net,
                 name="ActorDistributionNetwork"):
        super().__init__(input_tensor_spec=input_tensor_spec, state_spec=(), name=name)
        activation_fn = _activation_name(activation_fn)
        if preprocessing_layers or preprocessing_combiner or conv_layer_params or \
                dropout_layer_params:
            raise NotImplementedError("only fc_layer_params encoders are implemented")
        flat = nest_utils.flatten(output_tensor_spec)
        if len(flat) != 1 or flat[0].dtype != torch.float32:
            raise NotImplementedError("a single continuous (float32) action spec is supported")
        self._action_spec = flat[0]
        self._A = int(np.prod(flat[0].shape)) or 1
        proj = _resolve_projection(continuous_projection_net, flat[0], seed)
        if not isinstance(proj, TanhNormalProjectionNetwork):
            raise NotImplementedError("continuous_projection_net must build a "
                                      "TanhNormalProjectionNetwork")
        self._projection = proj
        ki = kernel_initializer or L.GlorotUniform()
        layers = [L.Dense(int(n), activation_fn, kernel_initializer=ki)
                  for n in (fc_layer_params or ())]
        layers.append(L.Dense(2 * self._A, None, kernel_initializer=L.GlorotUniform()))
        self._body = sequential.Sequential(layers, input_spec=input_tensor_spec, seed=seed)

    @property
    def body(self):
        return self._body

    @property
    def projection(self):
        return self._projection

    @property
    def action_dims(self):
        return self._A

    def create_variables(self, input_tensor_spec=None, device=None, **kwargs):
        self._body.create_variables(input_tensor_spec, device=device)
        self._built = True
        return {"loc": (self._A,), "scale_diag": (self._A,)}

    @property
    def flat_params(self):
        return self._body.flat_params

    @property
    def flat_grads(self):
        return self._body.flat_grads

    @property
    def variables(self):
        return self._body.variables

    def set_weights(self, arrays):
        self._body.set_weights(arrays)

    def get_weights(self):
        return self._body.get_weights()

    def forward(self, observation, slot=0, need_grad=False):
        """z[B, 2A] = [mean | raw_std] (buffer owned by the network)."""
        B = observation.shape[0]
        return self._body.forward(observation.reshape((B,) + tuple(
            self._body._input_tensor_spec.shape)), slot=slot, need_grad=need_grad)

    def forward_sample_ok(self, observation):
        """The body runs as ONE wide-MLP launch at this batch size, which can then also draw the
        tanh-squashed sample of its own head output (`forward_sample`)."""
        B = int(observation.shape[0])
        return observation.dtype == torch.float32 and observation.is_cuda and \
            not self._body._fused_small_ok() and self._body.wide_ok(B)

    def forward_sample(self, observation, tail, slot=0, need_grad=False):
        """`forward` + the actor head's sample in one launch (csrc/mlp_wide.hip:
        aa_mlp_wide_forward_sample); `tail`: a filled `_lib.SacSampleTail` with net = 0."""
        from agents_amd.networks import sequential
        B = observation.shape[0]
        x = observation.reshape((B,) + tuple(self._body._input_tensor_spec.shape))
        return sequential.forward_wide([self._body], [x], slot=slot, need_grad=need_grad,
                                       sample_tail=tail)[0]

    def forward_sample2(self, obs_a, tail_a, slot_a, need_grad_a, obs_b, tail_b, slot_b,
                        need_grad_b):
        """Two `forward_sample`s of THIS network (same weights, two inputs, two slots, two draws)
        in one launch: tail_a.net = 0, tail_b.net = 1 (aa_mlp_wide_forward_sample2)."""
        from agents_amd.networks import sequential
        shp = tuple(self._body._input_tensor_spec.shape)
        xa = obs_a.reshape((obs_a.shape[0],) + shp)
        xb = obs_b.reshape((obs_b.shape[0],) + shp)
        return sequential.forward_wide([self._body, self._body], [xa, xb],
                                       slots=[slot_a, slot_b],
                                       need_grads=[need_grad_a, need_grad_b],
                                       sample_tail=(tail_a, tail_b))

    def backward(self, dz, slot=0, side_stream=None):
        self._body.backward(dz, slot=slot, side_stream=side_stream)

    def call(self, inputs, step_type=None, network_state=(), training=False, **kwargs):
        return self.forward(inputs, slot="call").clone(), network_state
