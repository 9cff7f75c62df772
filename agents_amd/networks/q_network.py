"""QNetwork: encoder (conv + fc stack) followed by a linear Q-value layer.

Same constructor surface as tf_agents/networks/q_network.py:46-158 for the feed-forward case:
hidden layers use variance_scaling(scale=2, fan_in, truncated_normal)
(encoding_network.py:222-225), the Q layer U(-0.03, 0.03) with bias -0.2 (q_network.py:121-129).
Built on `sequential.Sequential`, so forward/backward run on the HIP GEMM kernels.
"""
import numpy as np

from agents_amd.networks import layers as L
from agents_amd.networks import sequential
from agents_amd.utils import nest_utils


def validate_specs(action_spec, observation_spec):
    """q_network.py:29-43: single observation, single scalar discrete action."""
    del observation_spec
    flat = nest_utils.flatten(action_spec)
    if len(flat) > 1:
        raise ValueError("Network only supports action_specs with a single action.")
    if flat[0].shape not in [(), (1,)]:
        raise ValueError("Network only supports action_specs with shape in [(), (1,)])")


class QNetwork(sequential.Sequential):
    def __init__(self, input_tensor_spec, action_spec, preprocessing_layers=None,
                 preprocessing_combiner=None, conv_layer_params=None, fc_layer_params=(75, 40),
                 dropout_layer_params=None, activation_fn="relu", kernel_initializer=None,
                 batch_squash=True, dtype=None, q_layer_activation_fn=None, name="QNetwork",
                 seed=None):
        validate_specs(action_spec, input_tensor_spec)
        # preprocessing_layers (q_network.py:70-79, encoding_network.py:117-140): for the single
        # observation tensor this class supports it is one layer, or a list of layers, applied in
        # front of the encoder -- e.g. the uint8 -> [0, 1] scaling of the Atari stack.  They are
        # this package's layer objects (networks/layers.py), prepended to the layer list, so the
        # scaling stays fused into the first convolution.  Nests of observations with a
        # preprocessing_combiner are not implemented.
        pre = []
        if preprocessing_layers is not None:
            if preprocessing_combiner is not None or isinstance(preprocessing_layers, dict) or \
                    len(nest_utils.flatten(input_tensor_spec)) != 1:
                raise NotImplementedError(
                    "preprocessing_layers: only layer(s) for a single observation tensor")
            pre = list(preprocessing_layers) if isinstance(preprocessing_layers, (list, tuple)) \
                else [preprocessing_layers]
            for l in pre:
                if not isinstance(l, L.Layer):
                    raise TypeError("preprocessing_layers must be agents_amd.networks.layers "
                                    f"objects (e.g. layers.Rescale(255.0)), got {type(l).__name__}")
        elif preprocessing_combiner is not None:
            raise NotImplementedError("preprocessing_combiner without preprocessing_layers")
        if dropout_layer_params:
            raise NotImplementedError("dropout layers are outside the hot-path scope")
        spec = nest_utils.flatten(action_spec)[0]
        num_actions = int(np.asarray(spec.maximum) - np.asarray(spec.minimum) + 1)
        init = kernel_initializer or L.VarianceScaling(2.0, "fan_in", "truncated_normal")
        layers = list(pre)
        for params in (conv_layer_params or ()):
            filters, kernel, stride = params[:3]
            layers.append(L.Conv2D(filters, kernel, stride, activation=activation_fn,
                                   kernel_initializer=init))
        if conv_layer_params:
            layers.append(L.Flatten())
        for units in (fc_layer_params or ()):
            layers.append(L.Dense(units, activation=activation_fn, kernel_initializer=init))
        layers.append(L.Dense(num_actions, activation=q_layer_activation_fn,
                              kernel_initializer=L.RandomUniform(-0.03, 0.03),
                              bias_initializer=L.Constant(-0.2)))
        super().__init__(layers, input_spec=input_tensor_spec, name=name, seed=seed)
        self._ctor = dict(input_tensor_spec=input_tensor_spec, action_spec=action_spec,
                          preprocessing_layers=preprocessing_layers,
                          conv_layer_params=conv_layer_params, fc_layer_params=fc_layer_params,
                          activation_fn=activation_fn, kernel_initializer=kernel_initializer,
                          q_layer_activation_fn=q_layer_activation_fn, name=name, seed=seed)
