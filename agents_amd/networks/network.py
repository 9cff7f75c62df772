"""Network base class: the protocol TF-Agents agents and policies expect from a network.

Counterpart of tf_agents/networks/network.py:752 (subset): `__call__(inputs, step_type=None,
network_state=(), training=False) -> (output, next_state)`, `create_variables`, `variables`,
`trainable_weights`, `copy`, `state_spec`, `input_tensor_spec`, `losses`.
"""


class Network:
    def __init__(self, input_tensor_spec=None, state_spec=(), name=None):
        self._input_tensor_spec = input_tensor_spec
        self._state_spec = state_spec
        self._name = name or type(self).__name__
        self._built = False

    @property
    def name(self):
        return self._name

    @property
    def input_tensor_spec(self):
        return self._input_tensor_spec

    @property
    def state_spec(self):
        return self._state_spec

    @property
    def built(self):
        return self._built

    def create_variables(self, input_tensor_spec=None, **kwargs):
        raise NotImplementedError

    @property
    def variables(self):
        raise NotImplementedError

    @property
    def trainable_weights(self):
        return self.variables

    @property
    def trainable_variables(self):
        return self.trainable_weights

    @property
    def non_trainable_weights(self):
        return []

    @property
    def losses(self):
        return []

    def copy(self, **kwargs):
        raise NotImplementedError

    def get_initial_state(self, batch_size=None):
        return ()

    def call(self, inputs, step_type=None, network_state=(), training=False):
        raise NotImplementedError

    def __call__(self, inputs, step_type=None, network_state=(), training=False, **kwargs):
        return self.call(inputs, step_type=step_type, network_state=network_state,
                         training=training, **kwargs)
