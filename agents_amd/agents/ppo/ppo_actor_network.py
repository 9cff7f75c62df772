"""Actor / value networks for PPO on HIP kernels.

  PPOActorNetwork.create_sequential_actor_net   tf_agents/agents/ppo/ppo_actor_network.py:42-113
      Dense(tanh) x n -> Dense(D) means -> tanh_and_scale_to_spec -> state-independent
      std = softplus(bias)  -> MultivariateNormalDiag(loc, scale)
  ValueNetwork                                   tf_agents/networks/value_network.py
  (the reference tests' DummyActorNet: one Dense whose output is split into (loc, scale),
   agents/ppo/ppo_agent_test.py:45-125 -- `SplitNormalActorNet` below)

No distribution objects are materialised: an actor network here maps observations to the two
parameter tensors (loc[N,D], scale[N,D]) of a diagonal Normal, and takes the gradients wrt both
back.  The MLP bodies are `networks.sequential.Sequential` (fp32 MFMA GEMMs); the heads are the
small kernels of csrc/ppo.hip.
"""
import math

import numpy as np
import torch

from agents_amd import _lib, ops
from agents_amd.networks import layers as L
from agents_amd.networks import network, sequential
from agents_amd.utils import nest_utils


def _flat_action_spec(action_spec):
    flat = nest_utils.flatten(action_spec)
    if len(flat) != 1:
        raise ValueError("PPO networks here support a single (possibly vector) continuous action")
    spec = flat[0]
    if spec.dtype != torch.float32:
        raise ValueError("PPO networks here support float32 (continuous) actions")
    if len(spec.shape) > 1:
        raise ValueError("action spec must be a scalar or a vector")
    return spec, int(np.prod(spec.shape)) if len(spec.shape) else 1


class NormalActorNet(network.Network):
    """Base: an MLP body plus a head that turns its output into (loc, scale)."""

    def __init__(self, body, action_spec, input_spec=None, name=None):
        super().__init__(input_tensor_spec=input_spec, state_spec=(), name=name)
        self._body = body
        self._action_spec = action_spec
        self._spec, self._D = _flat_action_spec(action_spec)
        self._bufs = {}

    # ---- parameters -------------------------------------------------------------------------
    @property
    def body(self):
        return self._body

    @property
    def action_dims(self):
        return self._D

    def create_variables(self, input_tensor_spec=None, device=None, **kwargs):
        if input_tensor_spec is not None:
            self._input_tensor_spec = input_tensor_spec
        out = self._body.create_variables(self._input_tensor_spec, device=device)
        self._check_body_output(out)
        if not self._built:
            self._create_head_variables(self._body.flat_params.device)
        self._built = True
        return {"loc": (self._D,), "scale": (self._D,)}

    def _check_body_output(self, out_shape):
        raise NotImplementedError

    def _create_head_variables(self, device):
        self._head_params = torch.zeros((0,), dtype=torch.float32, device=device)
        self._head_grads = torch.zeros((0,), dtype=torch.float32, device=device)

    @property
    def flat_size(self):
        return self._body.flat_size + self._head_size()

    def _head_size(self):
        return 0

    def rebind(self, flat_params, flat_grads):
        nb = self._body.flat_size
        self._body.rebind(flat_params[:nb], flat_grads[:nb])
        nh = self._head_size()
        if nh:
            flat_params[nb:nb + nh].copy_(self._head_params)
            flat_grads[nb:nb + nh].zero_()
            self._head_params = flat_params[nb:nb + nh]
            self._head_grads = flat_grads[nb:nb + nh]

    @property
    def variables(self):
        return self._body.variables

    @property
    def kernels(self):
        return self._body.kernels

    @property
    def kernel_grads(self):
        return self._body.kernel_grads

    def _buf(self, slot, N, names):
        key = (slot, N)
        b = self._bufs.get(key)
        if b is None:
            dev = self._body.flat_params.device
            b = {n: torch.empty((N, self._D), dtype=torch.float32, device=dev) for n in names}
            self._bufs[key] = b
        return b

    # ---- execution (subclasses) ---------------------------------------------------------------
    def forward(self, obs, slot=0, need_grad=False):
        """obs [N, *obs_shape] -> (loc[N,D], scale[N,D]); buffers owned by the network."""
        raise NotImplementedError

    def backward(self, dloc, dscale, slot=0):
        raise NotImplementedError

    def call(self, inputs, step_type=None, network_state=(), training=False, **kwargs):
        loc, scale = self.forward(inputs, slot="call")
        return {"loc": loc.clone(), "scale": scale.clone()}, network_state


class SplitNormalActorNet(NormalActorNet):
    """Body emits 2*D numbers per sample: the first D are loc, the last D are scale
    (DummyActorNet of the reference's tests; also a state-dependent-std projection)."""

    def _check_body_output(self, out_shape):
        if tuple(out_shape) != (2 * self._D,):
            raise ValueError(f"body must emit {2 * self._D} values, emits {tuple(out_shape)}")

    def forward(self, obs, slot=0, need_grad=False):
        out = self._body.forward(obs, slot=slot, need_grad=need_grad)
        N = out.shape[0]
        b = self._buf(slot, N, ("loc", "scale", "dout"))
        b["loc"].copy_(out[:, :self._D])
        b["scale"].copy_(out[:, self._D:])
        return b["loc"], b["scale"]

    def backward(self, dloc, dscale, slot=0, side_stream=None):
        N = dloc.shape[0]
        dout = torch.empty((N, 2 * self._D), dtype=torch.float32, device=dloc.device)
        dout[:, :self._D].copy_(dloc)
        dout[:, self._D:].copy_(dscale)
        self._body.backward(dout, slot=slot, side_stream=side_stream)


class TanhNormalActorNet(NormalActorNet):
    """PPOActorNetwork head: loc = mean + magnitude * tanh(z) for a bounded spec (z itself when
    unbounded), scale = softplus(std_bias) with one trainable bias per action dimension,
    initialised so that softplus(bias) = init_action_stddev (ppo_actor_network.py:30-113)."""

    def __init__(self, body, action_spec, init_action_stddev=0.35, input_spec=None, name=None):
        super().__init__(body, action_spec, input_spec, name or "PPOActorNetwork")
        self._init_std = float(init_action_stddev)
        lo = np.broadcast_to(np.asarray(self._spec.minimum, np.float32), (self._D,))
        hi = np.broadcast_to(np.asarray(self._spec.maximum, np.float32), (self._D,))
        self._bounded = bool(np.all(np.isfinite(lo)) and np.all(np.isfinite(hi)))
        self._mean_h = ((hi + lo) / 2.0).astype(np.float32)
        self._mag_h = ((hi - lo) / 2.0).astype(np.float32)
        self._mean = self._mag = None

    def _check_body_output(self, out_shape):
        if tuple(out_shape) != (self._D,):
            raise ValueError(f"body must emit {self._D} means, emits {tuple(out_shape)}")

    def _head_size(self):
        return (self._D + 3) // 4 * 4

    def _create_head_variables(self, device):
        n = self._head_size()
        # softplus(b) = s  <=>  b = log(exp(s) - 1)
        b0 = math.log(math.expm1(self._init_std))
        host = np.zeros((n,), np.float32)
        host[:self._D] = b0
        self._head_params = torch.from_numpy(host).to(device)
        self._head_grads = torch.zeros_like(self._head_params)
        if self._bounded:
            self._mean = torch.from_numpy(self._mean_h.copy()).to(device)
            self._mag = torch.from_numpy(self._mag_h.copy()).to(device)

    @property
    def std_bias(self):
        return self._head_params[:self._D]

    @property
    def std_bias_grad(self):
        return self._head_grads[:self._D]

    @property
    def variables(self):
        return self._body.variables + [self.std_bias]

    def forward(self, obs, slot=0, need_grad=False):
        lib = _lib.load()
        z = self._body.forward(obs, slot=slot, need_grad=need_grad)
        N = z.shape[0]
        b = self._buf(slot, N, ("loc", "scale", "dz", "dbias"))
        b["z"] = z
        _lib.check(lib.aa_ppo_head_forward(
            z.data_ptr(), self._head_params.data_ptr(), _lib.ptr(self._mean), _lib.ptr(self._mag),
            N, self._D, b["loc"].data_ptr(), b["scale"].data_ptr(), _lib.stream_ptr()),
            "aa_ppo_head_forward")
        return b["loc"], b["scale"]

    def forward_sample(self, obs, seed, call_counter, arrival, clip_lo, clip_hi, slot=0, z=None):
        """(loc, scale, action) as FRESH tensors of the caller's: `forward` + one draw of
        Normal(loc, scale) (+ the clip to [clip_lo, clip_hi], tensors [D] or None) in the head's
        launch (aa_ppo_head_forward_sample), which also advances `call_counter` -- the collect
        policy's step without the sample / counter launches and without copying loc and scale
        out of the network's buffers.  `z`: the body's output when the caller has already
        computed it (PPOPolicy runs the actor's and the value network's bodies in one launch)."""
        lib = _lib.load()
        z = self._body.forward(obs, slot=slot, need_grad=False) if z is None else z
        N = z.shape[0]
        f = lambda: torch.empty((N, self._D), dtype=torch.float32, device=z.device)
        loc, scale, action = f(), f(), f()
        _lib.check(lib.aa_ppo_head_forward_sample(
            z.data_ptr(), self._head_params.data_ptr(), _lib.ptr(self._mean), _lib.ptr(self._mag),
            N, self._D, loc.data_ptr(), scale.data_ptr(), seed, call_counter.data_ptr(),
            arrival.data_ptr(), _lib.ptr(clip_lo), _lib.ptr(clip_hi), action.data_ptr(),
            _lib.stream_ptr()), "aa_ppo_head_forward_sample")
        return loc, scale, action

    def backward(self, dloc, dscale, slot=0, side_stream=None):
        lib = _lib.load()
        N = dloc.shape[0]
        b = self._bufs[(slot, N)]
        _lib.check(lib.aa_ppo_head_backward(
            b["z"].data_ptr(), self._head_params.data_ptr(), _lib.ptr(self._mag),
            dloc.data_ptr(), dscale.data_ptr(), N, self._D, b["dz"].data_ptr(),
            b["dbias"].data_ptr(), _lib.stream_ptr()), "aa_ppo_head_backward")
        ops.colsum(b["dbias"], self.std_bias_grad)
        self._body.backward(b["dz"], slot=slot, side_stream=side_stream)


class PPOActorNetwork:
    """Factory with the reference's name and method (ppo_actor_network.py:30-113)."""

    def __init__(self, seed_stream_class=None):
        self.seed_stream_class = seed_stream_class

    def create_sequential_actor_net(self, fc_layer_units, action_tensor_spec, seed=None,
                                    init_action_stddev=0.35):
        spec, D = _flat_action_spec(action_tensor_spec)
        layers = [L.Dense(u, "tanh", kernel_initializer=L.Orthogonal(seed=seed))
                  for u in fc_layer_units]
        layers.append(L.Dense(D, None, kernel_initializer=L.VarianceScaling(0.1)))
        body = sequential.Sequential(layers, seed=seed, name="PPOActorBody")
        return TanhNormalActorNet(body, action_tensor_spec, init_action_stddev)


class ValueNet(network.Network):
    """MLP -> one value per sample (tf_agents/networks/value_network.py: fc layers + Dense(1),
    output squeezed)."""

    def __init__(self, body, input_spec=None, name=None):
        super().__init__(input_tensor_spec=input_spec, state_spec=(), name=name or "ValueNetwork")
        self._body = body

    @property
    def body(self):
        return self._body

    def create_variables(self, input_tensor_spec=None, device=None, **kwargs):
        if input_tensor_spec is not None:
            self._input_tensor_spec = input_tensor_spec
        out = self._body.create_variables(self._input_tensor_spec, device=device)
        if tuple(out) != (1,):
            raise ValueError(f"value network body must emit one value, emits {tuple(out)}")
        self._built = True
        return ()

    @property
    def flat_size(self):
        return self._body.flat_size

    def rebind(self, flat_params, flat_grads):
        self._body.rebind(flat_params, flat_grads)

    @property
    def variables(self):
        return self._body.variables

    @property
    def kernels(self):
        return self._body.kernels

    @property
    def kernel_grads(self):
        return self._body.kernel_grads

    def forward(self, obs, slot=0, need_grad=False, out=None):
        """`out` (a contiguous float32 [B] or [B, 1] tensor of the caller's, inference only): the
        value head writes it instead of the network's own buffer."""
        if out is not None and not need_grad:
            return self._body.forward(obs, slot=slot, need_grad=False, out=out.view(-1, 1)).view(-1)
        return self._body.forward(obs, slot=slot, need_grad=need_grad).view(-1)

    def backward(self, dv, slot=0, side_stream=None):
        self._body.backward(dv.view(-1, 1), slot=slot, side_stream=side_stream)

    def call(self, inputs, step_type=None, network_state=(), training=False, **kwargs):
        return self.forward(inputs, slot="call").clone(), network_state


def value_network(fc_layer_params=(75, 40), activation="relu", seed=None):
    """ValueNetwork(observation_spec, fc_layer_params=...) equivalent."""
    layers = [L.Dense(u, activation) for u in (fc_layer_params or ())]
    layers.append(L.Dense(1, None, kernel_initializer=L.RandomUniform(-0.03, 0.03)))
    return ValueNet(sequential.Sequential(layers, seed=seed, name="ValueBody"))
