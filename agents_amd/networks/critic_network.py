"""CriticNetwork: Q(observation, action) -> scalar, for SAC / DDPG-style agents.

Counterpart of tf_agents/agents/ddpg/critic_network.py:30-190 (the class the SAC example
instantiates, examples/sac/haarnoja18/sac_train_eval.py:182-190): observation and action are
concatenated and passed through `joint_fc_layer_params` Dense(relu) layers and a final Dense(1).
The optional per-input towers `observation_fc_layer_params` / `action_fc_layer_params`
(critic_network.py:126-145: Dense(activation_fn) stacks applied to each input before the
concatenation, :163-185) are `Sequential`s of their own in front of the joint body; the
convolutional observation encoder and the dropout lists are not implemented -- the SAC
configuration leaves all of them at None, and only the tower-less layout takes the twin-critic
fast path (`pair_ok`).

The body is a `networks.sequential.Sequential` on the concatenated input (fp32 MFMA GEMMs, the
small-N kernels for the 1-unit head).  `backward` can return d Q / d action: SAC's actor loss
differentiates through the critics.
"""
import numpy as np
import torch

from agents_amd import ops

from agents_amd.networks import layers as L
from agents_amd.networks import network, sequential
from agents_amd.specs import tensor_spec
from agents_amd.utils import nest_utils


class CriticNetwork(network.Network):
    def __init__(self, input_tensor_spec, observation_conv_layer_params=None,
                 observation_fc_layer_params=None, observation_dropout_layer_params=None,
                 action_fc_layer_params=None, action_dropout_layer_params=None,
                 joint_fc_layer_params=None, joint_dropout_layer_params=None,
                 activation_fn="relu", output_activation_fn=None, kernel_initializer=None,
                 last_kernel_initializer=None, last_layer=None, name="CriticNetwork", seed=None):
        super().__init__(input_tensor_spec=input_tensor_spec, state_spec=(), name=name)
        if any(p for p in (observation_conv_layer_params, observation_dropout_layer_params,
                           action_dropout_layer_params, joint_dropout_layer_params)):
            raise NotImplementedError("convolutional observation encoders and dropout layers "
                                      "are not implemented")
        if last_layer is not None:
            raise NotImplementedError("last_layer is not supported")
        obs_spec, act_spec = input_tensor_spec
        if len(nest_utils.flatten(act_spec)) > 1:
            raise ValueError("Only a single action is supported by this network")
        if len(nest_utils.flatten(obs_spec)) > 1:
            raise ValueError("Only a single observation is supported by this network.")
        self._obs_dim = int(np.prod(nest_utils.flatten(obs_spec)[0].shape))
        self._act_dim = int(np.prod(nest_utils.flatten(act_spec)[0].shape)) or 1
        self._ctor = dict(input_tensor_spec=input_tensor_spec,
                          observation_fc_layer_params=observation_fc_layer_params,
                          action_fc_layer_params=action_fc_layer_params,
                          joint_fc_layer_params=joint_fc_layer_params,
                          activation_fn=activation_fn, output_activation_fn=output_activation_fn,
                          kernel_initializer=kernel_initializer,
                          last_kernel_initializer=last_kernel_initializer, name=name)
        # reference defaults: VarianceScaling(1/3, fan_in, uniform); last layer U(-0.003, 0.003)
        ki = kernel_initializer or L.VarianceScaling(1.0 / 3.0, "fan_in", "uniform")
        lki = last_kernel_initializer or L.RandomUniform(-0.003, 0.003)
        self._seed = seed

        def tower(params, in_dim, k):
            # utils.mlp_layers(None, fc_layer_params, None, activation_fn, kernel_initializer)
            if not params:
                return None, in_dim
            net = sequential.Sequential(
                [L.Dense(int(n), activation_fn, kernel_initializer=ki) for n in params],
                input_spec=tensor_spec.TensorSpec((in_dim,), torch.float32),
                seed=None if seed is None else seed + 7919 * k)
            return net, int(params[-1])

        self._obs_tower, self._fo = tower(observation_fc_layer_params, self._obs_dim, 1)
        self._act_tower, self._fa = tower(action_fc_layer_params, self._act_dim, 2)
        layers = [L.Dense(int(n), activation_fn, kernel_initializer=ki)
                  for n in (joint_fc_layer_params or ())]
        layers.append(L.Dense(1, output_activation_fn, kernel_initializer=lki))
        self._body = sequential.Sequential(
            layers, input_spec=tensor_spec.TensorSpec((self._fo + self._fa,), torch.float32),
            seed=seed)
        self._flat = None        # (params, grads) of all parts when there are towers
        self._inputs = {}

    # ---- parameters -------------------------------------------------------------------------
    @property
    def body(self):
        return self._body

    @property
    def bodies(self):
        """The network's `Sequential`s in the order of its flat parameter buffer (= the order the
        reference creates the layers in: observation tower, action tower, joint stack)."""
        return [n for n in (self._obs_tower, self._act_tower, self._body) if n is not None]

    @property
    def has_towers(self):
        return self._obs_tower is not None or self._act_tower is not None

    def create_variables(self, input_tensor_spec=None, device=None, **kwargs):
        for n in self.bodies:
            n.create_variables(device=device)
        if self.has_towers and self._flat is None:
            total = sum(n.flat_size for n in self.bodies)
            dev = self._body.flat_params.device
            self._bind(torch.empty((total,), dtype=torch.float32, device=dev),
                       torch.zeros((total,), dtype=torch.float32, device=dev))
        self._built = True
        return ()

    def _bind(self, flat_params, flat_grads):
        off = 0
        for n in self.bodies:
            k = n.flat_size
            n.rebind(flat_params[off:off + k], flat_grads[off:off + k])
            off += k
        self._flat = (flat_params, flat_grads)

    @property
    def flat_params(self):
        return self._flat[0] if self.has_towers else self._body.flat_params

    @property
    def flat_grads(self):
        return self._flat[1] if self.has_towers else self._body.flat_grads

    @property
    def flat_size(self):
        return sum(n.flat_size for n in self.bodies)

    def rebind(self, flat_params, flat_grads):
        if not self.has_towers:
            self._body.rebind(flat_params, flat_grads)
            return
        if flat_params.numel() != self.flat_size or flat_grads.numel() != self.flat_size:
            raise ValueError(f"rebind needs views of {self.flat_size} elements")
        self._bind(flat_params, flat_grads)

    @property
    def variables(self):
        return [v for n in self.bodies for v in n.variables]

    @property
    def has_regularization(self):
        return any(n.has_regularization for n in self.bodies)

    def copy(self, **kwargs):
        args = dict(self._ctor)
        args.update(kwargs)
        seed = args.pop("seed", None if self._seed is None else self._seed + 1)
        return CriticNetwork(seed=seed, **args)

    def set_weights(self, arrays):
        arrays = list(arrays)
        for n in self.bodies:
            k = len(n.variables)
            n.set_weights(arrays[:k])
            arrays = arrays[k:]
        if arrays:
            raise ValueError("set_weights: more arrays than variables")

    def get_weights(self):
        return [w for n in self.bodies for w in n.get_weights()]

    # ---- execution ----------------------------------------------------------------------------
    def _input(self, slot, B, dev):
        key = (slot, B)
        buf = self._inputs.get(key)
        if buf is None:
            f = lambda n: torch.empty((B, n), dtype=torch.float32, device=dev)
            buf = {"x": f(self._fo + self._fa), "dx": f(self._fo + self._fa)}
            if self._obs_tower is not None:
                buf["o_in"], buf["do"] = f(self._obs_dim), f(self._fo)
            if self._act_tower is not None:
                buf["a_in"], buf["da_feat"], buf["da"] = f(self._act_dim), f(self._fa), \
                    f(self._act_dim)
            self._inputs[key] = buf
        return buf

    def _forward_towers(self, o2, a2, buf, slot, need_grad):
        """critic_network.py:163-185: each input through its Dense stack, then the concatenation."""
        def through(net, src, stage):
            if net is None:
                return src
            if src.dtype != torch.float32 or not src.is_contiguous():
                stage.copy_(src)          # tf.cast(..., tf.float32) / a strided slice
                src = stage
            return net.forward(src, slot=slot, need_grad=need_grad)
        of = through(self._obs_tower, o2, buf.get("o_in"))
        af = through(self._act_tower, a2, buf.get("a_in"))
        if of.dtype == torch.float32 and af.dtype == torch.float32 and \
                of.stride(-1) == 1 and af.stride(-1) == 1:
            ops.copy_segments([(of, buf["x"][:, :self._fo]), (af, buf["x"][:, self._fo:])])
        else:
            buf["x"][:, :self._fo].copy_(of)
            buf["x"][:, self._fo:].copy_(af)
        return self._body.forward(buf["x"], slot=slot, need_grad=need_grad)

    def forward(self, observation, action, slot=0, need_grad=False, x_cat=None):
        """q[B] for observation [B, *obs] and action [B, *act] (buffer owned by the network).
        x_cat: the caller's own [B, obs + act] concatenation of the two (SacAgent builds one per
        phase for both twin critics in a single launch): used as is, must stay untouched until the
        backward pass of this slot."""
        B = observation.shape[0]
        buf = self._input(slot, B, observation.device)
        if self.has_towers:      # (x_cat is the RAW [obs | act] row: not this layout's joint input)
            return self._forward_towers(observation.reshape(B, -1), action.reshape(B, -1), buf,
                                        slot, need_grad).view(B)
        if x_cat is not None:
            if tuple(x_cat.shape) != (B, self._obs_dim + self._act_dim) or \
                    not x_cat.is_contiguous() or x_cat.dtype != torch.float32:
                raise ValueError("x_cat must be a contiguous float32 [B, obs + act] tensor")
            return self._body.forward(x_cat, slot=slot, need_grad=need_grad).view(B)
        o2, a2 = observation.reshape(B, -1), action.reshape(B, -1)
        if o2.dtype == torch.float32 and a2.dtype == torch.float32 and \
                o2.stride(-1) == 1 and a2.stride(-1) == 1:
            ops.copy_segments([(o2, buf["x"][:, :self._obs_dim]),
                               (a2, buf["x"][:, self._obs_dim:])])
        else:
            # tf.cast(..., tf.float32) of the reference (critic_network.py:150-170): float64 /
            # integer observation or action specs
            buf["x"][:, :self._obs_dim].copy_(o2)
            buf["x"][:, self._obs_dim:].copy_(a2)
        return self._body.forward(buf["x"], slot=slot, need_grad=need_grad).view(B)

    def backward(self, dq, slot=0, param_grads=True, want_action_grad=False, side_stream=None):
        """Backpropagates d loss / d q [B]; returns d loss / d action [B, act] if asked."""
        B = dq.shape[0]
        buf = self._inputs[(slot, B)]
        if self.has_towers:
            return self._backward_towers(dq, buf, slot, param_grads, want_action_grad, side_stream)
        self._body.backward(dq.view(B, 1), slot=slot, side_stream=side_stream,
                            param_grads=param_grads,
                            input_grad=buf["dx"] if want_action_grad else None,
                            input_grad_cols=(self._obs_dim, self._obs_dim + self._act_dim))
        return buf["dx"][:, self._obs_dim:] if want_action_grad else None

    def _backward_towers(self, dq, buf, slot, param_grads, want_action_grad, side_stream):
        B = dq.shape[0]
        fo, fa = self._fo, self._fa
        need_o = self._obs_tower is not None and param_grads
        need_a = want_action_grad or (self._act_tower is not None and param_grads)
        cols = (0 if need_o else fo, fo + fa if need_a else fo)
        want_dx = cols[1] > cols[0]
        self._body.backward(dq.view(B, 1), slot=slot, side_stream=side_stream,
                            param_grads=param_grads, input_grad=buf["dx"] if want_dx else None,
                            input_grad_cols=cols if want_dx else None)
        # the towers take contiguous [B, features] gradients: both slices in one launch
        segs = []
        if need_o:
            segs.append((buf["dx"][:, :fo], buf["do"]))
        if need_a and self._act_tower is not None:
            segs.append((buf["dx"][:, fo:], buf["da_feat"]))
        if segs:
            ops.copy_segments(segs)
        if need_o:
            self._obs_tower.backward(buf["do"], slot=slot, param_grads=True)
        if not need_a:
            return None
        if self._act_tower is None:
            return buf["dx"][:, fo:]
        self._act_tower.backward(buf["da_feat"], slot=slot, param_grads=param_grads,
                                 input_grad=buf["da"] if want_action_grad else None)
        return buf["da"] if want_action_grad else None

    def call(self, inputs, step_type=None, network_state=(), training=False, **kwargs):
        obs, act = inputs
        return self.forward(obs, act, slot="call").clone(), network_state


# ---- twin critics in one launch (csrc/mlp_wide.hip through networks/sequential.py) -------------------
def pair_ok(c1, c2, observation, action):
    """True when `forward_pair` / `backward_pair` run BOTH critics per launch: their bodies share a
    layout the wide-MLP kernels take at this batch size, and (observation, action) are float32
    device tensors with unit column stride (they are read in place: no [obs | act] copy)."""
    B = int(observation.shape[0])
    o2, a2 = observation.reshape(B, -1), action.reshape(B, -1)
    return (isinstance(c1, CriticNetwork) and isinstance(c2, CriticNetwork) and
            not c1.has_towers and not c2.has_towers and c1.body.wide_ok(B) and c1.body.wide_key() is not None and
            c1.body.wide_key() == c2.body.wide_key() and
            o2.dtype == torch.float32 and a2.dtype == torch.float32 and o2.is_cuda and
            o2.stride(1) == 1 and a2.stride(1) == 1 and
            o2.shape[1] == c1._obs_dim and a2.shape[1] == c1._act_dim)


def forward_pair(c1, c2, observation, action, slot=0, need_grad=False):
    """(q1 [B], q2 [B]) of twin critics on the same (observation, action): one launch
    (sac_agent.py:286-330 evaluates both critics of a pair on the same inputs every time)."""
    B = int(observation.shape[0])
    o2, a2 = observation.reshape(B, -1), action.reshape(B, -1)
    q1, q2 = sequential.forward_wide([c1.body, c2.body], [o2, o2], slot=slot,
                                     need_grad=need_grad, x2s=[a2, a2])
    return q1.view(B), q2.view(B)


def forward_two_pairs(pair_a, obs_a, act_a, slot_a, pair_b, obs_b, act_b, slot_b,
                      need_grad_b=False):
    """Both critic pairs of SAC's critic loss in ONE launch of four networks
    (sac_agent.py:559-640: target critics on (next observation, next action), critics on
    (observation, action)): the two forwards are independent, and four networks of 64 workgroups
    fill the 256 CUs where a pair leaves half of them idle.  Returns ((qa1, qa2), (qb1, qb2));
    bit-identical to two `forward_pair` calls (every network is computed by its own workgroups)."""
    B = int(obs_a.shape[0])
    oa, aa = obs_a.reshape(B, -1), act_a.reshape(B, -1)
    ob, ab = obs_b.reshape(B, -1), act_b.reshape(B, -1)
    outs = sequential.forward_wide(
        [pair_a[0].body, pair_a[1].body, pair_b[0].body, pair_b[1].body], [oa, oa, ob, ob],
        x2s=[aa, aa, ab, ab], slots=[slot_a, slot_a, slot_b, slot_b],
        need_grads=[False, False, need_grad_b, need_grad_b])
    return (outs[0].view(B), outs[1].view(B)), (outs[2].view(B), outs[3].view(B))


def two_pairs_ok(pair_a, pair_b):
    return pair_a[0].body.wide_key() == pair_b[0].body.wide_key()


def backward_pair(c1, c2, dq1, dq2, slot=0, param_grads=True, want_action_grad=False, gen=None,
                  batch=None, adam=None):
    """`backward` of both critics of a `forward_pair`: the gradient chains in one launch, all
    weight gradients in a second one; returns (da1, da2) [B, act] views if asked.  `gen` (a filled
    `_lib.SacDoutGen` of kind CRITIC / ACTOR; then dq1 = dq2 = None and batch = B): the chain launch
    computes d loss / d q itself (the loss launch in front of it is gone)."""
    B = int(dq1.shape[0]) if gen is None else int(batch)
    dev = c1.body.flat_params.device
    bufs = [c._input(slot, B, dev) for c in (c1, c2)]
    lo, hi = c1._obs_dim, c1._obs_dim + c1._act_dim
    sequential.backward_wide([c1.body, c2.body],
                             None if gen is not None else [dq1.view(B, 1), dq2.view(B, 1)],
                             slot=slot, param_grads=param_grads,
                             input_grads=[b["dx"] for b in bufs] if want_action_grad else None,
                             input_grad_cols=(lo, hi), gen=gen, batch=B, adam=adam)
    if want_action_grad:
        return bufs[0]["dx"][:, lo:], bufs[1]["dx"][:, lo:]
    return None, None

