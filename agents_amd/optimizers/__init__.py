"""Fused flat-buffer optimizers (csrc/optim.hip) behind a keras-like surface.

The reference's scripts hand the agent a TF optimizer object (tf.compat.v1.train.AdamOptimizer in
agents/dqn/examples/v2/train_eval.py:180; keras RMSprop(rho=.95, momentum=.95, epsilon=.01,
centered=True) in examples/dqn/mnih15/dqn_train_eval_atari.py:176-182; Adam(eps=1e-5) for PPO).
These classes take the same hyper-parameters and expose `apply_gradients(grads_and_vars)` and
`variables()`.  Update arithmetic is documented in include/agents_amd.h; slots live in flat fp32
buffers parallel to the network's flat parameter buffer so one launch updates the whole model.
"""
import torch

from agents_amd import _lib
from agents_amd.utils import graph


class Optimizer:
    def __init__(self, name):
        self._name = name
        self._slots = {}       # id(flat param storage) -> dict of slot tensors
        self.iterations = 0    # host mirror of the device step counter
        self._pending = []     # restored slot sets not yet claimed (load_state_dict)

    def _bump_iterations(self):
        self.iterations += 1

    # -- flat API (used by agents) -------------------------------------------------------------
    # True: apply_flat(..., planes=<_lib.PlaneScatter>) also keeps the bf16 filter planes of the
    # bf16x6 convolutions current (networks/sequential.py: plane_scatter)
    supports_planes = False

    def apply_flat(self, params, grads, planes=None):
        """One fused update of the flat fp32 `params` with `grads` (same layout)."""
        raise NotImplementedError

    def _slot(self, params, names):
        key = params.data_ptr()
        s = self._slots.get(key)
        if s is None:
            s = {n: torch.zeros_like(params) for n in names}
            s["step"] = torch.zeros((1,), dtype=torch.int64, device=params.device)
            self._slots[key] = s
            if self._pending:   # restored state waiting for this (creation-ordered) slot set
                for k, v in self._pending.pop(0).items():
                    s[k].copy_(v)
        return s

    # -- keras-like API ----------------------------------------------------------------------------
    def apply_gradients(self, grads_and_vars, name=None):
        """Per-variable update; each (grad, var) pair must be fp32 device tensors of one shape.
        Variables that are consecutive views of one flat buffer should use `apply_flat`."""
        for g, v in grads_and_vars:
            if g is None:
                continue
            self.apply_flat(v.reshape(-1), g.reshape(-1))
        return None

    def variables(self):
        out = []
        for s in self._slots.values():
            out += list(s.values())
        return out

    def state_dict(self):
        return {"iterations": self.iterations,
                "slots": [{k: v.clone() for k, v in s.items()} for s in self._slots.values()]}

    def load_state_dict(self, sd):
        """Restores the step count and the slot tensors IN PLACE (captured HIP graphs keep pointing
        at them).  Slot sets are matched by creation order; sets that do not exist yet (the
        optimizer has not been applied since construction) are filled when they are created."""
        self.iterations = int(sd["iterations"])
        saved = list(sd["slots"])
        for s in self._slots.values():
            if not saved:
                break
            for k, v in saved.pop(0).items():
                s[k].copy_(v)
        self._pending = saved


class Adam(Optimizer):
    """TF ApplyAdam arithmetic; epsilon defaults to keras' 1e-7 (tf.compat.v1 uses 1e-8)."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, name="Adam"):
        super().__init__(name)
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = \
            float(learning_rate), float(beta_1), float(beta_2), float(epsilon)
        self._arrive = {}

    supports_planes = True

    supports_soft_target = True

    def apply_flat(self, params, grads, planes=None, soft_target=None):
        """soft_target = (target_flat, tau): the same launch also moves `target_flat` towards the
        updated parameters (soft_variables_update; SAC's target critics)."""
        import ctypes
        lib = _lib.load()
        _lib.require_cuda(params, grads)
        s = self._slot(params, ("m", "v"))
        key = params.data_ptr()
        arrive = self._arrive.get(key)
        if arrive is None:      # scratch of the in-launch step counter (not optimizer state)
            arrive = self._arrive[key] = torch.zeros((16,), dtype=torch.int64,
                                                     device=params.device)
        if soft_target is not None:
            if planes is not None:
                raise ValueError("soft_target and planes are not combined")
            target, tau = soft_target
            _lib.require_cuda(target)
            if target.numel() != params.numel() or target.dtype != torch.float32:
                raise ValueError("soft_target must mirror the parameter buffer")
            _lib.check(lib.aa_adam_step_counted_target(
                params.data_ptr(), grads.data_ptr(), s["m"].data_ptr(), s["v"].data_ptr(),
                params.numel(), self.learning_rate, self.beta_1, self.beta_2, self.epsilon,
                s["step"].data_ptr(), arrive.data_ptr(), target.data_ptr(), float(tau),
                _lib.stream_ptr()), "aa_adam_step_counted_target")
            graph.on_replay(self._bump_iterations)
            return
        # s["step"] = steps applied so far; the launch uses t = step + 1 and stores it back itself
        _lib.check(lib.aa_adam_step_counted(
            params.data_ptr(), grads.data_ptr(), s["m"].data_ptr(), s["v"].data_ptr(),
            params.numel(), self.learning_rate, self.beta_1, self.beta_2, self.epsilon,
            s["step"].data_ptr(), arrive.data_ptr(),
            None if planes is None else ctypes.byref(planes), _lib.stream_ptr()),
            "aa_adam_step_counted")
        graph.on_replay(self._bump_iterations)


    def fused_step_desc(self, params, soft_target=None):
        """What `apply_flat(params, grads, soft_target=...)` would launch, as the descriptor a
        weight-gradient launch takes to do it itself (`_lib.MlpWideAdam` with the BASE pointers of
        the flat buffers in slot 0: the caller fills the per-network offsets), plus the host
        bookkeeping of a step (`graph.on_replay(self._bump_iterations)`): call once per step."""
        _lib.require_cuda(params)
        s = self._slot(params, ("m", "v"))
        key = params.data_ptr()
        arrive = self._arrive.get(key)
        if arrive is None:
            arrive = self._arrive[key] = torch.zeros((16,), dtype=torch.int64,
                                                     device=params.device)
        a = _lib.MlpWideAdam()
        a.p[0], a.m[0], a.v[0] = params.data_ptr(), s["m"].data_ptr(), s["v"].data_ptr()
        if soft_target is not None:
            target, tau = soft_target
            _lib.require_cuda(target)
            if target.numel() != params.numel() or target.dtype != torch.float32:
                raise ValueError("soft_target must mirror the parameter buffer")
            a.target[0], a.tau = target.data_ptr(), float(tau)
        a.lr, a.beta1, a.beta2, a.eps = self.learning_rate, self.beta_1, self.beta_2, self.epsilon
        a.step_dev, a.arrival_dev = s["step"].data_ptr(), arrive.data_ptr()
        graph.on_replay(self._bump_iterations)
        return a


class AdamOptimizer(Adam):
    """tf.compat.v1.train.AdamOptimizer signature (epsilon 1e-8)."""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8,
                 use_locking=False, name="Adam"):
        super().__init__(learning_rate, beta1, beta2, epsilon, name)


class RMSprop(Optimizer):
    def __init__(self, learning_rate=0.001, rho=0.9, momentum=0.0, epsilon=1e-7, centered=False,
                 name="RMSprop"):
        super().__init__(name)
        self.learning_rate, self.rho, self.momentum, self.epsilon, self.centered = \
            float(learning_rate), float(rho), float(momentum), float(epsilon), bool(centered)

    supports_planes = True
    supports_grad_slabs = True
    supports_pack = True       # apply_flat(grad_slabs=..., pack=...) also copies a few scalars

    def apply_flat(self, params, grads, planes=None, grad_slabs=None, pack=None):
        """grad_slabs: an aa_grad_slabs (`Sequential.take_grad_slabs()`): those parameter ranges
        take their gradient from unsummed split-K slabs instead of `grads`.
        pack = (ctypes array of <= 8 device pointers to fp32 scalars, destination tensor): the
        launch also copies those scalars side by side into the destination (with grad_slabs
        only) -- the LossInfo sums Learner.run returns, without a copy launch of their own."""
        import ctypes
        lib = _lib.load()
        _lib.require_cuda(params, grads)
        names = ["ms"] + (["mg"] if self.centered else []) + (["mom"] if self.momentum > 0 else [])
        s = self._slot(params, names)
        if pack is not None and grad_slabs is not None:
            src, dst = pack
            _lib.check(lib.aa_rmsprop_step_slabs_pack(
                params.data_ptr(), grads.data_ptr(), s["ms"].data_ptr(),
                s["mg"].data_ptr() if self.centered else None,
                s["mom"].data_ptr() if self.momentum > 0 else None, params.numel(),
                self.learning_rate, self.rho, self.momentum, self.epsilon,
                None if planes is None else ctypes.byref(planes), ctypes.byref(grad_slabs),
                src, len(src), dst.data_ptr(), _lib.stream_ptr()), "aa_rmsprop_step_slabs_pack")
            graph.on_replay(self._bump_iterations)
            return
        if pack is not None:
            raise ValueError("apply_flat(pack=...) needs grad_slabs")
        _lib.check(lib.aa_rmsprop_step_slabs(
            params.data_ptr(), grads.data_ptr(), s["ms"].data_ptr(),
            s["mg"].data_ptr() if self.centered else None,
            s["mom"].data_ptr() if self.momentum > 0 else None, params.numel(),
            self.learning_rate, self.rho, self.momentum, self.epsilon,
            None if planes is None else ctypes.byref(planes),
            None if grad_slabs is None else ctypes.byref(grad_slabs), _lib.stream_ptr()),
            "aa_rmsprop_step_slabs")
        graph.on_replay(self._bump_iterations)


class SGD(Optimizer):
    def __init__(self, learning_rate=0.01, name="SGD"):
        super().__init__(name)
        self.learning_rate = float(learning_rate)

    def apply_flat(self, params, grads, planes=None):
        lib = _lib.load()
        _lib.require_cuda(params, grads)
        _lib.check(lib.aa_sgd_step(params.data_ptr(), grads.data_ptr(), params.numel(),
                                   self.learning_rate, _lib.stream_ptr()), "aa_sgd_step")
        graph.on_replay(self._bump_iterations)
