"""Helpers the reference scripts import from tf_agents.utils.common, on HIP kernels / torch memory.

  function                     common.py:128   (tf.function wrapper -> identity here)
  create_variable              common.py:204
  soft_variables_update        common.py:250-346  -> aa_soft_update (csrc/optim.hip)
  Periodically                 common.py:450-507
  element_wise_huber_loss / element_wise_squared_loss   common.py:1199-1208
  Checkpointer                 common.py:1045-1100 (format is ours: torch.save of state dicts)
"""
import functools
import os
import pickle
import zipfile

import torch

from agents_amd import _lib


def function(*args, **kwargs):
    """`common.function` wraps callables in tf.function; kernels here are enqueued eagerly on a
    HIP stream (and the agents capture HIP graphs themselves), so this is the identity."""
    def wrap(fn):
        # `common.function(agent.train)`: the tf.function analogue here is HIP-graph capture
        owner = getattr(fn, "__self__", None)
        if owner is not None and getattr(fn, "__name__", "") == "train" and \
                (hasattr(owner, "_train_phase_grads") or hasattr(owner, "_graph_train_whole")):
            from agents_amd.utils import graph
            return graph.graphed_train(owner)
        # `common.function(driver.run)` (train_eval.py:234-237): HIP-graph replay of the loop body
        if owner is not None and getattr(fn, "__name__", "") == "run" and \
                type(owner).__name__ == "DynamicStepDriver":
            from agents_amd.utils import graph
            return graph.graphed_driver_run(owner)
        return fn

    # `common.function(fn)` and `common.function(fn, autograph=False)` (the PPO script) wrap fn;
    # `@common.function(autograph=False)` returns the decorator
    if len(args) >= 1 and callable(args[0]):
        return wrap(args[0])
    return wrap


def function_in_tf1(*args, **kwargs):
    return function(*args, **kwargs)


class Variable:
    """Minimal stand-in for a scalar tf.Variable counter (train_step_counter, global_step):
    host value with an optional device mirror."""

    def __init__(self, initial_value=0, dtype=torch.int64, name=None, device=None):
        self._value = int(initial_value)
        self.name = name
        self.dtype = dtype

    def numpy(self):
        return self._value

    def value(self):
        return self._value

    def assign(self, v):
        self._value = int(v)
        return self

    def assign_add(self, v):
        self._value += int(v)
        return self

    def __int__(self):
        return self._value

    def __index__(self):
        return self._value

    def __repr__(self):
        return f"Variable({self.name!r}, {self._value})"

    def __eq__(self, other):
        return self._value == int(other)

    def __hash__(self):
        return id(self)


def create_variable(name, initial_value=0, shape=(), dtype=torch.int64, **kwargs):
    if shape in ((), None):
        return Variable(initial_value, dtype, name)
    return torch.full(tuple(shape), initial_value, dtype=dtype,
                      device=kwargs.get("device", "cuda"))


# Loss-function sentinels: DqnAgent maps these to the fused kernel's loss_kind.
def element_wise_squared_loss(x, y):
    d = x - y
    return d * d


element_wise_squared_loss.aa_loss_kind = _lib.AA_LOSS_SQUARED


def element_wise_huber_loss(x, y):
    err = y - x
    a = err.abs()
    quad = torch.clamp(a, max=1.0)
    return 0.5 * quad * quad + (a - quad)


element_wise_huber_loss.aa_loss_kind = _lib.AA_LOSS_HUBER


def soft_variables_update(source_variables, target_variables, tau=1.0, tau_non_trainable=None,
                          sort_variables_by_name=False):
    """w_t = (1 - tau) * w_t + tau * w_s for every pair (common.py:250-346); tau == 1 copies.

    `source_variables` / `target_variables` may be flat fp32 buffers (one launch for the whole
    model) or lists of tensors."""
    if tau < 0 or tau > 1:
        raise ValueError("Input `tau` should be in [0, 1].")
    if tau == 0.0:
        return
    if isinstance(source_variables, torch.Tensor):
        source_variables, target_variables = [source_variables], [target_variables]
    if len(source_variables) != len(target_variables):
        raise ValueError("Source and target variable lists have different lengths: "
                         f"{len(source_variables)} vs. {len(target_variables)}")
    lib = _lib.load()
    for s, t in zip(source_variables, target_variables):
        if tuple(s.shape) != tuple(t.shape):
            raise ValueError("source / target variable shapes differ")
        if tau == 1.0:
            t.copy_(s)
        else:
            _lib.require_cuda(s, t)
            _lib.check(lib.aa_soft_update(t.data_ptr(), s.data_ptr(), t.numel(), float(tau),
                                          _lib.stream_ptr()), "aa_soft_update")


def weak_method(bound):
    """`bound` (a bound method) as a callable that does not keep its object alive: for callbacks an
    object stores on ITSELF (agent._update_target = Periodically(agent._soft_update...)), where the
    bound method would close a reference cycle and leave the object -- with its device memory and
    HIP graphs -- to the cyclic collector."""
    import weakref
    ref = weakref.WeakMethod(bound)

    def call(*args, **kwargs):
        return ref()(*args, **kwargs)
    return call


class Periodically:
    """Runs `body` every `period` calls (common.py:450-507): period None -> never, 1 -> always,
    else when the call count is a positive multiple of period."""

    def __init__(self, body, period, name="periodically"):
        if not callable(body):
            raise TypeError("body must be callable.")
        self._body, self._period, self._counter = body, period, 0
        self.name = name

    def __call__(self):
        if self._period is None:
            return None
        if self._period == 1:
            return self._body()
        self._counter += 1
        if self._counter % int(self._period) == 0:
            return self._body()
        return None


class Checkpointer:
    """Saves/restores objects exposing state_dict()/load_state_dict() (agent, replay buffer,
    optimizer) plus plain values; restores the latest checkpoint in the constructor like the
    reference's (common.py:1045-1100).  On-disk format is torch.save of tensors / numbers /
    containers (loaded with weights_only=True: a checkpoint directory is data, not code).

    Data-parallel runs: every rank constructs the Checkpointer on the same directory and calls
    `save` at the same step; rank 0 writes (to a temporary file, then an atomic rename -- a crash
    mid-write never leaves a truncated `ckpt-N.pt`) and prunes, the others wait at a barrier.  A
    checkpoint FILE that cannot be read (truncated, not a torch archive) is skipped in favour of
    the previous one; an error while APPLYING a readable checkpoint (`load_state_dict`: a shape
    that no longer fits, a bug) propagates -- falling back silently would train from older weights
    or from scratch and later prune the good checkpoints, and a failure part-way would leave some
    objects restored from one file and the rest from another.  If files exist and none is readable
    the constructor raises instead of starting from scratch."""

    def __init__(self, ckpt_dir, max_to_keep=20, **kwargs):
        self._dir = ckpt_dir
        self._max_to_keep = max_to_keep
        self._objects = kwargs
        os.makedirs(ckpt_dir, exist_ok=True)
        self.checkpoint_exists = False
        self.restored_from = None
        files = self._list()
        unreadable = []
        for fname in reversed(files):
            try:
                blob = torch.load(os.path.join(self._dir, fname), weights_only=True)
            except (OSError, EOFError, RuntimeError, ValueError, KeyError, zipfile.BadZipFile,
                    pickle.UnpicklingError) as e:
                # the FILE is bad (truncated / foreign): the previous one may still be whole
                import warnings
                warnings.warn(f"Checkpointer: could not read {fname} ({type(e).__name__}: {e}); "
                              "trying the previous checkpoint")
                unreadable.append(fname)
                continue
            if not isinstance(blob, dict):
                unreadable.append(fname)
                continue
            self._apply(blob)           # errors here propagate (see the class docstring)
            self.checkpoint_exists = True
            self.restored_from = fname
            break
        if files and not self.checkpoint_exists:
            raise RuntimeError(
                f"Checkpointer: {len(files)} checkpoint file(s) in {ckpt_dir} and none could be "
                f"read ({', '.join(unreadable)}); refusing to start from scratch over them")

    def _list(self):
        fs = [f for f in os.listdir(self._dir) if f.startswith("ckpt-") and f.endswith(".pt")
              and f[5:-3].isdigit()]
        return sorted(fs, key=lambda f: int(f[5:-3]))

    @staticmethod
    def _dist():
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist
        return None

    def save(self, global_step):
        step = int(global_step)
        dist = self._dist()
        if dist is None or dist.get_rank() == 0:
            blob = {}
            for k, o in self._objects.items():
                blob[k] = o.state_dict() if hasattr(o, "state_dict") else (
                    int(o) if isinstance(o, Variable) else o)
            final = os.path.join(self._dir, f"ckpt-{step}.pt")
            tmp = final + f".tmp{os.getpid()}"
            torch.save(blob, tmp)
            os.replace(tmp, final)
            fs = self._list()
            while self._max_to_keep and len(fs) > self._max_to_keep:
                os.remove(os.path.join(self._dir, fs.pop(0)))
        if dist is not None:
            dist.barrier()

    def _restore(self, fname):
        self._apply(torch.load(os.path.join(self._dir, fname), weights_only=True))

    def _apply(self, blob):
        for k, o in self._objects.items():
            if k not in blob:
                continue
            if hasattr(o, "load_state_dict"):
                o.load_state_dict(blob[k])
            elif isinstance(o, Variable):
                o.assign(blob[k])

    def initialize_or_restore(self, session=None):
        return None
