"""Device-resident synthetic vector environment (one HIP launch per batched step).

Plays the role of tf_agents/environments/random_tf_environment.py:129 (an in-graph batched
random env) for the hot path, where the reference pays a device->host->Python hop per step
through TFPyEnvironment's tf.numpy_function (tf_py_environment.py:296-326).  Differences from the
reference fake, chosen to behave like real batched envs (BatchedPyEnvironment resets sub-envs
independently, batched_py_environment.py:155-180): every env ends its episode independently with
probability `episode_end_probability`, and an env whose current step is LAST ignores the action
and restarts (py_environment.py:233-239).  Observations / rewards come from the package's Philox
stream (csrc/rollout.hip documents the counter layout; oracle/env.py restates it).
"""
import numpy as np
import torch

from agents_amd import _lib
from agents_amd.utils import graph
from agents_amd.environments import tf_environment
from agents_amd.specs import tensor_spec
from agents_amd.trajectories import time_step as ts


class RandomTFEnvironment(tf_environment.TFEnvironment):
    def __init__(self, time_step_spec, action_spec, batch_size=1, episode_end_probability=0.1,
                 seed=0, device=None):
        super().__init__(time_step_spec, action_spec, batch_size)
        obs_spec = time_step_spec.observation
        if not isinstance(obs_spec, tensor_spec.TensorSpec):
            raise NotImplementedError("RandomTFEnvironment takes a single-tensor observation")
        if obs_spec.dtype == torch.uint8:
            self._obs_kind, self._lo, self._hi = _lib.AA_OBS_U8, 0.0, 255.0
        elif obs_spec.dtype == torch.float32:
            self._obs_kind = _lib.AA_OBS_F32
            if isinstance(obs_spec, tensor_spec.BoundedTensorSpec):
                self._lo = float(np.asarray(obs_spec.minimum).reshape(-1)[0])
                self._hi = float(np.asarray(obs_spec.maximum).reshape(-1)[0])
            else:
                self._lo, self._hi = -1.0, 1.0
        else:
            raise NotImplementedError("observations must be uint8 or float32")
        self._obs_spec = obs_spec
        self._p_end = float(episode_end_probability)
        self._seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._device = torch.device(device) if device is not None else torch.device("cuda")
        self._step_counter = torch.zeros((1,), dtype=torch.int64, device=self._device)
        # arrival counters of the step kernel's workgroups (9 x one 128-byte line; the last
        # workgroup to arrive advances the step counter inside the launch)
        self._arrival = torch.zeros((144,), dtype=torch.int64, device=self._device)
        self._time_step = None
        self._ring = None
        # Bumped by every step / reset / restore issued from Python (not by HIP-graph replays of a
        # captured step): a graphed driver uses it to tell whether the time step it last saw is
        # still the environment's (agents_amd/utils/graph.py: GraphedDriverRun).
        self.host_epoch = 0

    def _alloc(self):
        B = self._batch_size
        dev = self._device
        return ts.TimeStep(
            step_type=torch.empty((B,), dtype=torch.int32, device=dev),
            reward=torch.empty((B,), dtype=torch.float32, device=dev),
            discount=torch.empty((B,), dtype=torch.float32, device=dev),
            observation=torch.empty((B,) + tuple(self._obs_spec.shape),
                                    dtype=self._obs_spec.dtype, device=dev))

    def graph_ring(self):
        """Switches the environment to two alternating output buffers (instead of a fresh
        TimeStep per step) and returns the ring: what HIP-graph replay of a driver loop body
        needs, since a captured step reads and writes fixed addresses.  A TimeStep handed out by
        `step()` then stays valid for one further step."""
        if self._ring is None:
            self._ring = _TimeStepRing([self._alloc(), self._alloc()])
        return self._ring

    def _next_out(self):
        if self._ring is None:
            return self._alloc()
        cur = self._ring.slot_of(self._time_step) if self._time_step is not None else None
        return self._ring.slots[1 - cur] if cur is not None else self._ring.slots[0]

    def _set_time_step(self, out):
        self._time_step = out

    def _launch(self, cur_step_type, force_first):
        lib = _lib.load()
        graph.join_lanes(self._device)
        if not graph.capturing():
            self.host_epoch += 1
        out = self._next_out()
        with torch.cuda.device(self._device):
            st = _lib.stream_ptr()
            _lib.check(lib.aa_vecenv_random_step(
                None if cur_step_type is None else cur_step_type.data_ptr(), self._batch_size,
                self._obs_spec.num_elements, self._obs_kind, self._lo, self._hi, self._p_end,
                self._seed, self._step_counter.data_ptr(), self._arrival.data_ptr(),
                1 if force_first else 0, out.step_type.data_ptr(), out.reward.data_ptr(),
                out.discount.data_ptr(), out.observation.data_ptr(), st),
                "aa_vecenv_random_step")
        return out

    def state_dict(self):
        graph.join_lanes(self._device)
        return {"step_counter": int(self._step_counter[0].item()),
                "time_step": None if self._time_step is None else
                tuple(t.clone() for t in self._time_step)}

    def load_state_dict(self, sd):
        graph.join_lanes(self._device)
        self._step_counter[0] = int(sd["step_counter"])
        self.host_epoch += 1
        if sd["time_step"] is not None:
            if self._time_step is None:
                self._time_step = self._next_out()
            for dst, src in zip(self._time_step, sd["time_step"]):
                dst.copy_(src)

    def _current_time_step(self):
        graph.join_lanes(self._device)
        if self._time_step is None:
            self._time_step = self._reset()
        return self._time_step

    def _reset(self):
        self._time_step = self._launch(None, True)
        return self._time_step

    def _step(self, action):
        if self._time_step is None:
            return self._reset()
        out = self._launch(self._time_step.step_type, False)
        # under HIP-graph capture the reference to the current step must move on every replay
        graph.on_replay(lambda: self._set_time_step(out))
        if graph.capturing():
            self._time_step = out     # so that a second step captured in the same graph chains
        return out


class _TimeStepRing:
    """Two preallocated TimeSteps the environment alternates between."""

    def __init__(self, slots):
        self.slots = slots
        self._ptrs = [ts_.observation.data_ptr() for ts_ in slots]

    def slot_of(self, time_step):
        if time_step is None:
            return None
        try:
            return self._ptrs.index(time_step.observation.data_ptr())
        except ValueError:
            return None
