"""Table: one device-resident [capacity, *leaf] tensor per flattened leaf of a nested spec.

Same role and API as tf_agents/replay_buffers/table.py:32-137 (`read(rows)` / `write(rows,
values)` / `variables()`); reads and writes of ALL leaves go through one HIP launch each
(csrc/replay.hip: aa_rb_gather_rows / a row-scatter with explicit rows).  Like the reference's,
Table I/O is not thread-safe; ordering comes from the HIP stream.
"""
import ctypes

import torch

from agents_amd import _lib
from agents_amd.utils import nest_utils


class LeafPack:
    """Host-side arrays of device pointers / row sizes for one launch over all leaves."""

    def __init__(self, n):
        self.n = n
        self.tables = (ctypes.c_void_p * n)()
        self.ios = (ctypes.c_void_p * n)()
        self.row_bytes = (ctypes.c_int64 * n)()


class Table:
    def __init__(self, tensor_spec, capacity, scope="Table", device=None):
        self._tensor_spec = tensor_spec
        self._capacity = int(capacity)
        self._device = torch.device(device if device is not None else "cuda")
        self._flat_specs = nest_utils.flatten(tensor_spec)
        # zero-initialised like the reference's tf.zeros initial_value (table.py:54-77)
        self._storage = [torch.zeros((self._capacity,) + tuple(s.shape), dtype=s.dtype,
                                     device=self._device) for s in self._flat_specs]
        self._row_bytes = [s.row_bytes for s in self._flat_specs]
        if len(self._storage) > 24:
            raise ValueError("Table supports at most 24 leaves per launch (AA_MAX_LEAVES)")

    @property
    def capacity(self):
        return self._capacity

    @property
    def flat_specs(self):
        return self._flat_specs

    @property
    def row_bytes(self):
        return list(self._row_bytes)

    def variables(self):
        return list(self._storage)

    def pack(self, io_tensors):
        p = LeafPack(len(self._storage))
        for i, (tab, io) in enumerate(zip(self._storage, io_tensors)):
            p.tables[i] = tab.data_ptr()
            p.ios[i] = io.data_ptr()
            p.row_bytes[i] = self._row_bytes[i]
        return p

    def alloc_out(self, outer_shape):
        return [torch.empty(tuple(outer_shape) + tuple(s.shape), dtype=s.dtype,
                            device=self._device) for s in self._flat_specs]

    def read(self, rows, id_table=None, ids_out=None):
        """Gathers `rows` (int64 device tensor of any shape) from every leaf."""
        lib = _lib.load()
        rows = rows.contiguous()
        _lib.require_cuda(rows)
        outs = self.alloc_out(rows.shape)
        n = rows.numel()
        if n > 0:
            p = self.pack(outs)
            _lib.check(lib.aa_rb_gather_rows(
                p.tables, p.ios, p.row_bytes, p.n,
                None if id_table is None else id_table.data_ptr(),
                None if ids_out is None else ids_out.data_ptr(),
                rows.data_ptr(), n, _lib.stream_ptr()), "aa_rb_gather_rows")
        return nest_utils.pack_sequence_as(self._tensor_spec, outs)

    def check_values(self, values, batch):
        nest_utils.assert_same_structure(values, self._tensor_spec)
        flat = nest_utils.flatten(values)
        out = []
        for v, s in zip(flat, self._flat_specs):
            if not isinstance(v, torch.Tensor):
                v = torch.as_tensor(v, dtype=s.dtype, device=self._device)
            if v.device != self._device:
                v = v.to(self._device)
            if tuple(v.shape) != (batch,) + tuple(s.shape):
                raise ValueError(f"item leaf has shape {tuple(v.shape)}, expected "
                                 f"{(batch,) + tuple(s.shape)} (batch_size first)")
            if v.dtype != s.dtype:
                raise ValueError(f"item leaf has dtype {v.dtype}, spec says {s.dtype}")
            out.append(v.contiguous())
        return out
