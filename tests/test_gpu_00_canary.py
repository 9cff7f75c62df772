"""Collected FIRST among the `-m gpu` tests (file name sorts before every other test file): tells a
broken box from a broken tree.  GPUTEST_r03 died in the first GPU test on a plain
`torch.as_tensor(numpy, device=dev)` with "Memory access fault by GPU node" before any in-tree
kernel had been launched; with `-x` such a box now fails HERE, in pure torch, and a red record is
attributable.

1. pure torch, BEFORE the in-tree library is loaded: allocate, H2D copy of a numpy array (both
   spellings), one elementwise op, D2H, synchronize;
2. load `libagents_amd.so` and launch only its two trivial kernels (`aa_marker`, `aa_counter_add`);
3. the H2D path again after the library's code objects are resident.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _h2d_roundtrip(torch, dev, n):
    a = np.arange(n, dtype=np.float32)
    t1 = torch.as_tensor(a, device=dev)
    t2 = torch.from_numpy(a).to(dev)
    torch.cuda.synchronize()
    s = (t1 + t2) * 0.5
    torch.cuda.synchronize()
    assert np.array_equal(s.cpu().numpy(), a)
    b = np.arange(n, dtype=np.uint8)
    assert np.array_equal(torch.as_tensor(b, device=dev).cpu().numpy(), b)


def test_00_pure_torch_before_the_library_is_loaded(dev):
    import sys

    import torch
    print(f"[canary] torch {torch.__version__}, device {torch.cuda.get_device_name(0)}, "
          f"agents_amd._lib imported: {'agents_amd._lib' in sys.modules}", flush=True)
    z = torch.zeros(1 << 20, device=dev)
    torch.cuda.synchronize()
    assert float(z.sum().item()) == 0.0
    for n in (7, 4096, 1 << 20):
        _h2d_roundtrip(torch, dev, n)
    print("[canary] pure torch alloc / H2D / elementwise / D2H ok", flush=True)


def test_01_library_loads_and_trivial_kernels_launch(dev, lib):
    import torch

    from agents_amd import _lib
    assert lib.aa_abi_version() >= 11
    torch.cuda.synchronize()
    stream = _lib.stream_ptr()
    _lib.check(lib.aa_marker(0, stream), "aa_marker")
    torch.cuda.synchronize()
    c = torch.zeros(1, dtype=torch.int64, device=dev)
    for _ in range(3):
        _lib.check(lib.aa_counter_add(c.data_ptr(), 5, stream), "aa_counter_add")
    torch.cuda.synchronize()
    assert int(c.item()) == 15
    print("[canary] libagents_amd.so loaded, aa_marker / aa_counter_add ok", flush=True)


def test_02_h2d_after_the_library_is_resident(dev, lib):
    import torch
    for n in (33, 1 << 16):
        _h2d_roundtrip(torch, dev, n)
