"""pytest configuration: `-m gpu` tests need an MI355X; everything else runs on CPU."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a HIP device (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """The canary (tests/test_gpu_00_canary.py) runs before every other test, whatever the file
    order: with `-x` a box that cannot do a torch H2D copy then fails there, not in a parity test."""
    first = [it for it in items if "test_gpu_00_canary" in it.nodeid]
    if first:
        rest = [it for it in items if "test_gpu_00_canary" not in it.nodeid]
        items[:] = first + rest


@pytest.fixture(scope="session")
def lib():
    """The built C-ABI library (built in-tree if missing; hipcc cross-compiles without a GPU)."""
    from agents_amd import _build, _lib
    if not os.path.exists(_lib.LIB_PATH):
        _build.build(verbose=False)
    return _lib.load()


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda", 0)


@pytest.fixture(autouse=True, scope="module")
def _graph_stats(request):
    """AA_TEST_GRAPH_STATS=1: after every test module, one stderr line with the HIP graphs this
    process holds (recorded and not closed, closed and waiting for an idle device, recorded in
    all) -- the bookkeeping behind tests/test_gpu_lifetime.py, for hunting a leak across a run."""
    yield
    if os.environ.get("AA_TEST_GRAPH_STATS") == "1" and "agents_amd.utils.graph" in sys.modules:
        g = sys.modules["agents_amd.utils.graph"]
        live, parked = g.live_graphs()
        print(f"\n[graph stats] after {request.module.__name__}: live {live}, parked {parked}, "
              f"recorded so far {g.capture_count()}", file=sys.stderr, flush=True)
