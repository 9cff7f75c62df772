"""GPU: the guard against the hipGraphLaunch fault of this image's HIP runtime (version 70051831).

csrc/runtime_guard.hip / utils/graph.py (`_spread_exec`) have the analysis: a recorded graph with
side branches gets n internal parallel streams at instantiation, each on the hardware queue with
the fewest streams; hipGraphLaunch reads past their list when two of them share the launch
stream's queue.  This file provokes exactly that placement (raw HIP streams created and destroyed
so that one queue is under-loaded), shows with the guard OFF that a fresh exec then has two
parallel streams on one queue -- without launching it -- and with the guard ON that the recorded
graph's streams are spread and that it replays from the default stream and from a dozen others."""
import ctypes

import pytest
import torch

from agents_amd import _lib
from agents_amd.utils import graph

pytestmark = pytest.mark.gpu


def _hip():
    return ctypes.CDLL(torch.__file__.rsplit("/", 1)[0] + "/lib/libamdhip64.so")


def _raw_stream(hip):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    return s


def _wide(a, side):
    """A graph body with len(side) side branches (fork / join by events, like Sequential.backward)."""
    main = torch.cuda.current_stream()
    a.add_(1.0)
    outs = []
    for s in side:
        s.wait_stream(main)
        with torch.cuda.stream(s):
            outs.append(a * 2.0)
    for s in side:
        main.wait_stream(s)
    a.add_(1.0)
    return outs


def test_recorded_graphs_have_their_parallel_streams_on_different_queues(dev, monkeypatch):
    lib = _lib.load()
    n, worst = ctypes.c_int32(0), ctypes.c_int32(0)
    probe = graph._Captured("probe")
    a = torch.zeros(1024, device=dev)
    side = [torch.cuda.Stream() for _ in range(3)]
    with torch.cuda.device(dev):
        probe.capture(lambda: _wide(a, side))
        if probe.spread is None:
            pytest.skip("HIP runtime is not version 70051831: the guard does not apply")
        assert probe.spread[0] >= 3 and probe.spread[1] <= 1
        probe.close()
        graph.release_dead()
        # ---- provoke: four raw streams land one per queue; destroying ONE leaves its queue a
        # stream short of the others.  Which phase makes the greedy placement double up depends on
        # the loads we cannot see: try the four phases, guard off, never launching the probe exec
        hip = _hip()
        monkeypatch.setattr(graph, "EXEC_GUARD", False)
        hazard, keep = None, []
        for phase in range(8):
            four = [_raw_stream(hip) for _ in range(4)]
            hip.hipStreamDestroy(four[phase % 4])
            keep += [s for i, s in enumerate(four) if i != phase % 4]
            c = graph._Captured("probe")
            c.capture(lambda: _wide(a, side))
            spread = c.spread
            c.close()
            graph.release_dead()
            if spread[1] >= 2:
                hazard = spread
                break
        if hazard is None:
            for s in keep:
                hip.hipStreamDestroy(s)
            pytest.skip("could not provoke an uneven queue load on this box")
        # ---- the same process state, guard on: the exec that is kept is spread
        monkeypatch.setattr(graph, "EXEC_GUARD", True)
        before = dict(graph.exec_guard_stats)
        c = graph._Captured("guarded")
        c.capture(lambda: _wide(a, side))
        assert c.spread[0] == hazard[0] and c.spread[1] <= 1, (hazard, c.spread)
        assert graph.exec_guard_stats["respread"] == before["respread"] + 1
        assert lib.aa_hip_graph_exec_spread(c.exec.ptr, ctypes.byref(n),
                                            ctypes.byref(worst)) == 0 and worst.value <= 1
        a.zero_()
        c.replay()                                   # the default stream
        for st in [torch.cuda.Stream() for _ in range(12)]:
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                c.replay()
            torch.cuda.current_stream().wait_stream(st)
        torch.cuda.synchronize()
        assert float(a[0]) == 2.0 * 13
        c.close()
        for s in keep:
            hip.hipStreamDestroy(s)
        graph.release_dead()


def test_linear_graphs_need_no_parallel_streams(dev):
    c = graph._Captured("linear")
    a = torch.zeros(16, device=dev)
    with torch.cuda.device(dev):
        c.capture(lambda: a.add_(1.0))
        if c.spread is None:
            pytest.skip("HIP runtime is not version 70051831")
        assert c.spread == (1, 0)
        c.replay()
        torch.cuda.synchronize()
        assert float(a[0]) == 1.0
        c.close()
