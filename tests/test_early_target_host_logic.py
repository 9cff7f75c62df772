"""CPU: the host-side bookkeeping of the early target forward (utils/graph.py: GraphedTrain).

The device side is pinned bit for bit by tests/test_gpu_early_target.py; this file pins, without a
GPU, WHEN an early forward is issued and what it is ordered behind: the successor prediction, the
events the lane waits for (gradient graph of the step in flight, a target update behind it, the draw
that filled the next ring slot), and every reason not to issue one.  Streams, events and graphs are
stand-ins that record what was asked of them."""
import types

import pytest

from agents_amd.utils import graph


class _Ev:
    n = 0

    def __init__(self, tag):
        _Ev.n += 1
        self.tag, self.id = tag, _Ev.n


class _Stream:
    def __init__(self, name):
        self.name, self.waits = name, []

    def wait_event(self, ev):
        self.waits.append(ev)


class _Lanes:
    def __init__(self):
        self.S = _Stream("S")
        self.ready, self.ready_seq = {}, {}
        self.aux_done = None
        self.recorded = []

    def event_on(self, stream):
        ev = _Ev(("on", stream.name))
        self.recorded.append(ev)
        return ev


class _Graph:
    def __init__(self):
        self.replays = 0

    def replay(self):
        self.replays += 1

    def close(self):
        pass


class _Agent:
    """What GraphedTrain needs from an agent to consider early target forwards."""

    def __init__(self):
        self.side = _Stream("side")
        self.key = (0, 0, 0)

    def train(self, experience, weights=None):     # GraphedTrain binds type(agent).train
        raise AssertionError("not used")

    def _train_phase_grads(self, *a, **k): ...
    def _train_phase_apply(self): ...
    def _train_phase_host(self): ...
    def _train_phase_target(self, experience): ...

    def _early_target_key(self):
        return self.key

    def _side_stream(self, dev):
        return self.side


def _entry(ptr0, with_target=True):
    e = graph._Entry()
    e.ptr0 = ptr0
    e.g_target = _Graph() if with_target else None
    return e


@pytest.fixture
def gt(monkeypatch):
    monkeypatch.setattr(graph.torch.cuda, "set_stream", lambda s: None)
    monkeypatch.setattr(graph, "EARLY_TARGET", "side")
    agent = _Agent()      # the agent owns its GraphedTrain; the pointer back is weak
    yield graph.GraphedTrain(agent)


def test_successor_is_learned_and_forward_is_issued_behind_its_dependencies(gt):
    lanes, cur = _Lanes(), _Stream("main")
    a, b = _entry(100), _entry(200)
    lanes.ready[200], lanes.ready_seq[200] = _Ev("draw b"), 7
    g1 = _Ev("grads a")
    gt._issue_early_target(a, lanes, None, cur, g1, False)
    assert gt._early is None and gt.early_issued == 0          # nothing known about what follows a
    gt._issue_early_target(b, lanes, None, cur, _Ev("grads b"), False)
    assert gt._succ[a] is b and gt._early is None              # ... now it is: a -> b
    g3 = _Ev("grads a again")
    gt._issue_early_target(a, lanes, None, cur, g3, False)
    side = gt.agent.side
    # on the agent's side stream, behind this step's gradient graph and the draw of b's slot
    assert side.waits == [g3, lanes.ready[200]] and b.g_target.replays == 1
    nxt, done, key, seq = gt._early[:4]
    assert nxt is b and key == gt.agent.key and seq == 7 and lanes.aux_done is done
    assert done.tag == ("on", "side") and gt.early_issued == 1


def test_target_update_behind_the_step_orders_the_forward_behind_it(gt):
    lanes, cur = _Lanes(), _Stream("main")
    a, b = _entry(1), _entry(2)
    gt._succ[a] = b
    lanes.ready[2], lanes.ready_seq[2] = _Ev("draw"), 1
    g = _Ev("grads")
    gt._issue_early_target(a, lanes, None, cur, g, True)
    waits = gt.agent.side.waits
    assert waits[0] is g and waits[1].tag == ("on", "main") and waits[2] is lanes.ready[2]


@pytest.mark.parametrize("why", ["off", "no lanes", "no gradient event", "no target graph",
                                 "slot never drawn", "unknown successor"])
def test_reasons_not_to_issue(gt, monkeypatch, why):
    lanes, cur = _Lanes(), _Stream("main")
    a, b = _entry(1), _entry(2, with_target=(why != "no target graph"))
    if why != "unknown successor":
        gt._succ[a] = b
    if why != "slot never drawn":
        lanes.ready[2], lanes.ready_seq[2] = _Ev("draw"), 1
    if why == "off":
        monkeypatch.setattr(graph, "EARLY_TARGET", "0")
    gt._issue_early_target(a, None if why == "no lanes" else lanes, None, cur,
                           None if why == "no gradient event" else _Ev("grads"), False)
    assert gt._early is None and gt.early_issued == 0 and gt.agent.side.waits == []
    assert gt._prev_entry is a            # the order of the calls is recorded all the same


def test_stream_choice(gt, monkeypatch):
    lanes = _Lanes()
    assert gt._early_target_stream(lanes, None) is gt.agent.side
    gt.agent._side_stream = lambda dev: None       # AA_TRAIN_SINGLE_STREAM=1: no side stream
    assert gt._early_target_stream(lanes, None) is lanes.S


def test_early_mark_only_for_agents_with_a_target_phase(gt):
    lanes, cur = _Lanes(), _Stream("main")
    assert gt._early_mark(lanes, cur).tag == ("on", "main")
    assert gt._early_mark(None, cur) is None
    plain = types.SimpleNamespace(train=lambda *a, **k: None)
    other = graph.GraphedTrain.__new__(graph.GraphedTrain)
    other._agent = plain
    assert other._early_mark(lanes, cur) is None


def test_eager_fallbacks_wait_for_a_pending_early_forward(gt, monkeypatch):
    """Advisor finding (round 4): the early returns to the eager train path left a pending early
    target forward neither waited for nor dropped; every one of them now orders the caller's stream
    behind it first, independent of what the agent's eager path joins."""
    cur = _Stream("main")
    monkeypatch.setattr(graph.torch.cuda, "current_stream", lambda *a: cur)
    calls = []
    gt._eager_train = lambda experience, weights=None, **kw: calls.append(kw) or "eager"
    done = _Ev("early done")
    gt._early = (_entry(1), done, gt.agent.key, 1)
    assert gt("exp", some_kwarg=1) == "eager"            # kwargs -> eager path
    assert gt._early is None and cur.waits == [done]
    gt._early = (_entry(1), done, gt.agent.key, 1)
    gt.agent.check_numerics = True
    assert gt("exp") == "eager" and cur.waits == [done, done] and gt._early is None


def test_entry_captured_under_another_hook_state_is_not_replayed(gt, monkeypatch):
    """Advisor finding (round 4): `_optimizer_sums_slabs()` is evaluated while the gradient phase
    is recorded; a gradient hook installed afterwards must not meet that graph."""
    cur = _Stream("main")
    monkeypatch.setattr(graph.torch.cuda, "current_stream", lambda *a: cur)
    state = {"hook": None}
    gt.agent._graph_capture_key = lambda: (state["hook"] is None,)
    e = _entry(5)
    e.cap_key = gt._capture_key()
    e.g_grads = _Graph()
    exp = object()
    gt._fast[id(exp)] = (exp, e, None, 5, True)
    gt._eager_train = lambda experience, weights=None, **kw: "eager"
    state["hook"] = lambda g: g                           # installed after capture
    assert gt(exp) == "eager" and e.g_grads.replays == 0
