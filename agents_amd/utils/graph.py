"""HIP-graph capture of the trainer loop's three device programs -- the MI355X counterpart of
wrapping them in `common.function` (tf.function) as the reference's scripts do
(tf_agents/agents/dqn/examples/v2/train_eval.py:234-237) and as `Learner._train` is
(tf_agents/train/learner.py:309-337):

  GraphedTrain      agent.train            (forwards + loss + backward | optimizer)
  GraphedDriverRun  DynamicStepDriver.run  (one loop body = policy forward, action select, env
                                            step, replay add, step counter)
  GraphedSampler    replay get_next        (index sampling + row gather; a ring of output buffers)

Each of these is a chain of 10-45 short kernels; launched one by one from Python the loop is
host-bound (~14 us per launch against 5-50 us of GPU work each), replayed as graphs the host
issues about five graph launches per iteration.

Python side effects inside a captured region happen at capture time only -- exactly tf.function's
tracing semantics.  Components whose host mirrors must advance per execution (the replay buffer's
`last_id` mirror, the environment's current-time-step reference, a decaying epsilon) register them
with `on_replay`, which runs them before every replay.
"""
import contextlib
import gc
import os
import sys
import time
import weakref

import torch

from agents_amd.utils import nest_utils

_WARMUP_CALLS = 2
BUCKETED_ALLREDUCE = True   # False: one all-reduce of the whole gradient buffer after backward
_MAX_BINDINGS = 16   # captured train graphs per input signature (one per sampler ring slot)

# ---- capture context: host bookkeeping that must run once per replay ------------------------
_CAPTURE = None


class _CaptureCtx:
    def __init__(self):
        self.hooks = []


def capturing():
    """True while one of this module's captures is recording."""
    return _CAPTURE is not None


def on_replay(fn):
    """Runs `fn()` now -- or, while a HIP graph is being captured, registers it to run before
    every replay of that graph (capture itself executes nothing, so it is not run now)."""
    if _CAPTURE is None:
        fn()
        return
    obj, func = getattr(fn, "__self__", None), getattr(fn, "__func__", None)
    if obj is not None and func is not None:
        # A bound method is kept as (weak object, function): the hook advances a HOST mirror of
        # `obj`; a graph recorded for an agent must not keep that agent alive through
        # `agent._bump_counter` (agent -> graphs -> hook -> agent is a cycle only the collector
        # frees).  What a graph touches on the DEVICE is kept alive by the graph's owner.
        try:
            ref = weakref.ref(obj)
        except TypeError:
            _CAPTURE.hooks.append(fn)
            return

        def hook(ref=ref, func=func):
            o = ref()
            if o is not None:
                func(o)
        _CAPTURE.hooks.append(hook)
    else:
        _CAPTURE.hooks.append(fn)


# A/B knob AA_COUNT_IN_ADD=0: the driver's step counter stays a launch of its own in front of the
# replay buffer's add_batch (bit-identical either way)
COUNT_IN_ADD = True

# A/B knob: replay the optimizer phase as its own HIP graph (0) or launch it directly (1)
APPLY_EAGER = True
# False = part (b) of a two-part whole-mode train step (SAC: the actor's optimizer launch, the
# alpha update, the target update) is replayed as its HIP graph instead of being issued directly
# (3.2 % slower: profiles/r06_zzzz_sac_part_split_ab.txt; tests flip it)
WHOLE_B_EAGER = True


# AA_EARLY_TARGET (0 = off, default on): with overlap on, GraphedTrain runs the target network's
# forward of the NEXT train step right behind the gradient graph of this one, on the agent's side
# stream, next to the optimizer launch (see GraphedTrain._issue_early_target): 0.3458 -> 0.3319 ms
# per DQN iteration (round 4).  Variants that were measured and removed ("Round 4, second
# session" and "Round 5"; git history has the code): a stream of its own (same), the sample lane's
# stream (0.3771: it shares a hardware queue with the collect lane), stream order on the caller's
# stream (0.3545 vs 0.2959), a high-priority stream (0.973), the gradient phase as two graphs
# around the forward's event (0.3277 vs 0.3232) or around the first layer's weight gradient
# (0.3214 vs 0.2974), the forward launched during the previous step through a second activation
# slot (0.3209 vs 0.3170).
EARLY_TARGET = "0" if os.environ.get("AA_EARLY_TARGET", "side") in ("0", "off", "false") \
    else "side"


class Lanes:
    """Opt-in overlap of the three graphs on separate HIP streams (`enable_overlap(device)`).

    The loop `collect; sample; train` has fewer dependencies than its program order: the policy
    forward of collect(k) and the forward/backward of train(k) both only READ theta_k, and with a
    prefetching dataset the batch train(k) consumes was drawn iterations ago.  With overlap on,
    the collect graph replays on stream C, the sampler on stream S and training stays on the
    caller's stream M, ordered by events exactly along the true dependencies:
        C waits  M's frontier (theta_k written by apply(k-1); any eager work) and the last draw
                 (the replay tables it overwrites are not being gathered)
        S waits  the last collect (a draw sees every add that precedes it in program order) and
                 M's frontier (the ring slot it overwrites has been consumed)
        M waits  the draw that produced its batch before the forward, and the last collect before
                 the optimizer phase overwrites theta_k
    so every kernel sees the same inputs as in the single-stream order and results are
    bit-identical (tests/test_gpu_graphs.py).  Tensors returned by the graphed driver / dataset
    are produced on C / S: call `join_lanes()` before touching them from other code (the
    package's own eager entry points do)."""

    def __init__(self, device):
        self.device = device
        self.C = torch.cuda.Stream(device)
        self.S = torch.cuda.Stream(device)
        self.collect_done = None
        self.sample_done = None
        self.ready = {}          # first-leaf data_ptr of a sampler ring slot -> ready Event
        self.ready_seq = {}      # ... -> number of the draw that last filled the slot
        self.n_draws = 0
        self.aux_done = None     # last work issued on a lane besides collect / sample (early target)
        # Events are recycled round robin (creating and destroying four per iteration showed in
        # the host profile of the loop).  A holder of a recycled event waits for its NEWER record,
        # i.e. for more than it asked for -- never for less, and never for work enqueued after
        # the waiter -- 128 records (about 20 iterations) after it was handed out.
        self._pool = [torch.cuda.Event() for _ in range(128)]
        self._pool_i = 0

    def event_on(self, stream):
        ev = self._pool[self._pool_i]
        self._pool_i = (self._pool_i + 1) & 127
        ev.record(stream)
        return ev

    @staticmethod
    def _event_on(stream):
        ev = torch.cuda.Event()
        ev.record(stream)
        return ev

    def main_frontier(self):
        return self.event_on(current_stream(self.device))

    def join(self):
        """The caller's stream waits for everything enqueued on the collect / sample lanes."""
        cur = current_stream(self.device)
        if self.collect_done is not None:
            cur.wait_event(self.collect_done)
        if self.sample_done is not None:
            cur.wait_event(self.sample_done)
        if self.aux_done is not None:
            cur.wait_event(self.aux_done)


class _on_stream:
    """`with torch.cuda.stream(s):` for a caller that knows the stream it is on and stays on one
    device: two set_stream calls instead of the context manager's device and current-stream
    queries (the loop enters a lane three times per iteration; ~6 us of host time each)."""
    __slots__ = ("s", "prev")

    def __init__(self, s, prev):
        self.s, self.prev = s, prev

    def __enter__(self):
        torch.cuda.set_stream(self.s)
        return self.s

    def __exit__(self, *exc):
        torch.cuda.set_stream(self.prev)
        return False


_LANES = {}
# Lanes objects of devices whose overlap was switched off: switched on again, a device gets ITS
# streams back.  New streams land on other hardware queues (HIP deals streams to its 4 queues
# round robin), and a sample lane that shares a queue with the training stream serialises the loop
# (measured: 0.407 ms per iteration before an off / on cycle that created new streams, 0.492 after).
_LANES_PARKED = {}


def enable_overlap(device=None):
    """Turns on stream overlap of the graphed collect / sample / train programs on `device`."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None \
        else torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _LANES:
        lanes = _LANES_PARKED.pop(key, None)
        if lanes is None:
            lanes = Lanes(torch.device("cuda", key[1]))
        else:
            lanes.collect_done = lanes.sample_done = lanes.aux_done = None
            lanes.ready = {}
            lanes.ready_seq = {}
        _LANES[key] = lanes
    return _LANES[key]


def disable_overlap(device=None):
    join_lanes(device)
    if device is None:
        _LANES_PARKED.update(_LANES)
        _LANES.clear()
    else:
        device = torch.device(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        lanes = _LANES.pop((device.type, idx), None)
        if lanes is not None:
            _LANES_PARKED[(device.type, idx)] = lanes


def lanes_for(device):
    if not _LANES:
        return None
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return _LANES.get((device.type, idx))


def join_lanes(device=None):
    """Orders the current stream after all work on the overlap lanes (no-op without overlap).
    Eager entry points of the package call this; call it before reading, from your own code,
    tensors that a graphed driver or dataset returned."""
    if not _LANES or capturing():
        return
    for l in list(_LANES.values()) if device is None else [lanes_for(device)]:
        if l is not None:
            l.join()


# Other threads of the process (the RCCL watchdog of torch.distributed polls events) must not
# invalidate a capture on this thread: thread-local capture mode instead of torch's default global.
_CAPTURE_MODE = "thread_local"


_CAPTURES = 0          # captures recorded by this module so far (all objects)
_BATCH_DEPTH = 0
_CAPTURE_STREAM = {}


def _finalizing():
    return sys.is_finalizing()


def capture_count():
    """Number of HIP-graph captures this module has recorded.  A caller that wants its timed
    region free of captures runs its loop until this stops changing (bench.py's priming)."""
    return _CAPTURES


@contextlib.contextmanager
def capture_batch():
    """Brackets one or MANY captures: device idle + cyclic garbage collected ONCE before, the
    collector paused during (a finaliser that runs mid-capture -- a torch CUDAGraph or
    pinned-memory owner of some earlier, now unreachable stack being destroyed -- issues HIP
    calls that are illegal while a stream is capturing and corrupt the graph being recorded),
    device idle once after.  Nested uses are free, which is what lets the sampler ring, both
    driver bodies and every per-slot train graph be captured up front at the price of one."""
    global _BATCH_DEPTH
    outer = _BATCH_DEPTH == 0
    was = False
    if outer:
        torch.cuda.synchronize()
        # (kept: objects of the CALLER's cycles that own device memory or graphs -- this module's
        # own objects no longer need the collector to be released)
        gc.collect()
        release_dead(synchronize=False)
        was = gc.isenabled()
        gc.disable()
    _BATCH_DEPTH += 1
    try:
        yield
    finally:
        _BATCH_DEPTH -= 1
        if outer:
            if was:
                gc.enable()
            torch.cuda.synchronize()


def _ensure_prepared():
    """Networks with prepared weights (networks/sequential.py) whose parameters were written by a
    torch op since their last pre-pass get it re-run before a graph that relies on it replays."""
    from agents_amd.networks import sequential
    if sequential._PREPARED_NETS:
        sequential.ensure_prepared()


def _device_ctx(dev):
    """torch.cuda.device(dev), or nothing when dev is already the current device (the context
    manager costs two device queries and two switches per use)."""
    if torch.cuda.current_device() == dev.index:
        return contextlib.nullcontext()
    return torch.cuda.device(dev)


def _capture_stream(device):
    key = torch.cuda.current_device() if device is None else torch.device(device).index
    st = _CAPTURE_STREAM.get(key)
    if st is None:
        st = _CAPTURE_STREAM[key] = torch.cuda.Stream(key)
    return st


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None) \
    if os.environ.get("AA_RAW_STREAM", "1") != "0" else None     # A/B knob: see _lib.stream_ptr
_STREAM_OBJECTS = {}    # (device index, raw hipStream_t) -> torch.cuda.Stream


def current_stream(device):
    """`torch.cuda.current_stream(device)` for the loop's hot paths: the raw handle comes from
    the C entry point, the Stream object wrapping it from a cache (the torch call resolves the
    device index in Python and builds a new object every time: ~6 us, five times per iteration
    of the DQN loop)."""
    idx = device.index
    if _RAW_STREAM is None or idx is None:
        return torch.cuda.current_stream(device)
    key = (idx, _RAW_STREAM(idx))
    s = _STREAM_OBJECTS.get(key)
    if s is None:
        s = _STREAM_OBJECTS[key] = torch.cuda.current_stream(device)
    return s


def _bind_launch():
    from agents_amd import _lib
    return _lib.load().aa_hip_graph_launch, _lib.stream_ptr


class _Lazy:
    """Resolves the library's launch entry on first use (importing this module must not load it)."""

    def __call__(self, *a):
        global _GRAPH_LAUNCH, _STREAM_PTR
        _GRAPH_LAUNCH, _STREAM_PTR = _bind_launch()
        return _GRAPH_LAUNCH(*a)


_GRAPH_LAUNCH = _Lazy()
_STREAM_PTR = lambda: _bind_launch()[1]()       # noqa: E731  (replaced with the first launch)
REPLAY_TIMERS = None    # set to {} to accumulate {kind: [launches, host seconds]}
TIMELINE = None         # set to [] to collect (tag, timing event) marks around the graph launches


def _mark(tag, stream=None):
    if TIMELINE is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream if stream is not None else torch.cuda.current_stream())
        TIMELINE.append((tag, ev))


# ---- lifetime of the recorded graphs ------------------------------------------------------------
# A `_Captured` is owned by exactly one Graphed* object, which is owned by its agent / driver /
# dataset iterator; the back-pointers (GraphedTrain -> agent, GraphedDriverRun -> driver) are weak,
# so dropping the owner releases its graphs by reference count -- no cyclic collection involved.
# The hipGraphExec itself is NOT destroyed at that moment: `close()` parks the torch CUDAGraph in
# `_GRAVEYARD` (a finaliser may run anywhere -- in the middle of somebody else's stream capture,
# where a HIP call is illegal, or while the last replay is still executing on the device) and the
# graveyard is emptied at the next point where this module knows the device to be idle: the start
# of a capture batch, `release_dead()`, or -- so that a process that only drops agents cannot pile
# them up -- as soon as `_GRAVEYARD_MAX` graphs wait, behind a device synchronisation of its own.
_GRAVEYARD = []
_GRAVEYARD_MAX = 32
_LIVE_GRAPHS = 0        # recorded and not yet closed (tests/test_gpu_lifetime.py watches it)


def live_graphs():
    """(recorded graphs not yet closed, closed graphs waiting for an idle device)."""
    return _LIVE_GRAPHS, len(_GRAVEYARD)


def release_dead(synchronize=True):
    """Destroys the hipGraphExecs of closed graphs.  With `synchronize` the device is made idle
    first (what a caller outside this module wants); capture_batch() passes False right after its
    own synchronisation.  A no-op while a capture is being recorded."""
    if not _GRAVEYARD or _CAPTURE is not None:
        return 0
    if synchronize:
        torch.cuda.synchronize()
    dead = _GRAVEYARD[:]
    del _GRAVEYARD[:len(dead)]
    n = len(dead)
    del dead            # _Exec.__del__ -> hipGraphExecDestroy; CUDAGraph.__del__ releases the pool
    # ... to the caching allocator's list of freeable pools, which only an out-of-memory retry or
    # empty_cache() returns to the device: without this a process that builds and drops agents
    # grows by a 20 MiB segment per recorded graph (tools/lifetime_probe.py)
    torch.cuda.empty_cache()
    return n


# ---- a HIP-runtime fault in hipGraphLaunch, and the guard against it -----------------------------
# (csrc/runtime_guard.hip has the analysis.)  A recorded graph with side branches gets n internal
# "parallel" streams when it is instantiated; each lands on the hardware queue that has the fewest
# streams at that moment, and hipGraphLaunch reads past their list -- SIGSEGV -- when two of them
# share the LAUNCH stream's queue.  That needs unevenly loaded queues, i.e. a process that has
# destroyed other graph execs: the GPU suite's 671st test in rounds 5 and 6, never the benchmark.
# The guard: keep the captured hipGraph (torch CUDAGraph(keep_graph=True), which also owns the
# capture's memory pool) and let the library instantiate it: where two of the fresh exec's parallel
# streams share a queue the exec is destroyed, one ballast stream evens the queue loads out, and
# the graph is instantiated again (csrc/runtime_guard.hip: aa_hip_graph_instantiate).  Streams on
# pairwise different queues are safe for EVERY launch stream.  Replays launch that exec directly
# (aa_hip_graph_launch = hipGraphLaunch on the current stream: what torch's replay() does).
# Capture time only (a few instantiations of ~100 us each); on any other runtime version the
# library instantiates once and checks nothing.
EXEC_GUARD = True           # tests switch it off to show the hazard without meeting it
exec_guard_stats = {"execs": 0, "respread": 0, "instantiations": 0, "unsupported": 0}


class _Exec:
    """A hipGraphExec of our own (see above), destroyed with the object."""
    __slots__ = ("ptr", "streams", "worst")

    def __init__(self, g):
        import ctypes
        from agents_amd import _lib
        ex, n, worst, tries = ctypes.c_void_p(), ctypes.c_int32(0), ctypes.c_int32(0), \
            ctypes.c_int32(0)
        self.ptr = None
        rc = _lib.load().aa_hip_graph_instantiate(
            g.raw_cuda_graph(), 1 if EXEC_GUARD else 0, ctypes.byref(ex), ctypes.byref(n),
            ctypes.byref(worst), ctypes.byref(tries))
        if rc == _lib.AA_ERR_RANGE:
            raise RuntimeError(
                "could not spread a HIP graph's parallel streams over different hardware queues "
                "(hipGraphLaunch of this runtime would read out of bounds): set "
                "DEBUG_HIP_FORCE_GRAPH_QUEUES=1 to record linear graphs")
        _lib.check(rc, "aa_hip_graph_instantiate")
        self.ptr, self.streams, self.worst = ex.value, n.value, worst.value
        exec_guard_stats["execs"] += 1
        exec_guard_stats["instantiations"] += tries.value
        exec_guard_stats["respread"] += int(tries.value > 1)
        exec_guard_stats["unsupported"] += int(n.value < 0)

    def __del__(self):
        ptr, self.ptr = self.ptr, None
        if ptr is not None:
            try:
                from agents_amd import _lib
                _lib.load().aa_hip_graph_exec_destroy(ptr)
            except Exception:       # interpreter shutdown
                pass


class _Captured:
    """A torch CUDAGraph plus the host hooks registered while it was captured."""

    def __init__(self, kind="graph"):
        self.graph = None         # torch CUDAGraph: the captured hipGraph and its memory pool
        self.exec = None          # _Exec: the instantiated graph that replays launch
        self.hooks = []
        self.out = None
        self.kind = kind

    @property
    def spread(self):
        """(streams of the exec, most parallel streams on one hardware queue), or None when the
        runtime is not the one the guard knows."""
        e = self.exec
        return None if e is None or e.streams < 0 else (e.streams, e.worst)

    def close(self):
        """Gives up the graph (see `_GRAVEYARD`) and everything its hooks keep alive."""
        global _LIVE_GRAPHS
        g, self.graph = self.graph, None
        ex, self.exec = self.exec, None
        self.hooks = []
        self.out = None
        if g is not None:
            _LIVE_GRAPHS -= 1
            _GRAVEYARD.append((ex, g))       # (released in this order: the exec, then its graph)
            if len(_GRAVEYARD) >= _GRAVEYARD_MAX and _BATCH_DEPTH == 0 and _CAPTURE is None \
                    and not _finalizing():
                release_dead()

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown
            pass

    def capture(self, fn):
        """Records `fn()`.  torch.cuda.graph() is not used: its __enter__ synchronises the device
        and empties the allocator cache for EVERY capture; `capture_batch` does the former once
        per batch and the latter is not wanted (it returns the cached blocks the eager warm-up
        calls just sized to the driver)."""
        global _CAPTURE, _CAPTURES
        if _CAPTURE is not None:
            raise RuntimeError("nested HIP-graph capture")
        ctx = _CaptureCtx()
        g = torch.cuda.CUDAGraph(keep_graph=True)     # the hipGraph stays: see _Exec
        with capture_batch():
            _CAPTURE = ctx
            try:
                with torch.cuda.stream(_capture_stream(None)):
                    g.capture_begin(capture_error_mode=_CAPTURE_MODE)
                    try:
                        self.out = fn()
                    finally:
                        g.capture_end()
            finally:
                _CAPTURE = None
            self.exec = _Exec(g)
        _CAPTURES += 1
        global _LIVE_GRAPHS
        _LIVE_GRAPHS += 1
        self.graph = g
        self.hooks = ctx.hooks
        return self.out

    def replay(self):
        for h in self.hooks:
            h()
        if REPLAY_TIMERS is None:
            if _GRAPH_LAUNCH(self.exec.ptr, _STREAM_PTR()):
                raise RuntimeError("hipGraphLaunch failed")
        else:     # host cost of the launch per kind of graph (bench.py --host-profile)
            t0 = time.perf_counter()
            if _GRAPH_LAUNCH(self.exec.ptr, _STREAM_PTR()):
                raise RuntimeError("hipGraphLaunch failed")
            acc = REPLAY_TIMERS.setdefault(self.kind, [0, 0.0])
            acc[0] += 1
            acc[1] += time.perf_counter() - t0
        return self.out


class _Entry:
    def __init__(self):
        self.static_in = None
        self.static_w = None
        self.g_grads = None       # _Captured: forwards + loss + backward (bucket mode: first half)
        self.g_apply = None       # _Captured: optimizer (shared by every entry of a signature)
        self.apply_state = None   # agent._apply_state() after this entry's gradient phase
        self.g_grads_b = None     # _Captured, bucket mode: second half of the backward
        self.captured = None      # _Captured, whole mode (part (a) when the agent splits it)
        self.captured_b = None    # _Captured, whole mode, part (b)
        self.part_a_state = None  # the agent's hand-over from part (a) to part (b), this entry's
        self.out = None
        # early target forward (ring-slot entries of agents with `_train_phase_target`)
        self.ptr0 = None          # first-leaf address of the batch this entry reads in place
        self.g_target = None      # _Captured: the target network's forward on this entry's batch
        self.g_grads_nt = None    # _Captured: g_grads without the target forward (reads g_target's)
        self.apply_state_nt = None
        # what the agent's phases looked at while they were recorded and a replay cannot re-evaluate
        # (DqnAgent: whether a gradient hook is installed decides if the backward pass leaves its
        # conv weight gradients as slabs for the optimizer); a replay under another state would
        # all-reduce stale gradients -- such a call takes the eager path instead
        self.cap_key = None

    def close(self):
        for name in ("g_grads", "g_apply", "g_grads_b", "captured", "captured_b", "g_target",
                     "g_grads_nt"):
            c = getattr(self, name)
            if c is not None:
                c.close()
                setattr(self, name, None)
        self.static_in = self.static_w = self.out = None
        self.apply_state = self.apply_state_nt = None


def _sig(experience, weights):
    leaves = nest_utils.flatten(experience)
    s = tuple((tuple(t.shape), t.dtype, t.device) for t in leaves)
    if isinstance(weights, torch.Tensor):
        return s + ((tuple(weights.shape), weights.dtype),)
    return s + (weights,)


# first-leaf address of a sampler ring slot -> weakref of the GraphedSampler that owns it: lets
# GraphedTrain capture one graph per slot of the WHOLE ring the first time it meets one of them
_RING_OWNER = {}


def _ring_of(ptr0):
    ref = _RING_OWNER.get(ptr0)
    owner = ref() if ref is not None else None
    if owner is None:
        _RING_OWNER.pop(ptr0, None)
    return owner


class GraphedTrain:
    """Callable with the signature of `agent.train`; falls back to the eager path for agents
    that do not expose graphable phases.

    Graphs are bound to the ADDRESSES of their inputs.  A batch that lives in a GraphedSampler
    ring slot gets a graph captured on that slot -- no copies -- and the first such batch
    triggers the capture for EVERY slot of its ring (one pause of a few tens of ms in the third
    call, none later).  Any other caller (fresh tensors every step) shares one graph captured on
    private clones and pays one copy per leaf; an unknown address set that comes back a second
    time (somebody else's static buffers, e.g. PPOLearner's minibatch) gets its own graph."""

    def __init__(self, agent):
        self._agent = agent    # (a weak reference: see the property)
        self._cache = {}       # signature -> {input address tuple | None: _Entry}
        self._warm = {}
        self._seen = {}
        self._fast = {}        # id(experience) -> (experience, entry, device, first address, True)
        # phase mode: two graphs around the gradient hook (DqnAgent); whole mode: the entire
        # `_train` in one graph, for agents whose train step has no host-side decisions and no
        # gradient hook installed (PPOAgent on one replica)
        self._phases = all(hasattr(agent, n) for n in
                           ("_train_phase_grads", "_train_phase_apply", "_train_phase_host"))
        self._whole = not self._phases and hasattr(agent, "_graph_train_whole")
        self.enabled = self._phases or self._whole
        self.replays = 0
        # early target forward (phase mode, overlap on, batches in sampler ring slots)
        self._succ = {}          # entry -> entry of the call that followed it (the ring is cyclic)
        self._prev_entry = None
        self._early = None       # (entry, done event, agent._early_target_key(), draw number, device)
        self._early_sized = False
        self.early_hits = 0
        self.early_issued = 0

    # The agent owns this object (`agent._graphed_train`, and `agent.train` itself where a script
    # rebinds it: `tf_agent.train = common.function(tf_agent.train)`); the pointer back is weak, so
    # agent and graphs are released by reference count the moment the agent is dropped.
    @property
    def _agent(self):
        a = self._agent_ref()
        if a is None:
            raise ReferenceError(
                "the agent of this graphed train function has been released; keep a reference to "
                "the agent for as long as the function returned by common.function(agent.train) "
                "is used")
        return a

    @_agent.setter
    def _agent(self, agent):
        try:
            self._agent_ref = weakref.ref(agent)
        except TypeError:            # not weakly referencable (a stand-in in a test)
            self._agent_ref = lambda: agent

    @property
    def agent(self):
        return self._agent

    def _eager_train(self, experience, weights=None, **kwargs):
        # the class's own method: `tf_agent.train = common.function(tf_agent.train)` (the PPO and
        # SAC scripts) rebinds the INSTANCE attribute to this object
        agent = self._agent
        return type(agent).train(agent, experience, weights=weights, **kwargs)

    def close(self):
        """Releases every recorded graph, their static inputs and the scratch of the early target
        forward.  Called when the agent goes away (weakref.finalize in `graphed_train`) and by
        `__del__`; the object falls back to recording again if it is called afterwards."""
        for bound in self._cache.values():
            for e in bound.values():
                e.close()
        for e in list(self._succ) + list(self._succ.values()):
            e.close()
        self._cache.clear()
        self._fast.clear()
        self._succ.clear()
        self._seen.clear()
        self._prev_entry = None
        self._early = None
        if self._early_sized:
            from agents_amd import ops
            ops.release_scope(("early_target", id(self)))
            self._early_sized = False

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown
            pass

    def _entry_for(self, sig, ptrs, experience, weights, dev):
        bound = self._cache.setdefault(sig, {})
        e = bound.get(ptrs)
        if e is not None:
            return e
        shared_apply = next((x.g_apply for x in bound.values() if x.g_apply is not None), None)
        sampler = _ring_of(ptrs[0])
        if sampler is not None and len(bound) < _MAX_BINDINGS:
            # a sampler ring slot: one graph per slot of that ring, all captured now
            join_lanes(dev)
            in_order = []
            with capture_batch():
                for exp_k in sampler.ring_experiences():
                    pk = tuple(t.data_ptr() for t in nest_utils.flatten(exp_k))
                    if pk in bound:
                        in_order.append(bound[pk])
                    if pk in bound or _sig(exp_k, weights) != sig or \
                            len(bound) >= _MAX_BINDINGS:
                        continue
                    ek = _Entry()
                    ek.ptr0 = pk[0]
                    self._capture(ek, exp_k, weights, clone=False, g_apply=shared_apply,
                                  ring=True)
                    shared_apply = ek.g_apply
                    bound[pk] = ek
                    in_order.append(ek)
            if len(in_order) == len(sampler.ring_experiences()):
                # the dataset hands the slots out in ring order: each entry's successor is known
                # from the start (a call that breaks the order re-learns it, _issue_early_target)
                for a, b in zip(in_order, in_order[1:] + in_order[:1]):
                    self._succ.setdefault(a, b)
            e = bound.get(ptrs)
            if e is not None:
                return e
        seen = self._seen.setdefault(sig, {})
        seen[ptrs] = seen.get(ptrs, 0) + 1
        if len(seen) > 4 * _MAX_BINDINGS:
            seen.clear()
        if seen.get(ptrs, 0) >= 2 and len(bound) < _MAX_BINDINGS:
            # an address set that came back: somebody's static buffers, worth their own graph
            join_lanes(dev)
            e = _Entry()
            self._capture(e, experience, weights, clone=False, g_apply=shared_apply)
            bound[ptrs] = e
            return e
        e = bound.get(None)
        if e is None:
            # captured on private clones; serves every other caller through copies
            join_lanes(dev)
            e = _Entry()
            self._capture(e, experience, weights, clone=True, g_apply=shared_apply)
            bound[None] = e
        return e

    def _capture_key(self):
        fn = getattr(self._agent, "_graph_capture_key", None)
        return fn() if fn is not None else None

    def _consume_early(self):
        """Orders the caller's stream behind a pending early target forward and forgets it: every
        path out of __call__ that does not use the result (eager fall-backs included) goes through
        here, so that no launch of this call can overlap the forward's writes of the target
        network's activation slot -- whatever the agent's eager path joins or does not join."""
        early, self._early = self._early, None
        if early is not None and not capturing():
            # on the device the forward was issued on (an eager fall-back runs on the agent's
            # device stream, which need not be the process's current device)
            dev = early[4] if len(early) > 4 else None
            (torch.cuda.current_stream(dev) if dev is not None
             else torch.cuda.current_stream()).wait_event(early[1])
        return early

    def __call__(self, experience, weights=None, **kwargs):
        agent = self._agent
        if not self.enabled or kwargs or getattr(agent, "check_numerics", False) or capturing() \
                or not getattr(agent, "graph_train_ok", True):
            # (graph_train_ok False: a DqnAgent with a td_errors_loss_fn of the caller's own, which
            # torch evaluates -- with autograd -- in the middle of the step)
            self._consume_early()
            return self._eager_train(experience, weights=weights, **kwargs)
        if self._whole and (getattr(agent, "gradient_hook", None) is not None or
                            not getattr(agent, "graph_train_whole_ok", True)):
            self._consume_early()
            return self._eager_train(experience, weights=weights)
        _ensure_prepared()
        # Steady state: the very same experience object (a sampler ring slot) as on an earlier
        # call whose graph reads it in place -- signature, trajectory checks and address tuple
        # were established then (this lookup replaces ~40 us of host work per step).
        hit = self._fast.get(id(experience)) if weights is None else None
        if hit is not None and hit[0] is experience:
            _, e, dev, ptr0, in_place = hit
        else:
            sig = _sig(experience, weights)
            if self._warm.get(sig, 0) < _WARMUP_CALLS:
                self._warm[sig] = self._warm.get(sig, 0) + 1
                self._consume_early()
                return self._eager_train(experience, weights=weights)
            if not agent._initialized:
                agent.initialize()
            if hasattr(agent, "_check_trajectory"):
                agent._check_trajectory(experience)
            ptrs = tuple(t.data_ptr() for t in nest_utils.flatten(experience))
            dev = experience.discount.device
            if dev.index is None:
                dev = torch.device("cuda", torch.cuda.current_device())
            with torch.cuda.device(dev):
                e = self._entry_for(sig, ptrs, experience, weights, dev)
            ptr0 = ptrs[0]
            in_place = all(d.data_ptr() == p for d, p in
                           zip(nest_utils.flatten(e.static_in), ptrs))
            if weights is None and in_place and e.static_w is None:
                if len(self._fast) > 256:
                    self._fast.clear()
                self._fast[id(experience)] = (experience, e, dev, ptr0, True)   # keeps it alive
        if e.cap_key is not None and e.cap_key != self._capture_key():
            # recorded under another hook / clipping state (a gradient hook installed after
            # `common.function(agent.train)` had captured): the graphs' backward and optimizer
            # phases no longer describe the step
            self._consume_early()
            return self._eager_train(experience, weights=weights)
        with _device_ctx(dev):
            lanes = _LANES.get((dev.type, dev.index)) if _LANES else None
            cur = current_stream(dev)     # (looked up once: ~2 us of host time each)
            use_early = False
            early, self._early = self._early, None
            if early is not None:
                # whatever this call launches comes after the early forward's writes (the target
                # network's activation slot, its output); its result is used only if it was made
                # for THIS entry, from the draw the slot still holds, and nothing it depends on
                # has been written since
                cur.wait_event(early[1])
                use_early = (early[0] is e and lanes is not None and e.g_grads_nt is not None
                             and early[2] == agent._early_target_key()
                             and early[3] == lanes.ready_seq.get(ptr0))
                self.early_hits += int(use_early)
            if lanes is not None:
                ev = lanes.ready.get(ptr0)
                if ev is not None:       # the draw that filled this ring slot (on lane S)
                    cur.wait_event(ev)
                else:
                    lanes.join()
            if not in_place:
                # only after the wait above: the source may still be being written on lane S
                for dst, src in zip(nest_utils.flatten(e.static_in),
                                    nest_utils.flatten(experience)):
                    if dst.data_ptr() != src.data_ptr():
                        dst.copy_(src, non_blocking=True)
            if e.static_w is not None and e.static_w.data_ptr() != weights.data_ptr():
                e.static_w.copy_(weights, non_blocking=True)
            if self._whole and e.captured_b is not None:
                # part (a) only reads what the collect step reads (the policy's weights) and the
                # batch (waited for above); part (b) overwrites the policy: after the collect
                # step that is using it.  NOT after the draw issued in this iteration: that batch is
                # consumed `prefetch` iterations from now behind its own `ready` event, and the
                # draw reads nothing part (b) writes.  (Round 3 joined both lanes here: the GPU
                # timeline of the SAC loop -- tools/bench_sac.py --timeline -- showed part (b)
                # starting 28 us after part (a) had finished, waiting for a 9 us gather.)
                _mark("train.begin")
                e.captured.replay()
                _mark("train.part_a_done")
                if lanes is not None and lanes.collect_done is not None:
                    cur.wait_event(lanes.collect_done)
                _mark("train.part_b_begin")
                if WHOLE_B_EAGER and getattr(agent, "graph_train_whole_b_eager_ok", False):
                    # part (b) is a handful of launches: issued directly they follow part (a)
                    # without a second graph-launch boundary on the critical stream
                    agent._part_a = e.part_a_state
                    out = agent._graph_train_whole_b()
                    _mark("train.apply_done")
                    self.replays += 1
                    return out
                e.captured_b.replay()
                _mark("train.apply_done")
            elif self._whole:
                if lanes is not None:
                    lanes.join()
                e.captured.replay()
            elif e.g_grads_b is not None:
                # bucket mode (data-parallel): [forward + loss + dense-tail backward] -> start the
                # all-reduce of the tail's gradients -> [conv backward] overlaps it -> all-reduce
                # of the (small) head -> wait for both -> optimizer
                tail, head = agent._q_network.grad_buckets(agent._bucket_split())
                (e.g_grads_nt if use_early else e.g_grads).replay()
                w1 = agent.gradient_hook_async(tail)
                e.g_grads_b.replay()
                grads_done = self._early_mark(lanes, cur)
                w2 = agent.gradient_hook_async(head)
                w1.wait()
                w2.wait()
                if lanes is not None and lanes.collect_done is not None:
                    cur.wait_event(lanes.collect_done)
                if hasattr(agent, "_set_apply_state"):
                    agent._set_apply_state(e.apply_state_nt if use_early else e.apply_state)
                e.g_apply.replay()
                tw = getattr(agent, "_target_writes", None)
                agent._train_phase_host()
                self._issue_early_target(e, lanes, dev, cur, grads_done,
                                         getattr(agent, "_target_writes", None) != tw)
            else:
                _mark("train.begin")
                (e.g_grads_nt if use_early else e.g_grads).replay()
                _mark("train.grads_done")
                grads_done = self._early_mark(lanes, cur)
                if agent.gradient_hook is not None:
                    agent.gradient_hook(agent._q_network.flat_grads)
                if lanes is not None and lanes.collect_done is not None:
                    # the optimizer overwrites theta_k: the collect policy's forward must be done
                    cur.wait_event(lanes.collect_done)
                if hasattr(agent, "_set_apply_state"):
                    agent._set_apply_state(e.apply_state_nt if use_early else e.apply_state)
                if APPLY_EAGER:
                    # the optimizer phase is one or two launches: issued directly they follow the
                    # gradient graph without a second graph-launch boundary on the critical stream
                    agent._train_phase_apply()
                else:
                    e.g_apply.replay()
                _mark("train.apply_done")
                tw = getattr(agent, "_target_writes", None)
                agent._train_phase_host()
                self._issue_early_target(e, lanes, dev, cur, grads_done,
                                         getattr(agent, "_target_writes", None) != tw)
        self.replays += 1
        return e.out

    # ---- early target forward ---------------------------------------------------------------
    # The target network's forward of train(k+1) reads the batch of step k+1 (drawn `prefetch`
    # iterations ago) and theta_target -- nothing the optimizer step of train(k) writes.  Its own
    # graph per ring slot, it is launched on a lane right behind the gradient graph of train(k):
    # it runs next to the optimizer launch and through the boundary between the iterations, where
    # the device otherwise executes one memory-bound kernel, and train(k+1) replays its gradient
    # graph WITHOUT the target forward.  Which batch train(k+1) will get is predicted from the
    # order the entries came in last time (a sampler ring is cyclic); a wrong guess, a target
    # update or an eager launch in between only cost the early forward (the full graph replays).
    def _early_mark(self, lanes, cur):
        """Event on the caller's stream behind the gradient graph (its loss launch has consumed
        the target output of THIS step), or None when early target forwards are off."""
        if lanes is None or EARLY_TARGET == "0" or \
                not hasattr(self._agent, "_train_phase_target"):
            return None
        return lanes.event_on(cur)

    def _early_target_stream(self, lanes, dev, cur=None):
        """The agent's side stream (the weight-gradient branch's, idle behind the gradient graph);
        the sample lane for an agent that keeps everything on one stream."""
        if getattr(self._agent, "_side_stream", None) is not None:
            st = self._agent._side_stream(dev)
            if st is not None:
                return st
        return lanes.S

    def _issue_early_target(self, e, lanes, dev, cur, grads_done, target_written):
        prev, self._prev_entry = self._prev_entry, e
        if prev is not None:
            self._succ[prev] = e
        if lanes is None or grads_done is None or EARLY_TARGET == "0":
            return
        nxt = self._succ.get(e)
        if nxt is None or nxt.g_target is None or nxt.ptr0 is None:
            return
        seq = lanes.ready_seq.get(nxt.ptr0)
        rdy = lanes.ready.get(nxt.ptr0)
        if seq is None or rdy is None:
            return
        st = self._early_target_stream(lanes, dev, cur)
        if st is not cur:
            st.wait_event(grads_done)
            if target_written:   # theta_target was updated behind the optimizer step, on this stream
                st.wait_event(lanes.event_on(cur))
        st.wait_event(rdy)
        with _on_stream(st, cur):
            _mark("early_target.begin", st)
            nxt.g_target.replay()
            done = lanes.event_on(st)
            _mark("early_target.done", st)
        lanes.aux_done = done
        self._early = (nxt, done, self._agent._early_target_key(), seq, dev)
        self.early_issued += 1

    def _capture_early(self, e, bucketed):
        """Per ring slot: the target forward alone, and the gradient phase that reads its output."""
        from agents_amd import ops
        agent = self._agent
        dev = e.static_in.discount.device
        with ops.workspace_scope(("early_target", id(self)), dev):
            if not self._early_sized:
                # one un-captured pass sizes this scope's scratch (it must not grow in a capture)
                agent._train_phase_target(e.static_in)
                agent._target_fwd_epoch += 1
                self._early_sized = True
            gt = _Captured("train.target")
            q_t = gt.capture(lambda: agent._train_phase_target(e.static_in))
        phase = agent._train_phase_grads_a if bucketed else agent._train_phase_grads
        gn = _Captured("train.grads")
        gn.capture(lambda: phase(e.static_in, None, q_next_target=q_t))
        e.apply_state_nt = agent._apply_state() if hasattr(agent, "_apply_state") else None
        e.g_target, e.g_grads_nt = gt, gn

    def _capture(self, e, experience, weights, clone=True, g_apply=None, ring=False):
        """Records the entry's graphs.  Host bookkeeping the phases do through `on_replay` (the
        optimizer's `iterations` mirror) is collected as replay hooks, and `capturing()` is true
        throughout, so nothing in the phases waits on un-captured events (join_lanes is a no-op)."""
        agent = self._agent
        e.static_in = nest_utils.map_structure(lambda t: t.clone(), experience) if clone \
            else experience
        e.static_w = weights.clone() if isinstance(weights, torch.Tensor) else None
        w_arg = e.static_w if e.static_w is not None else weights
        e.cap_key = self._capture_key()
        with capture_batch():
            if self._whole:
                if hasattr(agent, "_graph_train_whole_a"):
                    # two graphs: (a) everything that leaves the collect policy's weights alone,
                    # (b) the rest -- with overlap on, (a) runs beside the collect step
                    e.captured = _Captured("train.whole_a")
                    e.captured.capture(lambda: agent._graph_train_whole_a(e.static_in, w_arg))
                    # what part (a) hands to part (b): THIS entry's static tensors (an eagerly
                    # issued part (b) must not see the ones of whichever entry was captured last)
                    e.part_a_state = getattr(agent, "_part_a", None)
                    e.captured_b = _Captured("train.whole_b")
                    e.out = e.captured_b.capture(agent._graph_train_whole_b)
                    return
                e.captured = _Captured("train.whole")
                e.out = e.captured.capture(lambda: agent._graph_train_whole(e.static_in, w_arg))
                return
            bucketed = (getattr(agent, "gradient_hook_async", None) is not None and
                        hasattr(agent, "_train_phase_grads_a") and
                        agent._bucket_split() is not None and BUCKETED_ALLREDUCE)
            e.g_grads = _Captured("train.grads")
            e.out = e.g_grads.capture(
                (lambda: agent._train_phase_grads_a(e.static_in, w_arg)) if bucketed else
                (lambda: agent._train_phase_grads(e.static_in, w_arg)))
            if bucketed:
                e.g_grads_b = _Captured("train.grads_b")
                e.g_grads_b.capture(agent._train_phase_grads_b)
            # what the optimizer phase takes over from THIS entry's gradient phase besides
            # flat_grads (DqnAgent: the unsummed conv weight-gradient slabs of its batch size);
            # replays put it back before the phase runs, eagerly or from its own graph
            e.apply_state = agent._apply_state() if hasattr(agent, "_apply_state") else None
            if g_apply is not None and e.apply_state is None and \
                    not getattr(g_apply, "has_state", False):
                e.g_apply = g_apply          # the optimizer phase does not depend on the inputs
            else:
                e.g_apply = _Captured("train.apply")
                e.g_apply.has_state = e.apply_state is not None
                e.g_apply.capture(agent._train_phase_apply)
            if ring and not clone and weights is None and EARLY_TARGET != "0" and \
                    hasattr(agent, "_train_phase_target") and \
                    hasattr(agent, "_early_target_key"):
                self._capture_early(e, bucketed)

    def static_inputs(self, experience_like=None):
        """Static input nest of the (single) captured signature, or None before capture."""
        for bound in self._cache.values():
            for e in bound.values():
                if e.static_in is not None:
                    return e.static_in
        return None


def graphed_train(agent):
    """One GraphedTrain per agent (shared by common.function and the Learner).  The agent owns it;
    it points back weakly."""
    g = getattr(agent, "_graphed_train", None)
    if g is None:
        g = GraphedTrain(agent)
        agent._graphed_train = g
    return g


class GraphedSampler:
    """`next()` == `rb.get_next(S, T)` (tf_uniform_replay_buffer.py:211-310), replayed as a HIP
    graph of the sampling + gather launch(es).  Outputs live in a ring of `ring` static buffer
    sets: an element stays valid until `ring - 1` further elements have been drawn (the
    reference's dataset hands out fresh tensors; consumers that keep samples longer than that
    should call `get_next` directly).  The Philox call counter is device-resident, so the sampled
    indices are bit-identical to the eager path's.  The whole ring is captured at the first
    graphed call (the third draw), so later draws never pause."""

    def __init__(self, rb, sample_batch_size, num_steps, ring=8):
        self._rb = rb
        self._S = sample_batch_size
        self._T = num_steps
        self._ring = [None] * max(int(ring), 2)
        self._i = 0
        self._warm = 0
        self._ref = None
        self._stamped = None      # per ring slot: packed arguments of rb.draw_into
        self.enabled = True
        self.replays = 0

    def ring_experiences(self):
        """The experience nest of every captured ring slot, in slot order."""
        return [c.out[0] for c in self._ring if c is not None]

    def _prime(self):
        import weakref
        rb = self._rb
        join_lanes(rb.device)
        try:
            with capture_batch():
                for k in range(len(self._ring)):
                    c = _Captured("sample")
                    c.capture(lambda: rb.get_next(self._S, self._T, time_stacked=True))
                    self._ring[k] = c
        except Exception:
            self.enabled = False
            raise
        self._ref = weakref.ref(self)
        for exp in self.ring_experiences():
            _RING_OWNER[nest_utils.flatten(exp)[0].data_ptr()] = self._ref

    def close(self):
        """Releases the ring's graphs and output buffers and what the module keeps per slot."""
        # only OUR registrations: the addresses may have been reused by a younger ring since
        ref = getattr(self, "_ref", None)
        for c in self._ring:
            if c is None:
                continue
            if c.out is not None:
                p0 = nest_utils.flatten(c.out[0])[0].data_ptr()
                if ref is not None and _RING_OWNER.get(p0) is ref:
                    del _RING_OWNER[p0]
                    for lanes in list(_LANES.values()) + list(_LANES_PARKED.values()):
                        lanes.ready.pop(p0, None)
                        lanes.ready_seq.pop(p0, None)
            c.close()
        self._ring = [None] * len(self._ring)
        self._stamped = None
        self._warm = 0

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown
            pass

    def next(self):
        rb = self._rb
        if not self.enabled or self._warm < _WARMUP_CALLS or capturing():
            self._warm += 1
            return rb.get_next(self._S, self._T, time_stacked=True)
        rb._check_not_empty(self._T)
        slot = self._i % len(self._ring)
        self._i += 1
        dev = rb.device
        with _device_ctx(dev):
            if self._ring[slot] is None:
                self._prime()
                self._ptr0 = [nest_utils.flatten(c.out[0])[0].data_ptr() for c in self._ring]
                if self._S is not None and self._T is not None and \
                        getattr(rb, "supports_stamped_draws", lambda: False)():
                    self._stamped = [rb.stamped_slot(c.out) for c in self._ring]
            c = self._ring[slot]
            # a buffer that mirrors its counters on the host draws with one eager launch that
            # carries them by value (no dependent counter read on the device); otherwise the
            # captured device-counter launch is replayed
            stamped = self._stamped[slot] if self._stamped is not None else None
            lanes = _LANES.get((dev.type, dev.index)) if _LANES else None
            if lanes is None:
                if stamped is not None:
                    rb.draw_into(stamped)
                    out = c.out
                else:
                    out = c.replay()
            else:
                if lanes.collect_done is not None:
                    lanes.S.wait_event(lanes.collect_done)
                cur = current_stream(dev)
                lanes.S.wait_event(lanes.event_on(cur))      # = the main frontier
                if lanes.aux_done is not None:
                    # an early target forward may still be reading the slot this draw overwrites
                    lanes.S.wait_event(lanes.aux_done)
                with _on_stream(lanes.S, cur):
                    _mark("sample.begin", lanes.S)
                    if stamped is not None:
                        rb.draw_into(stamped)
                        out = c.out
                    else:
                        out = c.replay()
                    lanes.sample_done = lanes.event_on(lanes.S)
                    _mark("sample.done", lanes.S)
                lanes.ready[self._ptr0[slot]] = lanes.sample_done
                lanes.n_draws += 1
                lanes.ready_seq[self._ptr0[slot]] = lanes.n_draws
        self.replays += 1
        return out


_FREE_MAILBOXES = []   # (host ptr, device ptr) of mailboxes whose owner was collected


class GraphedDriverRun:
    """`common.function(driver.run)` for a DynamicStepDriver: the loop body
    (dynamic_step_driver.py:118-172) is captured once per environment output buffer (two: the
    environment alternates between them) and replayed; the loop condition
    `sum(counter) < num_steps` (:113) is evaluated on the host, AHEAD of the body it concerns:
    a body's contribution `~traj.is_boundary()` (:170) only depends on the step types it
    receives, i.e. on the previous body's output, so the step-counter kernel runs at the END of
    each body on the time step just produced and posts "the total if one more body runs" to a
    pinned-memory mailbox.  Before launching a body the host therefore already knows whether it
    will be the last one of the run: in the usual case (the run needs exactly ceil(num_steps/B)
    bodies) `run()` returns without waiting for the GPU at all -- the value it needs was posted
    by the previous run's last body, an iteration ago -- and only a run that has to make up for
    boundary steps waits for its own bodies.  (Counting at the START of the body and waiting for
    it, as round 1 did, put the host in lock-step with the GPU: it could not enqueue train(k)
    before the GPU had finished train(k-1).)  Same iteration count as the reference's in-graph
    while_loop in every case.

    Requirements, checked at first use: the environment supports `graph_ring()`, the policy has no
    state, observers are device-side (the replay buffer's add_batch) or tolerate tf.function-style
    tracing.  Otherwise every call falls through to the eager `driver.run`."""

    def __init__(self, driver):
        self._driver_ref = weakref.ref(driver)    # the driver owns this object (`_graphed_run`)
        self._graphs = {}
        self._warm = 0
        self._seq = 0                          # mailbox posts issued so far
        self._t_counted = 0                    # counted steps of every body launched so far
        self._pub = None                       # (env host_epoch, ring slot) the last post describes
        self.wait_seconds = 0.0                # host time spent waiting for mailbox posts
        self._total = None
        self._counter = None
        self._mbox_host = None
        self._mbox_dev = None
        self.enabled = hasattr(driver.env, "graph_ring") and not driver._transition_observers
        self.replays = 0

    @property
    def _driver(self):
        d = self._driver_ref()
        if d is None:
            raise ReferenceError(
                "the driver of this graphed run function has been released; keep a reference to "
                "the driver for as long as the function returned by common.function(driver.run) "
                "is used")
        return d

    def _eager_run(self, time_step=None, policy_state=None, maximum_iterations=None):
        # the class's own method: `collect_driver.run = common.function(collect_driver.run)`
        # (train_eval.py:234-237) rebinds the INSTANCE attribute to this object
        drv = self._driver
        return type(drv).run(drv, time_step, policy_state, maximum_iterations)

    def close(self):
        """Releases the recorded loop bodies (see `_GRAVEYARD`), their scratch and the mailbox."""
        graphs, self._graphs = self._graphs, {}
        for c in graphs.values():
            c.close()
        if graphs:
            from agents_amd import ops
            ops.release_scope(("collect", id(self)))
        # Back to the free list, NOT hipHostFree: a finaliser can run at any point, including in
        # the middle of another object's stream capture, where a synchronising HIP call would
        # invalidate the capture (seen as an intermittent crash in the test suite).  The next owner
        # synchronises before it resets the words (`_setup`): the bodies parked in the graveyard
        # are never replayed again, and the last replay in flight has finished by then.
        if self._mbox_host is not None:
            _FREE_MAILBOXES.append((self._mbox_host, self._mbox_dev))
            self._mbox_host = None
        self._total = self._counter = None
        self._pub = None
        self._seq = self._t_counted = 0

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown
            pass

    def _setup(self, dev, B):
        import ctypes
        from agents_amd import _lib
        lib = _lib.load()
        if _FREE_MAILBOXES:
            self._mbox_host, self._mbox_dev = _FREE_MAILBOXES.pop()
            torch.cuda.synchronize()              # no kernel of the previous owner is in flight
            ctypes.memset(self._mbox_host, 0, 16)
            self._seq = 0
        else:
            h, d = ctypes.c_void_p(), ctypes.c_void_p()
            _lib.check(lib.aa_mailbox_create(2, ctypes.byref(h), ctypes.byref(d)),
                       "aa_mailbox_create")
            self._mbox_host, self._mbox_dev = h.value, d.value
        self._total = torch.zeros((1,), dtype=torch.int64, device=dev)
        self._counter = torch.zeros((B,), dtype=torch.int32, device=dev)

    def _count(self, step_type):
        """Posts (mailbox) the counted total if a body consumes a time step of these types."""
        from agents_amd import _lib
        _lib.check(_lib.load().aa_count_steps(step_type.data_ptr(), step_type.numel(),
                                              self._counter.data_ptr(), self._total.data_ptr(),
                                              self._mbox_dev, _lib.stream_ptr()),
                   "aa_count_steps")

    def _body(self, time_step, policy_state):
        """One loop body, followed by the step count of the NEXT one (see the class docstring)."""
        from agents_amd.trajectories import trajectory
        drv = self._driver
        action_step = drv.policy.action(time_step, policy_state)
        next_time_step = drv.env.step(action_step.action)
        traj = trajectory.from_transition(time_step, action_step, next_time_step)
        # The step count of the NEXT body rides in the replay buffer's add_batch launch when the
        # first observer is one (csrc/replay.hip: aa_rb_scatter_rows_count): one launch and one
        # graph node less per body.  Otherwise its own launch, as early as possible (the host of
        # the next run is waiting for the post).
        fused = None
        if COUNT_IN_ADD and drv._observers:
            owner = getattr(drv._observers[0], "__self__", None)
            if getattr(drv._observers[0], "__name__", "") == "add_batch" and \
                    getattr(owner, "supports_counting_add", lambda: False)() and \
                    next_time_step.step_type.dtype == torch.int32:
                fused = owner
        if fused is None:
            self._count(next_time_step.step_type)
        for k, observer in enumerate(drv._observers):
            if k == 0 and fused is not None:
                fused.add_batch_counting(traj, next_time_step.step_type, self._counter,
                                         self._total, self._mbox_dev)
            else:
                observer(traj)
        return next_time_step

    def _read_post(self):
        """Value of the latest post (waits for it if the GPU has not got there yet)."""
        import ctypes
        from agents_amd import _lib
        v = ctypes.c_int64(0)
        t0 = time.perf_counter()
        _lib.check(_lib.load().aa_mailbox_wait(self._mbox_host, self._seq, 60_000_000,
                                               ctypes.byref(v)), "aa_mailbox_wait")
        self.wait_seconds += time.perf_counter() - t0
        return int(v.value)

    def _epoch(self):
        return getattr(self._driver.env, "host_epoch", None)

    def __call__(self, time_step=None, policy_state=None, maximum_iterations=None):
        drv = self._driver
        env = drv.env
        if not self.enabled or capturing():
            return self._eager_run(time_step, policy_state, maximum_iterations)
        if policy_state is None:
            policy_state = drv.policy.get_initial_state(env.batch_size)
        if policy_state == ():
            # from the first call on the environment alternates between its two ring buffers, so
            # the eager warm-up calls already leave the loop where the graphs will pick it up
            env.graph_ring()
        if self._warm < _WARMUP_CALLS or policy_state != ():
            self._warm += 1
            return self._eager_run(time_step, policy_state, maximum_iterations)
        _ensure_prepared()
        if time_step is None:
            time_step = env.current_time_step()
        st = time_step.step_type
        if st.dim() != 1 or st.dtype != torch.int32 or not st.is_cuda:
            return self._eager_run(time_step, policy_state, maximum_iterations)
        B = st.numel()
        with _device_ctx(st.device):
            if self._total is None:
                self._setup(st.device, B)
            ring = env.graph_ring()
            num_steps = drv._num_steps
            n_min = -(-num_steps // B)
            lanes = lanes_for(st.device)
            base = None
            it = 0
            while maximum_iterations is None or it < maximum_iterations:
                if num_steps <= 0:
                    break         # `counter < num_steps` is false from the start: no body runs
                if it >= n_min and self._t_counted - base >= num_steps:
                    break
                slot = ring.slot_of(time_step)
                if slot is None or slot != ring.slot_of(env._time_step):
                    # a TimeStep that is not the environment's current ring buffer (first calls,
                    # or a caller-made one): one eager run brings the loop into the ring
                    self._pub = None
                    return self._eager_run(time_step, policy_state,
                                           None if maximum_iterations is None
                                           else maximum_iterations - it)
                c = self._graphs.get(slot)
                if c is None:
                    # Both loop bodies are captured now (the environment alternates between its
                    # two output buffers): capturing the body on slot s leaves the environment's
                    # host-side reference on slot 1-s -- exactly the input of the other body --
                    # and capturing that one brings it back to s, the true current step.
                    join_lanes(st.device)
                    from agents_amd import ops
                    try:
                        with capture_batch():
                            for k in (slot, 1 - slot):
                                ck = _Captured("collect")
                                ts_in = ring.slots[k]
                                # private GEMM scratch: may replay next to the train graphs
                                with ops.workspace_scope(("collect", id(self)), st.device):
                                    ck.capture(lambda: self._body(ts_in, policy_state))
                                self._graphs[k] = ck
                    except Exception:
                        self.enabled = False
                        self._graphs.clear()
                        raise
                    c = self._graphs[slot]
                if lanes is not None and it == 0:
                    cur = current_stream(st.device)
                    lanes.C.wait_event(lanes.event_on(cur))      # = the main frontier
                    if lanes.sample_done is not None:
                        lanes.C.wait_event(lanes.sample_done)
                if self._pub != (self._epoch(), slot) or self._epoch() is None:
                    # nothing has posted the count of THIS time step (first graphed run, an eager
                    # step in between, an environment without `host_epoch`): post it now.  What
                    # the abandoned post had added to the device total counts as consumed.
                    self._t_counted = self._read_post()
                    with _on_stream(lanes.C, current_stream(st.device)) \
                            if lanes is not None else contextlib.nullcontext():
                        self._count(time_step.step_type)
                    self._seq += 1
                if base is None:
                    base = self._t_counted
                # the body about to be launched is counted by the latest post.  A body that cannot
                # be the run's last (fewer than n_min bodies with it) is launched WITHOUT reading
                # it: the loop condition is not evaluated before body n_min anyway, and the read
                # made the host wait for the GPU once per body -- a run of 129 bodies (PPO:
                # num_steps = envs x (T + 1)) took 155 us per body for ~70 us of kernels.  The
                # post read in front of body n_min - 1 is cumulative, so nothing is lost.
                if maximum_iterations is not None or it + 1 >= n_min:
                    self._t_counted = self._read_post()
                if lanes is None:
                    time_step = c.replay()
                else:
                    if it > 0:
                        cur = current_stream(st.device)
                    with _on_stream(lanes.C, cur):
                        _mark("collect.begin", lanes.C)
                        time_step = c.replay()
                        lanes.collect_done = lanes.event_on(lanes.C)
                        _mark("collect.done", lanes.C)
                self._seq += 1
                self._pub = (self._epoch(), ring.slot_of(time_step))
                self.replays += 1
                it += 1
        return time_step, policy_state


def graphed_driver_run(driver):
    g = getattr(driver, "_graphed_run", None)
    if g is None:
        g = GraphedDriverRun(driver)
        driver._graphed_run = g
    return g
