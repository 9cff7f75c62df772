"""HIP-graph capture of an agent's train step: the MI355X counterpart of wrapping `agent.train`
in `common.function` (tf.function) as the reference's scripts do
(tf_agents/agents/dqn/examples/v2/train_eval.py:234-237) and as `Learner._train` is
(tf_agents/train/learner.py:309-337).

A train step is ~45 short kernels; launched one by one from Python it is host-bound (~14 us per
launch against 5-50 us of GPU work each).  `GraphedTrain` runs the step eagerly twice (buffers
and workspaces get created), then captures its two device phases -- [forwards + loss + backward]
and [optimizer] -- into hipGraphs on static input buffers and replays them: per step the host
issues a handful of copies of the sampled batch into the static buffers plus two graph launches.
The gradient hook (the Learner's RCCL all-reduce) runs between the two graphs, outside capture.
Host bookkeeping (train_step_counter, optimizer.iterations, the periodic target update decided by
a host counter) stays in Python, exactly as in the eager path.
"""
import torch

from agents_amd.utils import nest_utils

_WARMUP_CALLS = 2


class _Entry:
    def __init__(self):
        self.calls = 0
        self.static_in = None
        self.static_w = None
        self.g_grads = None
        self.g_apply = None
        self.out = None


def _sig(experience, weights):
    leaves = nest_utils.flatten(experience)
    s = tuple((tuple(t.shape), t.dtype, t.device) for t in leaves)
    if isinstance(weights, torch.Tensor):
        return s + ((tuple(weights.shape), weights.dtype),)
    return s + (weights,)


class GraphedTrain:
    """Callable with the signature of `agent.train`; falls back to the eager path for agents
    that do not expose graphable phases."""

    def __init__(self, agent):
        self._agent = agent
        self._cache = {}
        self.enabled = all(hasattr(agent, n) for n in
                           ("_train_phase_grads", "_train_phase_apply", "_train_phase_host"))
        self.replays = 0

    @property
    def agent(self):
        return self._agent

    def __call__(self, experience, weights=None, **kwargs):
        agent = self._agent
        if not self.enabled or kwargs or getattr(agent, "check_numerics", False):
            return agent.train(experience, weights=weights, **kwargs)
        key = _sig(experience, weights)
        e = self._cache.get(key)
        if e is None:
            e = self._cache[key] = _Entry()
        if e.calls < _WARMUP_CALLS:
            e.calls += 1
            return agent.train(experience, weights=weights)
        if not agent._initialized:
            agent.initialize()
        agent._check_trajectory(experience)
        dev = experience.discount.device
        with torch.cuda.device(dev):
            if e.g_grads is None:
                self._capture(e, experience, weights)
            # copy the sampled batch into the graph's static inputs (skipped when the caller
            # already wrote into them, e.g. a sampler bound to `static_inputs()`)
            for dst, src in zip(nest_utils.flatten(e.static_in), nest_utils.flatten(experience)):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
            if e.static_w is not None and e.static_w.data_ptr() != weights.data_ptr():
                e.static_w.copy_(weights, non_blocking=True)
            e.g_grads.replay()
            if agent.gradient_hook is not None:
                agent.gradient_hook(agent._q_network.flat_grads)
            e.g_apply.replay()
            agent._optimizer.iterations += 1
            agent._train_phase_host()
        self.replays += 1
        return e.out

    def _capture(self, e, experience, weights):
        agent = self._agent
        e.static_in = nest_utils.map_structure(lambda t: t.clone(), experience)
        e.static_w = weights.clone() if isinstance(weights, torch.Tensor) else None
        w_arg = e.static_w if e.static_w is not None else weights
        torch.cuda.synchronize()
        iters = agent._optimizer.iterations
        e.g_grads = torch.cuda.CUDAGraph()
        with torch.cuda.graph(e.g_grads):
            e.out = agent._train_phase_grads(e.static_in, w_arg)
        e.g_apply = torch.cuda.CUDAGraph()
        with torch.cuda.graph(e.g_apply):
            agent._train_phase_apply()
        agent._optimizer.iterations = iters  # capture enqueues nothing; undo the host mirror bump
        torch.cuda.synchronize()

    def static_inputs(self, experience_like=None):
        """Static input nest of the (single) captured signature, or None before capture."""
        for e in self._cache.values():
            if e.static_in is not None:
                return e.static_in
        return None


def graphed_train(agent):
    """One GraphedTrain per agent (shared by common.function and the Learner)."""
    g = getattr(agent, "_graphed_train", None)
    if g is None:
        g = GraphedTrain(agent)
        agent._graphed_train = g
    return g
