"""A small Dataset pipeline object standing in for the tf.data.Dataset that
`ReplayBuffer.as_dataset` returns in the reference (replay_buffer.py:134-243).

The scripts chain a few combinators on it: `.prefetch(3)` then `iter()` (DQN train_eval.py:276-281),
`.unbatch().filter(...).batch(n).prefetch(5)` (SAC train_eval.py:285-296),
`.take/.cache/.repeat/.map/.shuffle/.batch` (train/ppo_learner.py:220-248).  Elements are nests of
device tensors; `prefetch(n)` enqueues n elements ahead on the HIP stream (sampling kernels are
asynchronous, so running ahead is what hides their latency).
"""
import collections

import torch

from agents_amd.utils import nest_utils


class Dataset:
    def __init__(self, make_iter, infinite=False):
        self._make_iter = make_iter
        self._infinite = infinite

    def __iter__(self):
        return iter(self._make_iter())

    # ---- combinators ----------------------------------------------------------------------
    def prefetch(self, buffer_size):
        n = max(int(buffer_size), 0)
        src = self

        def gen():
            it = iter(src)
            q = collections.deque()
            done = False
            while True:
                while not done and len(q) <= n:
                    try:
                        q.append(next(it))
                    except StopIteration:
                        done = True
                if not q:
                    return
                yield q.popleft()

        return Dataset(gen, self._infinite)

    def map(self, fn, num_parallel_calls=None, deterministic=None):
        src = self

        def gen():
            for e in src:
                yield fn(*e) if isinstance(e, tuple) and not hasattr(e, "_fields") else fn(e)

        return Dataset(gen, self._infinite)

    def filter(self, pred):
        src = self

        def gen():
            for e in src:
                keep = pred(*e) if isinstance(e, tuple) and not hasattr(e, "_fields") else pred(e)
                if isinstance(keep, torch.Tensor):
                    keep = bool(keep.item())
                if keep:
                    yield e

        return Dataset(gen, self._infinite)

    def take(self, count):
        src = self

        def gen():
            if count == 0:
                return
            for i, e in enumerate(src):
                yield e
                if count > 0 and i + 1 >= count:
                    return

        return Dataset(gen, self._infinite and count < 0)

    def repeat(self, count=None):
        src = self

        def gen():
            k = 0
            while count is None or k < count:
                empty = True
                for e in src:
                    empty = False
                    yield e
                if empty:
                    return
                k += 1

        return Dataset(gen, self._infinite or count is None)

    def cache(self):
        src = self
        store = []
        state = {"full": False}

        def gen():
            if state["full"]:
                yield from store
                return
            for e in src:
                store.append(e)
                yield e
            state["full"] = True

        return Dataset(gen, self._infinite)

    def unbatch(self):
        src = self

        def gen():
            for e in src:
                leaves = [l for l in nest_utils.flatten(e) if isinstance(l, torch.Tensor)]
                n = leaves[0].shape[0]
                for i in range(n):
                    yield nest_utils.map_structure(
                        lambda l: l[i] if isinstance(l, torch.Tensor) else l, e)

        return Dataset(gen, self._infinite)

    def batch(self, batch_size, drop_remainder=False):
        src = self

        def gen():
            buf = []
            for e in src:
                buf.append(e)
                if len(buf) == batch_size:
                    yield nest_utils.map_structure(lambda *ls: torch.stack(ls, 0), *buf)
                    buf = []
            if buf and not drop_remainder:
                yield nest_utils.map_structure(lambda *ls: torch.stack(ls, 0), *buf)

        return Dataset(gen, self._infinite)

    def shuffle(self, buffer_size, seed=None, reshuffle_each_iteration=True):
        src = self
        epoch = {"n": 0}

        def gen():
            g = torch.Generator()
            g.manual_seed((0 if seed is None else int(seed)) + epoch["n"])
            if reshuffle_each_iteration:
                epoch["n"] += 1
            buf = []
            for e in src:
                buf.append(e)
                if len(buf) >= buffer_size:
                    j = int(torch.randint(len(buf), (1,), generator=g))
                    buf[j], buf[-1] = buf[-1], buf[j]
                    yield buf.pop()
            while buf:
                j = int(torch.randint(len(buf), (1,), generator=g))
                buf[j], buf[-1] = buf[-1], buf[j]
                yield buf.pop()

        return Dataset(gen, self._infinite)

    def with_options(self, options):
        return self
