"""ReplayBuffer abstract base: same public surface as tf_agents/replay_buffers/replay_buffer.py:31-315."""
import abc

from agents_amd.utils import nest_utils


class ReplayBuffer(abc.ABC):
    """add_batch / get_next / as_dataset / gather_all / clear / num_frames."""

    def __init__(self, data_spec, capacity, stateful_dataset=False):
        self._data_spec = data_spec
        self._capacity = capacity
        self._stateful_dataset = stateful_dataset

    @property
    def data_spec(self):
        return self._data_spec

    @property
    def capacity(self):
        return self._capacity

    @property
    def stateful_dataset(self):
        return self._stateful_dataset

    def num_frames(self):
        return self._num_frames()

    def add_batch(self, items):
        return self._add_batch(items)

    def get_next(self, sample_batch_size=None, num_steps=None, time_stacked=True):
        return self._get_next(sample_batch_size, num_steps, time_stacked)

    def as_dataset(self, sample_batch_size=None, num_steps=None, num_parallel_calls=None,
                   sequence_preprocess_fn=None, single_deterministic_pass=False):
        """Dataset of (items, BufferInfo); see replay_buffer.py:134-243 for the contract."""
        if nest_utils.has_lists(self._data_spec):
            raise ValueError(
                "Cannot perform gather; data spec contains lists and this conflicts with "
                "gathering operator.  Convert any lists to tuples.  For example, if your spec "
                "looks like [a, b, c], change it to (a, b, c).  Spec structure is:\n  {}".format(
                    nest_utils.map_structure(lambda s: s.dtype, self._data_spec)))
        if single_deterministic_pass:
            return self._single_deterministic_pass_dataset(
                sample_batch_size=sample_batch_size, num_steps=num_steps,
                sequence_preprocess_fn=sequence_preprocess_fn,
                num_parallel_calls=num_parallel_calls)
        return self._as_dataset(sample_batch_size=sample_batch_size, num_steps=num_steps,
                                sequence_preprocess_fn=sequence_preprocess_fn,
                                num_parallel_calls=num_parallel_calls)

    def gather_all(self):
        return self._gather_all()

    def clear(self):
        return self._clear()

    @abc.abstractmethod
    def _num_frames(self):
        ...

    @abc.abstractmethod
    def _add_batch(self, items):
        ...

    @abc.abstractmethod
    def _get_next(self, sample_batch_size, num_steps, time_stacked):
        ...

    @abc.abstractmethod
    def _as_dataset(self, sample_batch_size, num_steps, sequence_preprocess_fn,
                    num_parallel_calls):
        ...

    @abc.abstractmethod
    def _single_deterministic_pass_dataset(self, sample_batch_size, num_steps,
                                           sequence_preprocess_fn, num_parallel_calls):
        ...

    @abc.abstractmethod
    def _gather_all(self):
        ...

    @abc.abstractmethod
    def _clear(self):
        ...
