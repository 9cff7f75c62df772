"""TFAgent base class and LossInfo (tf_agents/agents/tf_agent.py:37,41-648, hot-path subset):
initialize / train / loss / preprocess_sequence wrappers with the same argument validation
(experience structure vs training_data_spec, [B, T] outer dims, T == train_sequence_length;
agents/data_converter.py:175-231)."""
import collections

from agents_amd.trajectories import trajectory as trajectory_lib
from agents_amd.utils import common, nest_utils

LossInfo = collections.namedtuple("LossInfo", ("loss", "extra"))


class TFAgent:
    def __init__(self, time_step_spec, action_spec, policy, collect_policy,
                 train_sequence_length, num_outer_dims=2, training_data_spec=None,
                 train_argspec=None, debug_summaries=False, summarize_grads_and_vars=False,
                 enable_summaries=True, train_step_counter=None, validate_args=True):
        self._time_step_spec = time_step_spec
        self._action_spec = action_spec
        self._policy = policy
        self._collect_policy = collect_policy
        self._train_sequence_length = train_sequence_length
        self._num_outer_dims = num_outer_dims
        self._debug_summaries = debug_summaries
        self._summarize_grads_and_vars = summarize_grads_and_vars
        self._validate_args = validate_args
        if train_step_counter is None:
            train_step_counter = common.Variable(0, name="train_step")
        self._train_step_counter = train_step_counter
        self._collect_data_spec = collect_policy.trajectory_spec
        self._training_data_spec = training_data_spec or self._collect_data_spec
        self._initialized = False

    # ---- properties (tf_agent.py:493-560) --------------------------------------------------------
    @property
    def time_step_spec(self):
        return self._time_step_spec

    @property
    def action_spec(self):
        return self._action_spec

    @property
    def policy(self):
        return self._policy

    @property
    def collect_policy(self):
        return self._collect_policy

    @property
    def collect_data_spec(self):
        return self._collect_data_spec

    @property
    def training_data_spec(self):
        return self._training_data_spec

    @property
    def train_sequence_length(self):
        return self._train_sequence_length

    @property
    def train_step_counter(self):
        return self._train_step_counter

    @property
    def debug_summaries(self):
        return self._debug_summaries

    @property
    def summarize_grads_and_vars(self):
        return self._summarize_grads_and_vars

    # ---- public API ---------------------------------------------------------------------------
    def initialize(self):
        self._initialized = True
        return self._initialize()

    def preprocess_sequence(self, experience):
        return self._preprocess_sequence(experience)

    def _preprocess_sequence(self, experience):
        return experience

    def _check_trajectory(self, experience):
        """Structure and [B, T] + spec.shape validation (data_converter.py:175-231)."""
        if not self._validate_args:
            return
        if not isinstance(experience, trajectory_lib.Trajectory):
            raise ValueError(f"experience must be a Trajectory, got {type(experience).__name__}")
        nest_utils.assert_same_structure(
            experience, self._training_data_spec,
            message="experience and training_data_spec structures do not match")
        rank = nest_utils.get_outer_rank(experience, self._training_data_spec)
        if rank != self._num_outer_dims:
            raise ValueError(
                "All of the Tensors in `experience` must have two outer dimensions: batch size "
                f"and time; saw outer rank {rank}")
        if self._train_sequence_length is not None:
            T = experience.discount.shape[1]
            if T != self._train_sequence_length:
                raise ValueError(
                    "The agent was configured to expect a `train_sequence_length` of "
                    f"'{self._train_sequence_length}'. Experience is expected to be shaped "
                    f"`[Batch x Time x ...]` but saw time dimension {T}.")

    def train(self, experience, weights=None, **kwargs):
        if not self._initialized:
            self.initialize()
        self._check_trajectory(experience)
        loss_info = self._train(experience=experience, weights=weights, **kwargs)
        if not isinstance(loss_info, LossInfo):
            raise TypeError(f"loss_info is not a subclass of LossInfo: {loss_info}")
        return loss_info

    def loss(self, experience, weights=None, training=False, **kwargs):
        self._check_trajectory(experience)
        loss_info = self._loss(experience, weights=weights, training=training, **kwargs)
        if not isinstance(loss_info, LossInfo):
            raise TypeError(f"loss_info is not a subclass of LossInfo: {loss_info}")
        return loss_info

    def _initialize(self):
        raise NotImplementedError

    def _train(self, experience, weights):
        raise NotImplementedError

    def _loss(self, experience, weights, training=False, **kwargs):
        raise NotImplementedError
