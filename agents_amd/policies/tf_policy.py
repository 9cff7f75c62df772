"""TFPolicy base class (subset of tf_agents/policies/tf_policy.py:643): action(),
get_initial_state(), and the spec properties (time_step_spec, action_spec, policy_state_spec,
info_spec, policy_step_spec, trajectory_spec, collect_data_spec)."""
from agents_amd.trajectories import policy_step, trajectory
from agents_amd.utils import nest_utils


class TFPolicy:
    def __init__(self, time_step_spec, action_spec, policy_state_spec=(), info_spec=(),
                 clip=True, emit_log_probability=False, automatic_state_reset=True,
                 observation_and_action_constraint_splitter=None, name=None):
        self._time_step_spec = time_step_spec
        self._action_spec = action_spec
        self._policy_state_spec = policy_state_spec
        self._info_spec = info_spec
        self._emit_log_probability = emit_log_probability
        self._observation_and_action_constraint_splitter = \
            observation_and_action_constraint_splitter
        self._name = name or type(self).__name__
        self._policy_step_spec = policy_step.PolicyStep(action=action_spec,
                                                        state=policy_state_spec, info=info_spec)
        self._trajectory_spec = trajectory.from_transition_spec(time_step_spec, action_spec,
                                                                info_spec)

    @property
    def time_step_spec(self):
        return self._time_step_spec

    @property
    def action_spec(self):
        return self._action_spec

    @property
    def policy_state_spec(self):
        return self._policy_state_spec

    @property
    def info_spec(self):
        return self._info_spec

    @property
    def policy_step_spec(self):
        return self._policy_step_spec

    @property
    def trajectory_spec(self):
        return self._trajectory_spec

    @property
    def collect_data_spec(self):
        return self._trajectory_spec

    @property
    def emit_log_probability(self):
        return self._emit_log_probability

    @property
    def observation_and_action_constraint_splitter(self):
        return self._observation_and_action_constraint_splitter

    def variables(self):
        return self._variables()

    def _variables(self):
        return []

    def get_initial_state(self, batch_size=None):
        return self._get_initial_state(batch_size)

    def _get_initial_state(self, batch_size):
        return ()

    def action(self, time_step, policy_state=(), seed=None):
        """PolicyStep(action, state, info) for a batch of time steps (tf_policy.py:276)."""
        nest_utils.assert_same_structure(time_step, self._time_step_spec)
        return self._action(time_step, policy_state, seed)

    def _action(self, time_step, policy_state, seed):
        raise NotImplementedError

    def distribution(self, time_step, policy_state=()):
        return self._distribution(time_step, policy_state)

    def _distribution(self, time_step, policy_state):
        raise NotImplementedError("distributions are not materialised; use action()")

    def update(self, policy, tau=1.0, tau_non_trainable=None, sort_variables_by_name=False):
        from agents_amd.utils import common
        return common.soft_variables_update(policy.variables(), self.variables(), tau=tau)
