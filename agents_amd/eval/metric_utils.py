"""metric_utils of the train_eval scripts (tf_agents/eval/metric_utils.py): `eager_compute` runs
`num_episodes` episodes of a policy on an environment with the metrics as observers and returns
their results; `MetricsGroup` bundles metrics for a Checkpointer; `log_metrics` logs them."""
import collections
import logging

from agents_amd.drivers import dynamic_episode_driver


class MetricsGroup:
    """metric_utils.py:33-47: lets `common.Checkpointer(metrics=MetricsGroup(...))` save and
    restore the step metrics with the agent."""

    def __init__(self, metrics, name=None):
        self.name = name
        self.metrics = list(metrics)

    def results(self):
        return collections.OrderedDict((m.name, m.result()) for m in self.metrics)

    def state_dict(self):
        return {m.name: m.state_dict() for m in self.metrics}

    def load_state_dict(self, sd):
        for m in self.metrics:
            if m.name in sd:
                m.load_state_dict(sd[m.name])


def log_metrics(metrics, prefix=""):
    """metric_utils.py:50-53."""
    logging.info("%s \n\t\t %s", prefix,
                 "\n\t\t ".join(f"{m.name} = {m.result()}" for m in metrics))


def eager_compute(metrics, environment, policy, num_episodes=1, train_step=None,
                  summary_writer=None, summary_prefix="", use_function=True):
    """metric_utils.py:135-201: resets the metrics, runs a DynamicEpisodeDriver for `num_episodes`
    episodes from a fresh `environment.reset()`, returns OrderedDict(name -> result).
    (`summary_writer` is accepted and ignored: summaries are outside the hot-path scope.)"""
    for m in metrics:
        m.reset()
    time_step = environment.reset()
    policy_state = policy.get_initial_state(environment.batch_size)
    driver = dynamic_episode_driver.DynamicEpisodeDriver(environment, policy, observers=metrics,
                                                         num_episodes=num_episodes)
    driver.run(time_step, policy_state)
    return collections.OrderedDict((m.name, m.result()) for m in metrics)
