r"""Train and eval PPO (clip agent) on the MI355X hot path: the same `train_eval(...)` entry point,
keyword arguments, defaults and loop structure as
tf_agents/agents/ppo/examples/v2/train_eval_clip_agent.py:94-331, with the import root changed from
`tf_agents` to `agents_amd`.

  python -m agents_amd.agents.ppo.examples.v2.train_eval_clip_agent --root_dir=/tmp/ppo \
      --num_environment_steps=200000

Per iteration (train_eval_clip_agent.py:262-276): `collect_driver.run()` (DynamicEpisodeDriver,
`collect_episodes_per_iteration` episodes over the parallel environments, observers =
[replay_buffer.add_batch] + train metrics) -> `replay_buffer.gather_all()` ->
`tf_agent.train(experience=trajectories)` (GAE + `num_epochs` epochs inside the agent) ->
`replay_buffer.clear()`.

What differs, and why:
  * environments: no MuJoCo in this image; `env_name` selects a device-resident synthetic
    environment with HalfCheetah's specs (`environments/suite_synthetic.py`) and
    `num_parallel_environments` is its batch size (the reference wraps that many py environments
    in a ParallelPyEnvironment); pass `env_load_fn(env_name, batch_size=...)` for anything else.
  * `use_rnns=True`, summary writers, the SavedModel `PolicySaver` and gin files are outside the
    hot-path scope (SURVEY.md section 8 / DESIGN.md section 7).
"""
import argparse
import logging
import os
import time

from agents_amd import optimizers
from agents_amd.agents.ppo import ppo_clip_agent
from agents_amd.drivers import dynamic_episode_driver
from agents_amd.environments import suite_synthetic
from agents_amd.eval import metric_utils
from agents_amd.metrics import tf_metrics
from agents_amd.networks import actor_distribution_network
from agents_amd.networks import value_network
from agents_amd.replay_buffers import tf_uniform_replay_buffer
from agents_amd.utils import common

try:
    import gin
except ImportError:                                      # gin is not part of this image
    class gin:                                           # noqa: N801
        @staticmethod
        def configurable(fn=None, **_):
            return fn if fn is not None else (lambda f: f)


@gin.configurable
def train_eval(
        root_dir,
        env_name="HalfCheetah-v2",
        env_load_fn=suite_synthetic.load,
        random_seed=None,
        actor_fc_layers=(200, 100),
        value_fc_layers=(200, 100),
        use_rnns=False,
        lstm_size=(20,),
        # Params for collect
        num_environment_steps=25000000,
        collect_episodes_per_iteration=30,
        num_parallel_environments=30,
        replay_buffer_capacity=1001,  # Per-environment
        # Params for train
        num_epochs=25,
        learning_rate=1e-3,
        # Params for eval
        num_eval_episodes=30,
        eval_interval=500,
        # Params for summaries and logging
        train_checkpoint_interval=500,
        policy_checkpoint_interval=500,
        log_interval=50,
        summary_interval=50,
        summaries_flush_secs=1,
        use_tf_functions=True,
        debug_summaries=False,
        summarize_grads_and_vars=False,
        # Additions (trailing, optional)
        eval_metrics_callback=None):
    """A simple train and eval for PPO (same contract as the reference's function)."""
    if root_dir is None:
        raise AttributeError("train_eval requires a root_dir.")
    if use_rnns:
        raise NotImplementedError("recurrent actor / value networks are outside the hot-path "
                                  "scope (DESIGN.md section 7)")
    root_dir = os.path.expanduser(root_dir)
    train_dir = os.path.join(root_dir, "train")

    global_step = common.Variable(0, name="global_step")
    seed_kw = {} if random_seed is None else {"seed": int(random_seed)}
    eval_tf_env = env_load_fn(env_name, batch_size=1, **seed_kw)
    eval_metrics = [
        tf_metrics.AverageReturnMetric(buffer_size=num_eval_episodes,
                                       batch_size=eval_tf_env.batch_size),
        tf_metrics.AverageEpisodeLengthMetric(buffer_size=num_eval_episodes,
                                              batch_size=eval_tf_env.batch_size),
    ]
    tf_env = env_load_fn(env_name, batch_size=num_parallel_environments, **seed_kw)
    optimizer = optimizers.AdamOptimizer(learning_rate=learning_rate)

    actor_net = actor_distribution_network.ActorDistributionNetwork(
        tf_env.observation_spec(), tf_env.action_spec(), fc_layer_params=actor_fc_layers,
        activation_fn="tanh", seed=random_seed)
    value_net = value_network.ValueNetwork(
        tf_env.observation_spec(), fc_layer_params=value_fc_layers, activation_fn="tanh",
        seed=None if random_seed is None else random_seed + 1)

    tf_agent = ppo_clip_agent.PPOClipAgent(
        tf_env.time_step_spec(), tf_env.action_spec(), optimizer, actor_net=actor_net,
        value_net=value_net, entropy_regularization=0.0, importance_ratio_clipping=0.2,
        normalize_observations=False, normalize_rewards=False, use_gae=True,
        num_epochs=num_epochs, debug_summaries=debug_summaries,
        summarize_grads_and_vars=summarize_grads_and_vars, train_step_counter=global_step)
    tf_agent.initialize()

    environment_steps_metric = tf_metrics.EnvironmentSteps()
    step_metrics = [tf_metrics.NumberOfEpisodes(), environment_steps_metric]
    train_metrics = step_metrics + [
        tf_metrics.AverageReturnMetric(batch_size=num_parallel_environments),
        tf_metrics.AverageEpisodeLengthMetric(batch_size=num_parallel_environments),
    ]

    eval_policy = tf_agent.policy
    collect_policy = tf_agent.collect_policy

    replay_buffer = tf_uniform_replay_buffer.TFUniformReplayBuffer(
        tf_agent.collect_data_spec, batch_size=num_parallel_environments,
        max_length=replay_buffer_capacity)

    train_checkpointer = common.Checkpointer(
        ckpt_dir=train_dir, agent=tf_agent, global_step=global_step,
        metrics=metric_utils.MetricsGroup(train_metrics, "train_metrics"))
    policy_checkpointer = common.Checkpointer(
        ckpt_dir=os.path.join(train_dir, "policy"), policy=eval_policy, global_step=global_step)
    train_checkpointer.initialize_or_restore()

    collect_driver = dynamic_episode_driver.DynamicEpisodeDriver(
        tf_env, collect_policy, observers=[replay_buffer.add_batch] + train_metrics,
        num_episodes=collect_episodes_per_iteration)

    def train_step():
        trajectories = replay_buffer.gather_all()
        return tf_agent.train(experience=trajectories)

    if use_tf_functions:
        # (the episode driver's loop length is data dependent and gather_all's batch shape changes
        # from iteration to iteration: both stay eager launches here; the agent's own kernels are
        # what a tf.function would have fused)
        collect_driver.run = common.function(collect_driver.run, autograph=False)
        tf_agent.train = common.function(tf_agent.train, autograph=False)
        train_step = common.function(train_step)

    def evaluate():
        results = metric_utils.eager_compute(
            eval_metrics, eval_tf_env, eval_policy, num_episodes=num_eval_episodes,
            train_step=global_step, summary_prefix="Metrics")
        if eval_metrics_callback is not None:
            eval_metrics_callback(results, global_step.numpy())
        metric_utils.log_metrics(eval_metrics)
        return results

    collect_time = 0
    train_time = 0
    timed_at_step = global_step.numpy()
    total_loss = None

    while environment_steps_metric.result() < num_environment_steps:
        global_step_val = global_step.numpy()
        if global_step_val % eval_interval == 0:
            evaluate()

        start_time = time.time()
        collect_driver.run()
        collect_time += time.time() - start_time

        start_time = time.time()
        total_loss, _ = train_step()
        replay_buffer.clear()
        train_time += time.time() - start_time

        if global_step_val % log_interval == 0:
            logging.info("step = %d, loss = %f", global_step_val, float(total_loss))
            steps_per_sec = (global_step_val - timed_at_step) / max(collect_time + train_time,
                                                                    1e-9)
            logging.info("%.3f steps/sec", steps_per_sec)
            logging.info("collect_time = %.3f, train_time = %.3f", collect_time, train_time)
            if global_step_val % train_checkpoint_interval == 0:
                train_checkpointer.save(global_step=global_step_val)
            if global_step_val % policy_checkpoint_interval == 0:
                policy_checkpointer.save(global_step=global_step_val)
            timed_at_step = global_step_val
            collect_time = 0
            train_time = 0

    # One final eval before exiting.
    evaluate()
    train_eval.last_run = dict(agent=tf_agent, replay_buffer=replay_buffer, actor_net=actor_net,
                               value_net=value_net, train_metrics=train_metrics,
                               global_step=global_step)
    return total_loss


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--root_dir", required=True)
    ap.add_argument("--env_name", default="HalfCheetah-v2")
    ap.add_argument("--replay_buffer_capacity", type=int, default=1001)
    ap.add_argument("--num_parallel_environments", type=int, default=30)
    ap.add_argument("--num_environment_steps", type=int, default=25000000)
    ap.add_argument("--num_epochs", type=int, default=25)
    ap.add_argument("--collect_episodes_per_iteration", type=int, default=30)
    ap.add_argument("--num_eval_episodes", type=int, default=30)
    ap.add_argument("--use_rnns", action="store_true")
    a = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    train_eval(a.root_dir, env_name=a.env_name, use_rnns=a.use_rnns,
               num_environment_steps=a.num_environment_steps,
               collect_episodes_per_iteration=a.collect_episodes_per_iteration,
               num_parallel_environments=a.num_parallel_environments,
               replay_buffer_capacity=a.replay_buffer_capacity, num_epochs=a.num_epochs,
               num_eval_episodes=a.num_eval_episodes)


if __name__ == "__main__":
    main()
