r"""Train and eval SAC on the MI355X hot path: the same `train_eval(...)` entry point, keyword
arguments, defaults and loop structure as tf_agents/agents/sac/examples/v2/train_eval.py:70-355,
with the import root changed from `tf_agents` to `agents_amd`.

  python -m agents_amd.agents.sac.examples.v2.train_eval --root_dir=/tmp/sac --num_iterations=20000

The loop (train_eval.py:253-318): initial collect with a RandomTFPolicy -> per iteration
`collect_driver.run()` (DynamicStepDriver, observers = [replay_buffer.add_batch] + train metrics)
and `train_steps_per_iteration` x `tf_agent.train(next(iterator))`, where the iterator is the
reference's dataset pipeline verbatim:

    replay_buffer.as_dataset(sample_batch_size=batch_size, num_steps=2)
        .unbatch().filter(_filter_invalid_transition).batch(batch_size).prefetch(5)

(the filter drops transitions that START on a boundary step; here the pipeline compacts the
surviving rows on the device: replay_buffers/dataset.py, csrc/replay.hip aa_rb_compact_*).

What differs, and why:
  * environments: no MuJoCo in this image; `env_name` selects a device-resident synthetic
    environment with the task's specs (`environments/suite_synthetic.py`); pass
    `env_load_fn(env_name)` for anything else.  The trailing `num_parallel_environments` is the
    environment (and replay) batch size -- 1 in the reference.
  * `td_errors_loss_fn` defaults to `common.element_wise_squared_loss` (= tf.math.squared_difference).
  * summary writers and gin files are outside the hot-path scope.
"""
import argparse
import functools
import logging
import os
import time

from agents_amd import optimizers
from agents_amd.agents.ddpg import critic_network
from agents_amd.agents.sac import sac_agent
from agents_amd.agents.sac import tanh_normal_projection_network
from agents_amd.drivers import dynamic_step_driver
from agents_amd.environments import suite_synthetic
from agents_amd.eval import metric_utils
from agents_amd.metrics import tf_metrics
from agents_amd.networks import actor_distribution_network
from agents_amd.policies import greedy_policy
from agents_amd.policies import random_tf_policy
from agents_amd.replay_buffers import tf_uniform_replay_buffer
from agents_amd.utils import common

try:
    import gin
except ImportError:                                      # gin is not part of this image
    class gin:                                           # noqa: N801
        @staticmethod
        def configurable(fn=None, **_):
            return fn if fn is not None else (lambda f: f)

        @staticmethod
        def parse_config_files_and_bindings(files, bindings):
            if files or bindings:
                raise RuntimeError("gin is not installed: pass arguments to train_eval() instead")


@gin.configurable
def train_eval(
        root_dir,
        env_name="HalfCheetah-v2",
        eval_env_name=None,
        env_load_fn=suite_synthetic.load,
        # The SAC paper reported:
        # Hopper and Cartpole results up to 1000000 iters,
        # Humanoid results up to 10000000 iters,
        # Other mujoco tasks up to 3000000 iters.
        num_iterations=3000000,
        actor_fc_layers=(256, 256),
        critic_obs_fc_layers=None,
        critic_action_fc_layers=None,
        critic_joint_fc_layers=(256, 256),
        # Params for collect
        # HalfCheetah and Ant take 10000 initial collection steps.  Other mujoco tasks take 1000.
        initial_collect_steps=10000,
        collect_steps_per_iteration=1,
        replay_buffer_capacity=1000000,
        # Params for target update
        target_update_tau=0.005,
        target_update_period=1,
        # Params for train
        train_steps_per_iteration=1,
        batch_size=256,
        actor_learning_rate=3e-4,
        critic_learning_rate=3e-4,
        alpha_learning_rate=3e-4,
        td_errors_loss_fn=common.element_wise_squared_loss,
        gamma=0.99,
        reward_scale_factor=0.1,
        gradient_clipping=None,
        use_tf_functions=True,
        # Params for eval
        num_eval_episodes=30,
        eval_interval=10000,
        # Params for summaries and logging
        train_checkpoint_interval=50000,
        policy_checkpoint_interval=50000,
        rb_checkpoint_interval=50000,
        log_interval=1000,
        summary_interval=1000,
        summaries_flush_secs=10,
        debug_summaries=False,
        summarize_grads_and_vars=False,
        eval_metrics_callback=None,
        # Additions (trailing, optional)
        num_parallel_environments=1):
    """A simple train and eval for SAC (same contract as the reference's function)."""
    root_dir = os.path.expanduser(root_dir)
    train_dir = os.path.join(root_dir, "train")

    global_step = common.Variable(0, name="global_step")

    train_env_load_fn = env_load_fn
    if env_load_fn is suite_synthetic.load:
        train_env_load_fn = functools.partial(suite_synthetic.load,
                                              batch_size=num_parallel_environments)
    tf_env = train_env_load_fn(env_name)
    eval_env_name = eval_env_name or env_name
    eval_tf_env = env_load_fn(eval_env_name)
    # (the reference's eval environment has batch size 1, the metrics' default; an env_load_fn
    # that returns a batched environment is followed)
    eval_metrics = [
        tf_metrics.AverageReturnMetric(buffer_size=num_eval_episodes,
                                       batch_size=eval_tf_env.batch_size),
        tf_metrics.AverageEpisodeLengthMetric(buffer_size=num_eval_episodes,
                                              batch_size=eval_tf_env.batch_size),
    ]

    time_step_spec = tf_env.time_step_spec()
    observation_spec = time_step_spec.observation
    action_spec = tf_env.action_spec()

    actor_net = actor_distribution_network.ActorDistributionNetwork(
        observation_spec, action_spec, fc_layer_params=actor_fc_layers,
        continuous_projection_net=tanh_normal_projection_network.TanhNormalProjectionNetwork)
    critic_net = critic_network.CriticNetwork(
        (observation_spec, action_spec), observation_fc_layer_params=critic_obs_fc_layers,
        action_fc_layer_params=critic_action_fc_layers,
        joint_fc_layer_params=critic_joint_fc_layers, kernel_initializer="glorot_uniform",
        last_kernel_initializer="glorot_uniform")

    tf_agent = sac_agent.SacAgent(
        time_step_spec, action_spec, actor_network=actor_net, critic_network=critic_net,
        actor_optimizer=optimizers.AdamOptimizer(learning_rate=actor_learning_rate),
        critic_optimizer=optimizers.AdamOptimizer(learning_rate=critic_learning_rate),
        alpha_optimizer=optimizers.AdamOptimizer(learning_rate=alpha_learning_rate),
        target_update_tau=target_update_tau, target_update_period=target_update_period,
        td_errors_loss_fn=td_errors_loss_fn, gamma=gamma,
        reward_scale_factor=reward_scale_factor, gradient_clipping=gradient_clipping,
        debug_summaries=debug_summaries, summarize_grads_and_vars=summarize_grads_and_vars,
        train_step_counter=global_step)
    tf_agent.initialize()

    # Make the replay buffer.
    replay_buffer = tf_uniform_replay_buffer.TFUniformReplayBuffer(
        data_spec=tf_agent.collect_data_spec, batch_size=tf_env.batch_size,
        max_length=replay_buffer_capacity)
    replay_observer = [replay_buffer.add_batch]

    train_metrics = [
        tf_metrics.NumberOfEpisodes(),
        tf_metrics.EnvironmentSteps(),
        tf_metrics.AverageReturnMetric(buffer_size=num_eval_episodes,
                                       batch_size=tf_env.batch_size),
        tf_metrics.AverageEpisodeLengthMetric(buffer_size=num_eval_episodes,
                                              batch_size=tf_env.batch_size),
    ]

    eval_policy = greedy_policy.GreedyPolicy(tf_agent.policy)
    initial_collect_policy = random_tf_policy.RandomTFPolicy(tf_env.time_step_spec(),
                                                             tf_env.action_spec())
    collect_policy = tf_agent.collect_policy

    train_checkpointer = common.Checkpointer(
        ckpt_dir=train_dir, agent=tf_agent, global_step=global_step,
        metrics=metric_utils.MetricsGroup(train_metrics, "train_metrics"))
    policy_checkpointer = common.Checkpointer(
        ckpt_dir=os.path.join(train_dir, "policy"), policy=eval_policy, global_step=global_step)
    rb_checkpointer = common.Checkpointer(
        ckpt_dir=os.path.join(train_dir, "replay_buffer"), max_to_keep=1,
        replay_buffer=replay_buffer)
    train_checkpointer.initialize_or_restore()
    rb_checkpointer.initialize_or_restore()

    initial_collect_driver = dynamic_step_driver.DynamicStepDriver(
        tf_env, initial_collect_policy, observers=replay_observer + train_metrics,
        num_steps=initial_collect_steps)
    collect_driver = dynamic_step_driver.DynamicStepDriver(
        tf_env, collect_policy, observers=replay_observer + train_metrics,
        num_steps=collect_steps_per_iteration)

    train = tf_agent.train
    if use_tf_functions:
        initial_collect_driver.run = common.function(initial_collect_driver.run)
        collect_driver.run = common.function(collect_driver.run)
        train = common.function(tf_agent.train)

    if replay_buffer.num_frames() == 0:
        # Collect initial replay data.
        logging.info("Initializing replay buffer by collecting experience for %d steps with a "
                     "random policy.", initial_collect_steps)
        initial_collect_driver.run()

    def evaluate(step):
        results = metric_utils.eager_compute(
            eval_metrics, eval_tf_env, eval_policy, num_episodes=num_eval_episodes,
            train_step=global_step, summary_prefix="Metrics")
        if eval_metrics_callback is not None:
            eval_metrics_callback(results, step)
        metric_utils.log_metrics(eval_metrics)
        return results

    evaluate(global_step.numpy())

    time_step = None
    policy_state = collect_policy.get_initial_state(tf_env.batch_size)
    timed_at_step = global_step.numpy()
    time_acc = 0

    # Prepare replay buffer as dataset with invalid transitions filtered.
    def _filter_invalid_transition(trajectories, unused_arg1):
        return ~trajectories.is_boundary()[0]

    dataset = (replay_buffer.as_dataset(sample_batch_size=batch_size, num_steps=2)
               .unbatch().filter(_filter_invalid_transition).batch(batch_size).prefetch(5))
    # Dataset generates trajectories with shape [Bx2x...]
    iterator = iter(dataset)

    def train_step():
        experience, _ = next(iterator)
        return train(experience)

    if use_tf_functions:
        train_step = common.function(train_step)

    train_loss = None
    global_step_val = global_step.numpy()
    while global_step_val < num_iterations:
        start_time = time.time()
        time_step, policy_state = collect_driver.run(time_step=time_step,
                                                     policy_state=policy_state)
        for _ in range(train_steps_per_iteration):
            train_loss = train_step()
        time_acc += time.time() - start_time

        global_step_val = global_step.numpy()
        if global_step_val % log_interval == 0:
            logging.info("step = %d, loss = %f", global_step_val, float(train_loss.loss))
            steps_per_sec = (global_step_val - timed_at_step) / max(time_acc, 1e-9)
            logging.info("%.3f steps/sec", steps_per_sec)
            timed_at_step = global_step_val
            time_acc = 0
        if global_step_val % eval_interval == 0:
            evaluate(global_step_val)
        if global_step_val % train_checkpoint_interval == 0:
            train_checkpointer.save(global_step=global_step_val)
        if global_step_val % policy_checkpoint_interval == 0:
            policy_checkpointer.save(global_step=global_step_val)
        if global_step_val % rb_checkpoint_interval == 0:
            rb_checkpointer.save(global_step=global_step_val)
    train_eval.last_run = dict(agent=tf_agent, replay_buffer=replay_buffer, actor_net=actor_net,
                               critic_net=critic_net, train_metrics=train_metrics,
                               global_step=global_step, dataset=dataset)
    return train_loss


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--root_dir", required=True)
    ap.add_argument("--num_iterations", type=int, default=3000000)
    ap.add_argument("--gin_file", action="append")
    ap.add_argument("--gin_param", action="append")
    a = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    gin.parse_config_files_and_bindings(a.gin_file, a.gin_param)
    train_eval(a.root_dir, num_iterations=a.num_iterations)


if __name__ == "__main__":
    main()
