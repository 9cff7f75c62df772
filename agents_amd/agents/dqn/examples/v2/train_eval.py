r"""Train and eval DQN on the MI355X hot path -- the executable form of "drops into the existing
train_eval scripts": the same `train_eval(...)` entry point, keyword arguments, defaults and loop
structure as tf_agents/agents/dqn/examples/v2/train_eval.py:86-339, with the import root changed
from `tf_agents` to `agents_amd`.

  python -m agents_amd.agents.dqn.examples.v2.train_eval --root_dir=/tmp/dqn --num_iterations=2000

What differs, and why:
  * environments: there is no gym in this image, so `env_name` selects a device-resident
    synthetic environment of the same specs (CartPole: observation f32[4] in [-4, 4], two
    actions -- the shapes of the reference's own benchmark, benchmark/dqn_benchmark_test.py:66-83);
    pass `env_load_fn` to use anything that implements `TFEnvironment`.
  * `common.function(...)` records HIP graphs instead of tracing a tf.function
    (agents_amd/utils/graph.py); `use_tf_functions=False` keeps every launch eager.
  * summaries / TensorBoard writers, gin files and the RNN branch are outside the hot-path scope
    (SURVEY.md §8): `gin.configurable` is a pass-through when gin is not installed, step metrics
    are small device-side counters, evaluation averages return and length over
    `num_eval_episodes` greedy episodes on the host.
"""
import argparse
import functools
import logging
import os
import time

import torch

from agents_amd import optimizers
from agents_amd.agents.dqn import dqn_agent
from agents_amd.drivers import dynamic_step_driver
from agents_amd.environments import random_tf_environment
from agents_amd.networks import layers as keras_layers
from agents_amd.networks import sequential
from agents_amd.policies import q_policy as random_tf_policy
from agents_amd.replay_buffers import tf_uniform_replay_buffer
from agents_amd.specs import tensor_spec
from agents_amd.trajectories import time_step as ts
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


_SYNTHETIC_ENVS = {
    # name -> (observation spec, number of actions)
    "CartPole-v0": (tensor_spec.BoundedTensorSpec((4,), torch.float32, -4.0, 4.0), 2),
    "CartPole-v1": (tensor_spec.BoundedTensorSpec((4,), torch.float32, -4.0, 4.0), 2),
    "Pong-v0": (tensor_spec.TensorSpec((84, 84, 4), torch.uint8), 6),
}


def load_synthetic_env(env_name, batch_size=1, seed=0, episode_end_probability=0.02):
    """Device-resident stand-in for `TFPyEnvironment(suite_gym.load(env_name))`."""
    if env_name not in _SYNTHETIC_ENVS:
        raise ValueError(f"no synthetic environment named {env_name!r}: one of "
                         f"{sorted(_SYNTHETIC_ENVS)} or pass env_load_fn")
    obs_spec, num_actions = _SYNTHETIC_ENVS[env_name]
    action_spec = tensor_spec.BoundedTensorSpec((), torch.int64, 0, num_actions - 1)
    return random_tf_environment.RandomTFEnvironment(
        ts.time_step_spec(obs_spec), action_spec, batch_size=batch_size,
        episode_end_probability=episode_end_probability, seed=seed)


class _StepMetric:
    """tf_metrics.NumberOfEpisodes / EnvironmentSteps: a device counter fed by the driver (one
    small launch per observed trajectory, capturable in the collect graph)."""

    def __init__(self, name, of):
        self.name, self._of, self._count = name, of, None

    def __call__(self, traj):
        hit = self._of(traj)
        if self._count is None:
            self._count = torch.zeros((), dtype=torch.int64, device=hit.device)
        self._count += hit.sum()

    def result(self):
        return 0 if self._count is None else int(self._count.item())


def _eager_compute(env, policy, num_episodes):
    """metric_utils.eager_compute for AverageReturn / AverageEpisodeLength: `num_episodes`
    episodes of `policy` on `env`, averaged on the host."""
    step = env.reset()
    B = env.batch_size
    ret = torch.zeros((B,), dtype=torch.float64, device=step.reward.device)
    length = torch.zeros_like(ret)
    returns, lengths = [], []
    state = policy.get_initial_state(B)
    for _ in range(100000):
        action = policy.action(step, state)
        step = env.step(action.action)
        live = ~step.is_first()
        ret += torch.where(live, step.reward.double(), torch.zeros_like(ret))
        length += live.double()
        done = step.is_last()
        if bool(done.any()):
            returns += ret[done].tolist()
            lengths += length[done].tolist()
            ret[done] = 0
            length[done] = 0
        if len(returns) >= num_episodes:
            break
    n = max(min(len(returns), num_episodes), 1)
    return {"AverageReturn": sum(returns[:n]) / n, "AverageEpisodeLength": sum(lengths[:n]) / n}


@gin.configurable
def train_eval(
        root_dir,
        env_name="CartPole-v0",
        num_iterations=100000,
        train_sequence_length=1,
        # Params for QNetwork
        fc_layer_params=(100,),
        # Params for QRnnNetwork
        input_fc_layer_params=(50,),
        lstm_size=(20,),
        output_fc_layer_params=(20,),
        # Params for collect
        initial_collect_steps=1000,
        collect_steps_per_iteration=1,
        epsilon_greedy=0.1,
        replay_buffer_capacity=100000,
        # Params for target update
        target_update_tau=0.05,
        target_update_period=5,
        # Params for train
        train_steps_per_iteration=1,
        batch_size=64,
        learning_rate=1e-3,
        n_step_update=1,
        gamma=0.99,
        reward_scale_factor=1.0,
        gradient_clipping=None,
        use_tf_functions=True,
        # Params for eval
        num_eval_episodes=10,
        eval_interval=1000,
        # Params for checkpoints
        train_checkpoint_interval=10000,
        policy_checkpoint_interval=5000,
        rb_checkpoint_interval=20000,
        # Params for summaries and logging
        log_interval=1000,
        summary_interval=1000,
        summaries_flush_secs=10,
        debug_summaries=False,
        summarize_grads_and_vars=False,
        eval_metrics_callback=None,
        # Additions (trailing, optional): environment factory and its batch size
        env_load_fn=None,
        num_parallel_environments=1):
    """A simple train and eval for DQN (same contract as the reference's function)."""
    root_dir = os.path.expanduser(root_dir)
    train_dir = os.path.join(root_dir, "train")
    load = env_load_fn or functools.partial(load_synthetic_env,
                                            batch_size=num_parallel_environments)
    tf_env = load(env_name)
    eval_tf_env = load(env_name)
    if train_sequence_length != 1 and n_step_update != 1:
        raise NotImplementedError("train_eval does not currently support n-step updates with "
                                  "stateful networks (i.e., RNNs)")
    if train_sequence_length > 1:
        raise NotImplementedError("recurrent Q-networks are outside the hot-path scope "
                                  "(DESIGN.md section 7)")
    action_spec = tf_env.action_spec()
    num_actions = int(action_spec.maximum) - int(action_spec.minimum) + 1
    q_net = create_feedforward_network(fc_layer_params, num_actions,
                                       tf_env.time_step_spec().observation)
    train_sequence_length = n_step_update
    global_step = common.Variable(0, name="global_step")

    tf_agent = dqn_agent.DqnAgent(
        tf_env.time_step_spec(), action_spec, q_network=q_net, epsilon_greedy=epsilon_greedy,
        n_step_update=n_step_update, target_update_tau=target_update_tau,
        target_update_period=target_update_period,
        optimizer=optimizers.AdamOptimizer(learning_rate=learning_rate),
        td_errors_loss_fn=common.element_wise_squared_loss, gamma=gamma,
        reward_scale_factor=reward_scale_factor, gradient_clipping=gradient_clipping,
        debug_summaries=debug_summaries, summarize_grads_and_vars=summarize_grads_and_vars,
        train_step_counter=global_step)
    tf_agent.initialize()

    train_metrics = [_StepMetric("NumberOfEpisodes", lambda tr: tr.is_last()),
                     _StepMetric("EnvironmentSteps", lambda tr: ~tr.is_boundary())]
    eval_policy = tf_agent.policy
    collect_policy = tf_agent.collect_policy

    replay_buffer = tf_uniform_replay_buffer.TFUniformReplayBuffer(
        data_spec=tf_agent.collect_data_spec, batch_size=tf_env.batch_size,
        max_length=replay_buffer_capacity)
    collect_driver = dynamic_step_driver.DynamicStepDriver(
        tf_env, collect_policy, observers=[replay_buffer.add_batch] + train_metrics,
        num_steps=collect_steps_per_iteration)

    train_checkpointer = common.Checkpointer(ckpt_dir=train_dir, agent=tf_agent,
                                             global_step=global_step)
    policy_checkpointer = common.Checkpointer(ckpt_dir=os.path.join(train_dir, "policy"),
                                              policy=eval_policy, global_step=global_step)
    rb_checkpointer = common.Checkpointer(ckpt_dir=os.path.join(train_dir, "replay_buffer"),
                                          max_to_keep=1, replay_buffer=replay_buffer)
    train_checkpointer.initialize_or_restore()
    rb_checkpointer.initialize_or_restore()

    train = tf_agent.train
    if use_tf_functions:
        # HIP-graph capture of the collect loop body and of the train step
        collect_driver.run = common.function(collect_driver.run)
        train = common.function(tf_agent.train)

    initial_collect_policy = random_tf_policy.RandomTFPolicy(tf_env.time_step_spec(),
                                                             action_spec)
    logging.info("Initializing replay buffer by collecting experience for %d steps with a random "
                 "policy.", initial_collect_steps)
    dynamic_step_driver.DynamicStepDriver(
        tf_env, initial_collect_policy, observers=[replay_buffer.add_batch] + train_metrics,
        num_steps=initial_collect_steps).run()

    def evaluate():
        results = _eager_compute(eval_tf_env, eval_policy, num_eval_episodes)
        if eval_metrics_callback is not None:
            eval_metrics_callback(results, global_step.numpy())
        logging.info("step = %d: %s", global_step.numpy(),
                     ", ".join(f"{k} = {v:.3f}" for k, v in results.items()))
        return results

    evaluate()
    time_step = None
    policy_state = collect_policy.get_initial_state(tf_env.batch_size)
    timed_at_step = global_step.numpy()
    time_acc = 0.0

    # Dataset generates trajectories with shape [B x 2 x ...]
    dataset = replay_buffer.as_dataset(num_parallel_calls=3, sample_batch_size=batch_size,
                                       num_steps=train_sequence_length + 1).prefetch(3)
    iterator = iter(dataset)

    def train_step():
        experience, _ = next(iterator)
        return train(experience)

    train_loss = None
    for _ in range(num_iterations):
        start_time = time.time()
        time_step, policy_state = collect_driver.run(time_step=time_step,
                                                     policy_state=policy_state)
        for _ in range(train_steps_per_iteration):
            train_loss = train_step()
        time_acc += time.time() - start_time
        step = global_step.numpy()
        if step % log_interval == 0:
            logging.info("step = %d, loss = %f", step, float(train_loss.loss))
            steps_per_sec = (step - timed_at_step) / max(time_acc, 1e-9)
            logging.info("%.3f steps/sec", steps_per_sec)
            timed_at_step, time_acc = step, 0.0
        if step % train_checkpoint_interval == 0:
            train_checkpointer.save(global_step=step)
        if step % policy_checkpoint_interval == 0:
            policy_checkpointer.save(global_step=step)
        if step % rb_checkpoint_interval == 0:
            rb_checkpointer.save(global_step=step)
        if step % eval_interval == 0:
            evaluate()
    train_eval.last_run = dict(agent=tf_agent, replay_buffer=replay_buffer, q_net=q_net,
                               train_metrics=train_metrics, global_step=global_step)
    return train_loss


def create_feedforward_network(fc_layer_units, num_actions, input_spec=None):
    """dense(relu, VarianceScaling(2, fan_in, truncated_normal)) x n + logits(U(-0.03, 0.03),
    bias -0.2): the reference's `create_feedforward_network` (train_eval.py:342-366)."""
    dense = functools.partial(keras_layers.Dense, activation="relu",
                              kernel_initializer=keras_layers.VarianceScaling(
                                  2.0, "fan_in", "truncated_normal"))
    logits = functools.partial(keras_layers.Dense, activation=None,
                               kernel_initializer=keras_layers.RandomUniform(-0.03, 0.03),
                               bias_initializer=keras_layers.Constant(-0.2))
    return sequential.Sequential([dense(n) for n in fc_layer_units] + [logits(num_actions)],
                                 input_spec=input_spec)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--root_dir", required=True)
    ap.add_argument("--num_iterations", type=int, default=100000)
    ap.add_argument("--gin_file", action="append")
    ap.add_argument("--gin_param", action="append")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    gin.parse_config_files_and_bindings(args.gin_file, args.gin_param)
    train_eval(args.root_dir, num_iterations=args.num_iterations)


if __name__ == "__main__":
    main()
