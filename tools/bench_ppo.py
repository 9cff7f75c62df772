#!/usr/bin/env python
"""BASELINE.json configs[2] (a parity-test configuration, measured here for the record, not the
bench.py line): PPO on HalfCheetah-shaped synthetic data -- 2,048 parallel envs, 128-step
collection, GAE(lambda=0.95), actor/value MLPs (64, 64) tanh, minibatch 4,096, 10 epochs
(tf_agents/examples/ppo/schulman17/train_eval_lib.py:85-112,200-202).

One iteration = collect 2048 x 129 env steps into the replay table (DynamicStepDriver + collect
policy: actor forward, Normal sample, value forward) -> preprocess_sequence (value bootstrap, GAE,
returns, advantage normalisation) -> PPOLearner: 10 epochs x 64 shuffled minibatches of 4,096 frames,
each one PPOClipAgent.train (actor+value forward, clipped surrogate + value loss, backward, global-
norm clip 0.5, Adam).  Prints one JSON line.   python tools/bench_ppo.py [--iters 3]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_baseline(minibatch, steps, threads):
    """torch-CPU restatement of one PPOClipAgent minibatch step (oracle/ppo.py losses, autograd,
    global-norm clip, oracle Adam) on the same (64, 64) tanh MLPs and shapes: steps/s."""
    import numpy as np
    from oracle import optim as ooptim
    from oracle import ppo as oppo
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    N, D = minibatch, 6

    def mlp(sizes):
        ps = []
        for a, b in zip(sizes[:-1], sizes[1:]):
            ps += [(torch.randn(a, b, generator=g) / a ** 0.5).requires_grad_(True),
                   torch.zeros(b, requires_grad=True)]
        return ps
    actor, value = mlp([17, 64, 64, D]), mlp([17, 64, 64, 1])
    sb = torch.zeros(D, requires_grad=True)
    params = actor + [sb] + value
    opt = ooptim.Adam(3e-4, eps=1e-5)
    obs = torch.randn(N, 17, generator=g)
    acts, old_loc = torch.randn(N, D, generator=g), torch.randn(N, D, generator=g) * 0.1
    old_scale = torch.ones(N, D)
    old_logp = oppo.normal_log_prob(old_loc, old_scale, acts)
    adv, ret = torch.randn(N, generator=g), torch.randn(N, generator=g)
    w = torch.ones(N)

    def fwd(ps, x):
        h = x
        for i in range(0, len(ps) - 2, 2):
            h = torch.tanh(h @ ps[i] + ps[i + 1])
        return h @ ps[-2] + ps[-1]

    def one():
        loc = torch.tanh(fwd(actor, obs))
        scale = torch.nn.functional.softplus(sb).expand_as(loc)
        val = fwd(value, obs)[:, 0]
        out = oppo.losses(loc, scale, acts, old_logp, adv, ret, val, w, clip_eps=0.2, c_v=0.5)
        grads = torch.autograd.grad(out["total"], params)
        gn = torch.sqrt(sum((x ** 2).sum() for x in grads))
        sc = 0.5 * min(1.0 / float(gn), 1.0 / 0.5)
        opt.step(params, [x * sc for x in grads])

    one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    return steps / (time.perf_counter() - t0)


def build(dev, envs=2048, steps=128, minibatch=4096, epochs=10, after_train_step_fn=None,
          episode_end_probability=1e-3, rank=0):
    """configs[2] as this benchmark runs it (also what tests/test_gpu_bench_config_ppo.py checks
    against oracle/ppo.py + oracle/tensor_normalizer.py)."""
    from agents_amd import optimizers
    from agents_amd.agents.ppo import ppo_actor_network as pan
    from agents_amd.agents.ppo import ppo_clip_agent
    from agents_amd.drivers import dynamic_step_driver
    from agents_amd.environments import random_tf_environment
    from agents_amd.replay_buffers import tf_uniform_replay_buffer as rb_lib
    from agents_amd.specs import tensor_spec
    from agents_amd.train import ppo_learner
    from agents_amd.trajectories import time_step as ts
    from agents_amd.utils import common

    B, T = envs, steps
    obs = tensor_spec.BoundedTensorSpec((17,), torch.float32, -1.0, 1.0)
    act = tensor_spec.BoundedTensorSpec((6,), torch.float32, -1.0, 1.0)
    tss = ts.time_step_spec(obs)
    actor = pan.PPOActorNetwork().create_sequential_actor_net((64, 64), act, seed=1)
    value = pan.value_network((64, 64), "tanh", seed=2)
    agent = ppo_clip_agent.PPOClipAgent(
        tss, act, optimizers.Adam(3e-4, epsilon=1e-5), actor_net=actor, value_net=value,
        importance_ratio_clipping=0.2, lambda_value=0.95, discount_factor=0.99, use_gae=True,
        num_epochs=1, gradient_clipping=0.5, normalize_observations=True,
        normalize_rewards=True, compute_value_and_advantage_in_train=False,
        update_normalizers_in_train=False)     # schulman17/train_eval_lib.py:197-226 defaults
    agent.initialize()
    env = random_tf_environment.RandomTFEnvironment(tss, act, batch_size=B,
                                                    episode_end_probability=episode_end_probability,
                                                    seed=3 + 1000 * rank, device=dev)
    rb = rb_lib.TFUniformReplayBuffer(agent.collect_data_spec, batch_size=B, max_length=T + 1,
                                      device=dev)
    drv = dynamic_step_driver.DynamicStepDriver(env, agent.collect_policy,
                                                observers=[rb.add_batch], num_steps=B * (T + 1))

    def raw_dataset_fn():
        return rb.as_dataset(sample_batch_size=B, num_steps=T + 1, single_deterministic_pass=True)

    def dataset_fn():
        return raw_dataset_fn().map(
            lambda traj, info: (agent.preprocess_sequence(traj), info))

    lrn = ppo_learner.PPOLearner(None, common.Variable(0), agent, dataset_fn, dataset_fn,
                                 num_samples=1, num_epochs=epochs,
                                 minibatch_size=minibatch, shuffle_buffer_size=B * (T + 1),
                                 after_train_strategy_step_fn=after_train_step_fn)
    # train_eval_clip_agent.py:248-251: `collect_driver.run = common.function(collect_driver.run)`
    collect = common.function(drv.run)
    return dict(agent=agent, actor=actor, value=value, env=env, rb=rb, collect_driver=drv,
                collect=collect, learner=lrn, raw_dataset_fn=raw_dataset_fn, obs_spec=obs,
                action_spec=act)


def run(args, dev=None, rank=0, world=1):
    """`world` > 1: one process per GPU (bench.py --config ppo --gpus N): every rank collects with
    its own environments and trains on its own frames; the gradient buffer is SUM all-reduced per
    minibatch step and the normalisers see the gathered batch (Learner's strategy)."""
    if dev is None:
        dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B, T = args.envs, args.steps
    w = build(dev, B, T, args.minibatch, args.epochs, rank=rank)
    agent, rb, drv, lrn = w["agent"], w["rb"], w["collect_driver"], w["learner"]
    # The eager driver loop.  AA_BENCH_PPO_GRAPHED_COLLECT=1 replays the loop body as HIP graphs
    # (common.function(driver.run), bit-identical: tests/test_gpu_round5_entries.py) -- measured
    # SLOWER here, same box: 9.85 M vs 12.7 M env steps/s (208 vs 161 us per body).  A PPO body is
    # ~20 dependent launches of a few microseconds each; a HIP graph orders dependent nodes with
    # completion signals (~8-10 us per edge on this runtime), the in-order stream dispatches them
    # back to back, so the graph is GPU-bound above what the eager loop's host can issue.
    collect = w["collect"] if os.environ.get("AA_BENCH_PPO_GRAPHED_COLLECT") == "1" else drv.run
    tsx = [None]

    def sync_all():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def one_iteration():
        sync_all()
        t0 = time.perf_counter()
        rb.clear()
        tsx[0], _ = collect(tsx[0])
        sync_all()
        t1 = time.perf_counter()
        lrn._train_iter = lrn._norm_iter = None   # a fresh deterministic pass over the new data
        li = lrn.run()
        sync_all()
        t2 = time.perf_counter()
        n_steps = (lrn.num_frames_for_training // args.minibatch) * args.epochs
        return t1 - t0, t2 - t1, n_steps, float(li.loss)

    one_iteration()   # warm-up (buffers, workspaces)
    strategy = lrn.strategy
    if hasattr(strategy, "reset_stats"):
        strategy.reset_stats()
    tc = tt = 0.0
    steps = 0
    mark = getattr(args, "mark", None) or (lambda label: None)
    mark("loop.begin")
    for _ in range(args.iters):
        c, t, n, loss = one_iteration()
        tc += c
        tt += t
        steps += n
    mark("loop.end")
    coll = None
    if world > 1:
        import torch.distributed as dist
        v = torch.tensor([tc, tt], dtype=torch.float64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)      # the slowest rank's time counts
        tc, tt = float(v[0]), float(v[1])
        st = dict(strategy.stats)
        strategy.reset_stats()
        strategy.profile = True
        _c, _t, n_prof, _l = one_iteration()
        coll = {"allreduce_calls_per_step": st["calls"] / max(steps, 1),
                "allreduce_bytes_per_step": st["bytes"] / max(steps, 1),
                "allreduce_exposed_ms_per_step": strategy.exposed_ms() / max(n_prof, 1),
                "backend": strategy.backend, "ranks": strategy.num_replicas_in_sync}
        strategy.profile = False
    frames = B * (T + 1)
    out = {"workload": "configs[2]: PPO HalfCheetah-shaped, %d envs x %d steps, minibatch %d, "
                       "%d epochs, MLP (64,64)" % (B, T, args.minibatch, args.epochs),
           "collect_env_steps_per_sec": world * frames * args.iters / tc,
           "collect_s_per_iteration": tc / args.iters,
           "train_minibatch_steps_per_sec": steps / tt,
           "train_frames_per_sec": world * steps * args.minibatch / tt,
           "train_s_per_iteration": tt / args.iters,
           "minibatch_steps_per_iteration": steps // args.iters,
           "iteration_s": (tc + tt) / args.iters, "final_loss": loss, "n_gpus": world,
           "collectives": coll}
    out["frames_per_iteration"] = frames
    out["total_params"] = int(agent.flat_params.numel())
    out["agent"] = agent
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--minibatch", type=int, default=4096)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--iters", type=int, default=3)
    out = run(ap.parse_args())
    out.pop("agent")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
