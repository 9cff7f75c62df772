#!/usr/bin/env python
"""BASELINE.json configs[4] at one GPU (a parity configuration; measured for the record): SAC on
Humanoid-shaped synthetic data -- obs f32[376], action f32[17], 4,096 parallel envs, replay
4096 x 1000 frames = 4.1 M rows = 6.5 GB (configs[4]'s "replay cap=4M"), batch 256, actor (256,256) with tanh-Normal projection, twin critics (256,256),
three Adam(3e-4), tau 0.005 every step (tf_agents/examples/sac/haarnoja18/sac_train_eval.py:182-199;
SURVEY.md §8d config 5).  One iteration = 1 collect step (4,096 envs) + 1 SacAgent.train.
    python tools/bench_sac.py [--iters 200]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_baseline(batch, steps, threads):
    """oracle/sac.py's OracleSacAgent.train (torch-CPU autograd, three Adam steps, soft target
    update) on the same shapes: learner steps/s."""
    from oracle import nets as onets
    from oracle import sac as osac
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(0)
    O, A, B = 376, 17, batch
    al, cl = onets.mlp_q_layers((256, 256), 2 * A, "relu"), onets.mlp_q_layers((256, 256), 1, "relu")
    ag = osac.OracleSacAgent(O, A, (256, 256), (256, 256), [0.0] * A, [0.4] * A,
                             onets.init_params(al, (O,), seed=1),
                             onets.init_params(cl, (O + A,), seed=2),
                             onets.init_params(cl, (O + A,), seed=3), reward_scale_factor=0.1,
                             std_kind="clip_exp")
    r = lambda *s: torch.randn(*s, generator=g)
    obs, nobs, act = r(B, O), r(B, O), r(B, A).clamp(-0.4, 0.4)
    rew, disc = r(B), torch.ones(B)

    def one():
        ag.train(obs, act, nobs, rew, disc, r(B, A), r(B, A), r(B, A))

    one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    return steps / (time.perf_counter() - t0)


def build(dev, envs=4096, max_length=1000, batch=256, record_noise=False, prefill=True, rank=0):
    """configs[4] at one GPU as this benchmark runs it (also what
    tests/test_gpu_bench_config_sac.py checks against oracle/sac.py).  `rank`: data-parallel
    replica index -- its own environment and replay streams; the networks are seeded alike (and the
    Learner broadcasts rank 0's state anyway)."""
    from agents_amd import optimizers
    from agents_amd.agents.sac import sac_agent
    from agents_amd.drivers import dynamic_step_driver
    from agents_amd.environments import random_tf_environment
    from agents_amd.networks import actor_distribution_network as adn
    from agents_amd.networks import critic_network
    from agents_amd.networks import layers as L
    from agents_amd.replay_buffers import tf_uniform_replay_buffer as rb_lib
    from agents_amd.specs import tensor_spec
    from agents_amd.train import learner
    from agents_amd.trajectories import time_step as ts
    from agents_amd.utils import common

    obs = tensor_spec.BoundedTensorSpec((376,), torch.float32, -1.0, 1.0)
    act = tensor_spec.BoundedTensorSpec((17,), torch.float32, -0.4, 0.4)
    tss = ts.time_step_spec(obs)
    actor = adn.ActorDistributionNetwork(
        obs, act, fc_layer_params=(256, 256),
        continuous_projection_net=lambda spec: adn.TanhNormalProjectionNetwork(
            spec, std_transform=adn.std_clip_transform), seed=1)
    critic = critic_network.CriticNetwork((obs, act), joint_fc_layer_params=(256, 256),
                                          kernel_initializer=L.GlorotUniform(),
                                          last_kernel_initializer=L.GlorotUniform(), seed=2)
    agent = sac_agent.SacAgent(
        tss, act, critic_network=critic, actor_network=actor,
        actor_optimizer=optimizers.Adam(3e-4), critic_optimizer=optimizers.Adam(3e-4),
        alpha_optimizer=optimizers.Adam(3e-4), target_update_tau=0.005, target_update_period=1,
        td_errors_loss_fn=common.element_wise_squared_loss, gamma=0.99, reward_scale_factor=0.1)
    agent.record_noise = record_noise
    agent.initialize()
    env = random_tf_environment.RandomTFEnvironment(tss, act, batch_size=envs,
                                                    episode_end_probability=1e-3,
                                                    seed=3 + 1000 * rank, device=dev)
    rb = rb_lib.TFUniformReplayBuffer(agent.collect_data_spec, batch_size=envs,
                                      max_length=max_length, device=dev, seed=13 * rank)
    init = dynamic_step_driver.DynamicStepDriver(env, agent.collect_policy,
                                                 observers=[rb.add_batch],
                                                 num_steps=envs * max_length)
    if prefill:
        init.run()
    drv = dynamic_step_driver.DynamicStepDriver(env, agent.collect_policy,
                                                observers=[rb.add_batch], num_steps=1)
    collect = common.function(drv.run)
    lrn = learner.Learner(None, common.Variable(0), agent)
    dataset = rb.as_dataset(sample_batch_size=batch, num_steps=2).prefetch(3)
    return dict(agent=agent, actor=actor, critic=critic, env=env, rb=rb, init_driver=init,
                collect_driver=drv, collect=collect, learner=lrn, dataset=dataset,
                obs_spec=obs, action_spec=act)


def run(args, dev=None, rank=0, world=1):
    """`world` > 1: one process per GPU (bench.py --config sac --gpus N): per-rank envs and replay
    shard, the three gradient buffers SUM all-reduced per train step (Learner's strategy); timed
    between barriers, the slowest rank's time counts."""
    from agents_amd.utils import graph

    if dev is None:
        dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    w = build(dev, args.envs, args.max_length, args.batch, rank=rank)
    agent, collect, lrn = w["agent"], w["collect"], w["learner"]
    if not getattr(args, "no_overlap", False):
        # collect / sample / train graphs on three HIP streams, ordered along the true data
        # dependencies (utils/graph.py: Lanes): the critic update runs beside the collect step
        graph.enable_overlap(dev)
    it = iter(w["dataset"])
    tsx = None

    host_delay = float(os.environ.get("AA_BENCH_HOST_DELAY_US", "0")) * 1e-6   # see bench.py

    def step():
        nonlocal tsx
        if host_delay:
            t_ = time.perf_counter() + host_delay
            while time.perf_counter() < t_:
                pass
        tsx, _ = collect(tsx)
        return lrn.run(iterations=1, iterator=it)

    def sync_all():
        graph.join_lanes(dev)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    strategy = lrn.strategy
    # the per-launch work of the wide-MLP forward kernel (the roofline's numerator): the first
    # iteration runs every launch of the loop eagerly, i.e. through the Python wrappers
    from agents_amd.networks import sequential
    sequential.WIDE_FWD_LOG = []
    step()
    fwd_log, sequential.WIDE_FWD_LOG = sequential.WIDE_FWD_LOG, None
    for _ in range(40):
        step()
    sync_all()
    mark = getattr(args, "mark", None) or (lambda label: None)
    mark("loop.begin")
    if hasattr(strategy, "reset_stats"):
        strategy.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        li = step()
    t_host = (time.perf_counter() - t0) / args.iters    # the host is done enqueueing here
    sync_all()
    dt = (time.perf_counter() - t0) / args.iters
    mark("loop.end")
    coll = None
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt[0])
        st = dict(strategy.stats)
        # a second, untimed pass with timing events around the collectives: exposed stream time
        strategy.reset_stats()
        strategy.profile = True
        n_prof = min(args.iters, 50)
        for _ in range(n_prof):
            step()
        sync_all()
        coll = {"allreduce_calls_per_step": st["calls"] / args.iters,
                "allreduce_bytes_per_step": st["bytes"] / args.iters,
                "allreduce_exposed_ms_per_step": strategy.exposed_ms() / n_prof,
                "backend": strategy.backend, "ranks": strategy.num_replicas_in_sync}
        strategy.profile = False
    if getattr(args, "timeline", 0):
        # GPU-side event timeline of the loop (timing events around the graph launches), us after
        # the end of the previous iteration's train step: who waits for whom
        graph.TIMELINE = []
        for _ in range(args.timeline + 10):
            step()
        graph.join_lanes(dev)
        torch.cuda.synchronize()
        marks, graph.TIMELINE = graph.TIMELINE, None
        iters, cur = [], None
        for tag, ev in marks:
            if tag == "collect.begin":
                cur = {}
                iters.append(cur)
            if cur is not None:
                cur[tag] = ev
        rows = []
        for a, b in zip(iters[8:-1], iters[9:]):
            if "train.apply_done" in a and "train.apply_done" in b:
                rows.append({k: a["train.apply_done"].elapsed_time(v) * 1e3 for k, v in b.items()})
        keys = ["collect.begin", "collect.done", "sample.begin", "sample.done", "train.begin",
                "train.part_a_done", "train.part_b_begin", "train.apply_done"]
        print("[bench_sac] GPU timeline, us after the previous train step's end (mean of %d): " %
              len(rows) + ", ".join(f"{k} {sum(r[k] for r in rows if k in r) / max(len(rows), 1):.0f}"
                                    for k in keys), file=sys.stderr, flush=True)
    row = 4 + 376 * 4 + 17 * 4 + 4 + 4 + 4
    return ({
        "workload": "configs[4] at 1 GPU: SAC Humanoid-shaped, %d envs, batch %d, actor/critics "
                    "(256,256), replay %d x %d = %.2f M rows (%.1f GB)" % (
                        args.envs, args.batch, args.envs, args.max_length,
                        args.envs * args.max_length / 1e6,
                        args.envs * args.max_length * row / 1e9),
        "ms_per_iteration": dt * 1e3, "host_enqueue_ms_per_iteration": t_host * 1e3,
        "learner_steps_per_sec": 1.0 / dt,
        "env_steps_per_sec": world * args.envs / dt,
        "trained_transitions_per_sec": world * args.batch / dt,
        "replay_row_bytes": row, "replay_rows": args.envs * args.max_length,
        "replay_bytes": args.envs * args.max_length * row,
        "replay_rows_per_sec": world * args.batch * 2 / dt,      # S x T rows returned by get_next
        "wide_fwd_launches": fwd_log,
        "final_loss": float(li.loss), "n_gpus": world,
        "collectives": coll,
        "train_graph_replays": graph.graphed_train(agent).replays})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--max-length", type=int, default=1000,
                    help="replay frames per env (1000 x 4096 envs = configs[4]'s 4 M-row cap)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--timeline", type=int, default=0,
                    help="print a GPU-side event timeline averaged over this many iterations")
    print(json.dumps(run(ap.parse_args())))


if __name__ == "__main__":
    main()
