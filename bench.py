#!/usr/bin/env python
"""Benchmark of the trainer hot path on MI355X: DQN Atari (BASELINE.json configs[1] / [3]).

  python bench.py --gpus 1 --steps 50 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one iteration of the reference's train_eval hot loop
(tf_agents/agents/dqn/examples/v2/train_eval.py:290-297) on synthetic 84x84x4 uint8 frames:
  collect_driver.run()   256 envs x 1 step: Q-network forward + epsilon-greedy + env step +
                         TFUniformReplayBuffer.add_batch
  learner.run(1)         get_next(256, num_steps=2) from the per-GPU replay shard + DqnAgent.train
                         (online fwd, target fwd, TD/Huber, backward, [RCCL all-reduce], centred
                         RMSProp, periodic target copy)
Weak scaling: every rank owns 256 envs, a 256 x L replay shard and samples 256 transitions; the
global batch is 256 x N and the only exchange is one all-reduce of the 6.75 MB gradient buffer.
`value` = transitions trained on per second, whole job (= learner steps/s x 256 x N).
Prints ONE JSON line on rank 0 (plus a human-readable breakdown on stderr).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA dense peak
PRIME_MIN = 0                  # minimum untimed iterations in front of --warmup (see main: priming)
CPU_RING_FRAMES = 256          # replay frames per env of the cpu_baseline leg (see cpu_baseline)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
PMC_FILE = os.path.join("profiles", "r06_pmc.json")   # committed rocprofv3 --pmc passes (isolated)
PMC_OTHER_FILE = os.path.join("profiles", "r06_pmc_other.json")   # ... of the PPO / SAC kernels
L2_PEAK_GBS = 34500.0          # MI355X_MICROARCH.md: L2 aggregate (8 XCDs x 4 MiB), ~34.5 TB/s
# Set in the child of the in-loop profiling pass (see inloop_profile): the run brackets its timed
# region and every isolated kernel case with aa_marker launches and reports their labels in order.
TRACE_CHILD = os.environ.get("AA_BENCH_TRACE_CHILD") == "1"
MARKS = []


def mark(label):
    """Child of the profiling pass only: one aa_marker dispatch on the current stream, fenced by
    device synchronisation on both sides so that its position in the trace is unambiguous."""
    if not TRACE_CHILD:
        return
    from agents_amd import _lib
    torch.cuda.synchronize()
    _lib.check(_lib.load().aa_marker(len(MARKS), _lib.stream_ptr()), "aa_marker")
    torch.cuda.synchronize()
    MARKS.append(label)
OBS_SHAPE = (84, 84, 4)
NUM_ACTIONS = 6
ROW_BYTES = 4 + 28224 + 8 + 4 + 4 + 4   # Trajectory row: 28,248 B


def log(*a):
    print(*a, file=sys.stderr, flush=True)


LINE_LIMIT = 4096   # the driver keeps a 10 KB tail of stdout: the JSON line must fit with room


def _r(x, nd=4):
    """Round floats for the compact line (6 significant digits for large values)."""
    if isinstance(x, float):
        return float(f"{x:.6g}")
    return x


def _compact_roofline(r):
    """ONE roofline entry for the final line: the contract's keys + the in-loop duration and the
    few qualifiers a reader needs; the long notes stay in the detail file."""
    if not r:
        return None
    keep = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms",
            "isolated_ms", "launches_per_step", "frac_isolated", "pipe_frac", "frac_ceiling",
            "mfma_busy", "algorithmic_flop_per_launch", "algorithmic_bytes_per_launch")
    o = {k: _r(r[k]) for k in keep if k in r and r[k] is not None or k == "traffic"}
    o["kernel"] = str(r.get("kernel", ""))[:120]
    src = r.get("duration_source", "")
    o["duration_source"] = "in-loop rocprofv3" if src.startswith("in-loop") else (
        "isolated HIP events" if src else "wall clock")
    if r.get("traffic") is not None:
        o["traffic_source"] = "committed --pmc pass (" + str(r.get("traffic_source", "")
                                                              ).split(" ")[0] + ")"
    return o


def _compact_other(o):
    if not o or "error" in o:
        return {"error": str((o or {}).get("error", "missing"))[:160]}
    c = {"value": _r(o.get("value")), "unit": o.get("unit"), "ms_per_step": _r(o.get("ms_per_step")),
         "steps": o.get("steps")}
    for k in ("train_minibatch_steps_per_sec", "collect_env_steps_per_sec", "env_steps_per_sec",
              "replay_rows_per_sec"):
        if k in o:
            c[k] = _r(o[k])
    rows = (o.get("config") or {}).get("replay_rows_per_gpu")
    if rows:
        c["config"] = {"replay_rows_per_gpu": rows,
                       "replay_bytes_per_gpu": (o.get("config") or {}).get("replay_bytes_per_gpu")}
    rf = o.get("roofline") or {}
    c["roofline"] = {k: _r(rf.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac",
                                                "traffic", "avg_launch_ms", "launches_per_step",
                                                "algorithmic_bytes_per_launch",
                                                "algorithmic_flop_per_launch", "duration_source")}
    c["roofline"]["kernel"] = str(rf.get("kernel", ""))[:40]
    cb = o.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {k: _r(cb.get(k)) for k in ("value", "unit", "cores", "kind")}
    return c


def compact_line(out):
    """The ONE stdout line the driver parses: contract keys, one roofline entry (dominant kernel by
    time per step) plus the two replay kernels in brief, cpu_baseline, compact other_configs.
    Everything else (`roofline_all`, `inloop`, notes) goes to --detail-out and stderr.  Guaranteed
    shorter than LINE_LIMIT bytes: optional blocks are dropped, least important first."""
    # (the contract's numbers keep full precision; derived figures are rounded to 6 digits)
    line = {k: out[k] for k in (
        "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
        "scaling", "vs_baseline", "dtype", "data") if k in out}
    cfg = dict(out.get("config", {}))
    cfg["workload"] = str(cfg.get("workload", ""))[:300]
    line["config"] = cfg
    if "learner_steps_per_sec" in out:
        line["learner_steps_per_sec"] = out["learner_steps_per_sec"]
    for k in ("replay_rows_per_sec", "env_steps_per_sec", "host_enqueue_ms_per_step", "host_wait_ms_per_step",
              "host_work_ms_per_step", "prime_steps", "captures_in_timed_region", "final_loss", "step_algorithmic_gflop",
              "step_mfma_frac", "kernel_time_sum_ms", "steady_ms_per_step", "steady_steps",
              "rccl_ranks", "dominant_device_kernel", "lanes", "lane_probe"):
        if k in out:
            line[k] = _r(out[k])
    if out.get("collectives"):
        line["collectives"] = {k: _r(v) for k, v in out["collectives"].items()}
    if "roofline" in out:
        line["roofline"] = _compact_roofline(out["roofline"])
    for k in ("roofline_replay_gather", "roofline_replay_add"):
        if k in out:
            r = out[k]
            line[k] = {kk: _r(r.get(kk)) for kk in ("achieved", "unit", "frac", "frac_isolated",
                                                    "avg_launch_ms", "isolated_ms", "traffic")}
    il = out.get("inloop")
    if il and "error" not in il:
        line["inloop"] = {k: _r(il[k]) for k in ("steps", "ms_per_step_under_profiler",
                                                  "kernel_time_sum_us_per_step",
                                                  "launches_per_step", "gpu_wall_us_per_step")
                          if k in il}
    if "cpu_baseline" in out:
        cb = dict(out["cpu_baseline"])
        cb["sample"] = str(cb.get("sample", ""))[:240]
        line["cpu_baseline"] = {k: _r(v) for k, v in cb.items()}
    if "other_configs" in out:
        line["other_configs"] = {k: _compact_other(v) for k, v in out["other_configs"].items()}
    if "detail" in out:
        line["detail"] = out["detail"]
    s = json.dumps(line, separators=(",", ":"))
    for drop in ("inloop", "roofline_replay_add", "roofline_replay_gather", "other_configs",
                 "detail"):
        if len(s) < LINE_LIMIT:
            break
        line.pop(drop, None)
        s = json.dumps(line, separators=(",", ":"))
    assert len(s) < LINE_LIMIT, len(s)
    return s


def emit(out, detail_path):
    """Full record -> detail file (+ stderr pointer); compact line -> stdout (the LAST line)."""
    if detail_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as fh:
                json.dump(out, fh, indent=1)
            out = dict(out, detail=os.path.relpath(os.path.abspath(detail_path), ROOT)
                       if os.path.abspath(detail_path).startswith(ROOT) else detail_path)
            log(f"[bench] full record (roofline_all, inloop kernel table, notes): {detail_path}")
        except OSError as e:
            log(f"[bench] could not write {detail_path}: {e}")
    print(compact_line(out), flush=True)


def atari_layers(L, num_actions):
    vs = lambda: L.VarianceScaling(2.0)
    return [L.Rescale(255.0), L.Conv2D(32, (8, 8), 4, "relu", kernel_initializer=vs()),
            L.Conv2D(64, (4, 4), 2, "relu", kernel_initializer=vs()),
            L.Conv2D(64, (3, 3), 1, "relu", kernel_initializer=vs()), L.Flatten(),
            L.Dense(512, "relu", kernel_initializer=vs()),
            L.Dense(num_actions, None, kernel_initializer=vs())]


def fwd_macs_per_sample():
    return (20 * 20 * 32 * 256, 9 * 9 * 64 * 512, 7 * 7 * 64 * 576, 3136 * 512,
            512 * NUM_ACTIONS)


def train_flops_per_sample():
    """online fwd + target fwd + online bwd (weight grads for all layers, input grads for all but
    the first): SURVEY.md §8d restated without the conv1 input gradient nobody needs."""
    m = fwd_macs_per_sample()
    fwd = sum(m)
    bwd = sum(m) + sum(m[1:])
    return 2.0 * (2 * fwd + bwd)


def build_workload(dev, rank, world, B_env, max_length, S, seed, replay="uniform"):
    from agents_amd import optimizers
    from agents_amd.agents.dqn import dqn_agent
    from agents_amd.drivers import dynamic_step_driver
    from agents_amd.environments import random_tf_environment
    from agents_amd.networks import layers as L
    from agents_amd.networks import sequential
    from agents_amd.policies import q_policy
    from agents_amd.replay_buffers import tf_uniform_replay_buffer as rb_lib
    from agents_amd.specs import tensor_spec
    from agents_amd.train import learner
    from agents_amd.trajectories import time_step as ts
    from agents_amd.utils import common

    obs_spec = tensor_spec.TensorSpec(OBS_SHAPE, torch.uint8, "observation")
    aspec = tensor_spec.BoundedTensorSpec((), torch.int64, 0, NUM_ACTIONS - 1, "action")
    tss = ts.time_step_spec(obs_spec)
    env = random_tf_environment.RandomTFEnvironment(tss, aspec, batch_size=B_env,
                                                    episode_end_probability=1e-3,
                                                    seed=seed * 1000 + rank, device=dev)
    net = sequential.Sequential(atari_layers(L, NUM_ACTIONS), seed=2)  # same init on every rank
    train_step = common.Variable(0, name="train_step")
    agent = dqn_agent.DqnAgent(
        tss, aspec, q_network=net,
        optimizer=optimizers.RMSprop(2.5e-4, rho=0.95, momentum=0.95, epsilon=0.01, centered=True),
        td_errors_loss_fn=common.element_wise_huber_loss, gamma=0.99, epsilon_greedy=0.1,
        target_update_tau=1.0, target_update_period=2500, train_step_counter=train_step,
        seed=seed * 77 + rank)
    hook = None
    if replay == "prioritized":
        # proportional prioritized sampling (csrc/prio.hip) closed into a loop through the hooks
        # the reference provides: DqnLossInfo.td_error -> Learner(after_train_strategy_step_fn)
        # (tf_agents/train/learner.py:362-376, agents/dqn/dqn_agent.py:50-72)
        from agents_amd.replay_buffers import tf_prioritized_replay_buffer as prb
        rb = prb.TFPrioritizedReplayBuffer(agent.collect_data_spec, batch_size=B_env,
                                           max_length=max_length, device=dev,
                                           seed=seed * 13 + rank)
        hook = rb.update_priorities_from_loss
    else:
        rb = rb_lib.TFUniformReplayBuffer(agent.collect_data_spec, batch_size=B_env,
                                          max_length=max_length, device=dev,
                                          seed=seed * 13 + rank)
    random_policy = q_policy.RandomTFPolicy(tss, aspec, seed=seed * 5 + rank)
    init_driver = dynamic_step_driver.DynamicStepDriver(env, random_policy,
                                                        observers=[rb.add_batch],
                                                        num_steps=B_env * max_length)
    collect_driver = dynamic_step_driver.DynamicStepDriver(env, agent.collect_policy,
                                                           observers=[rb.add_batch], num_steps=1)
    dataset = rb.as_dataset(num_parallel_calls=3, sample_batch_size=S, num_steps=2)
    if replay != "prioritized":     # priorities change between draws: no sampling ahead
        dataset = dataset.prefetch(3)
    lrn = learner.Learner(None, train_step, agent, experience_dataset_fn=None,
                          after_train_strategy_step_fn=hook)
    return dict(env=env, agent=agent, rb=rb, init_driver=init_driver,
                collect_driver=collect_driver, dataset=dataset, learner=lrn, net=net)


def args_envs(w):
    return w["env"].batch_size


def kernel_breakdown(w, S, reps=20):
    """Times the kernels of one iteration individually: each op is captured `reps` times into a
    HIP graph on torch's current stream and the replay is bracketed by HIP events on that stream,
    so the figure is GPU time per launch (no Python launch overhead in it).
    Returns [(name, ms per launch, launches per iteration, flops per launch, bytes per launch,
    bf16 products per fp32 product)]: the last is 6 for the bf16x6 kernels (every fp32 operand =
    three exact bf16 pieces, six of the nine piece products accumulated), 3 for the uint8 layer
    (a byte is exact in bf16, only the filter is split), None for fp32-MFMA / non-matrix kernels."""
    from agents_amd import ops
    net = w["net"]
    agent = w["agent"]
    rb = w["rb"]
    exp, _ = rb.get_next(S, 2)
    # rank-local measurement: no collective may be issued here (the other ranks are not in it)
    agent.gradient_hook = None
    agent.gradient_hook_async = None
    agent.num_replicas = 1
    agent.train(exp)  # make sure every buffer exists
    torch.cuda.synchronize()
    slot = net._slots[("train", S)]
    kv, bv, gk, gb = net._kviews, net._bviews, net._gkviews, net._gbviews
    m = fwd_macs_per_sample()
    obs_t = exp.observation[:, 0]

    from agents_amd.utils import graph

    def timeit(fn):
        mark("op.begin")           # (profiling-pass child: the case's launches sit between marks)
        fn()
        c = graph._Captured()      # host mirrors (replay last_id, call counters) follow replays
        c.capture(lambda: [fn() for _ in range(reps)] and None)
        c.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        c.replay()
        b.record()
        torch.cuda.synchronize()
        mark("op.end")
        return a.elapsed_time(b) / reps

    dz4 = torch.randn(S, 512, device=obs_t.device)
    dz3 = torch.randn(S * 49, 64, device=obs_t.device)
    dz2 = torch.randn(S * 81, 64, device=obs_t.device)
    dz1 = torch.randn(S * 400, 32, device=obs_t.device)
    out = []
    f = lambda macs: 2.0 * macs * S
    # replay: rows of 28,248 B; sample = read + write of 512 rows (+ids), add = 256 rows
    items = w["env"].current_time_step()
    from agents_amd.trajectories import trajectory
    from agents_amd.trajectories import policy_step
    act = torch.zeros((S,), dtype=torch.int64, device=obs_t.device)
    traj = trajectory.from_transition(items, policy_step.PolicyStep(act, (), ()), items)
    # what the loop's dataset runs: the stamped launch (last_id and the call number by value, the
    # draw made by the host library: csrc/replay.hip aa_rb_gather_drawn_kernel).  Captured for the
    # timing only -- the rows freeze, the bytes moved are the same.
    if rb.supports_stamped_draws() and getattr(rb, "_dataset_ring", 1):
        elem = rb.get_next(S, 2)
        stamped = rb.stamped_slot(elem)
        out.append(("replay.get_next(sample+gather 512 rows)",
                    timeit(lambda: rb.draw_into(stamped)), 1, 0.0, 512 * (2.0 * ROW_BYTES + 24)))
    else:
        out.append(("replay.get_next(sample+gather 512 rows)", timeit(lambda: rb.get_next(S, 2)),
                    1, 0.0, 512 * (2.0 * ROW_BYTES + 24)))
    out.append(("replay.add_batch(scatter 256 rows)", timeit(lambda: rb.add_batch(traj)), 1, 0.0,
                256 * (2.0 * ROW_BYTES + 8)))
    from agents_amd.networks import sequential as _seq
    out.append(("conv1.fwd(u8)", timeit(lambda: ops.conv_forward(
        obs_t, kv[0], bv[0], 4, "relu", slot.ys[0], a_div=255.0)), 3, f(m[0]), 0))
    if _seq.FUSE_CONV_PAIRS and ops.conv_pair_supported(tuple(slot.ys[0].shape), kv[1], 2, kv[2], 1):
        # what the network runs: both layers in one launch (csrc/conv_pair.hip)
        out.append(("conv2+conv3.fwd(fused)", timeit(lambda: ops.conv_pair_forward(
            slot.ys[0], kv[1], bv[1], 2, "relu", slot.ys[1], kv[2], bv[2], 1, "relu",
            slot.ys[2])), 3, f(m[1] + m[2]), 0))
    else:
        out.append(("conv2.fwd", timeit(lambda: ops.conv_forward(
            slot.ys[0], kv[1], bv[1], 2, "relu", slot.ys[1])), 3, f(m[1]), 0))
        out.append(("conv3.fwd", timeit(lambda: ops.conv_forward(
            slot.ys[1], kv[2], bv[2], 1, "relu", slot.ys[2])), 3, f(m[2]), 0))
    x3 = slot.ys[2].view(S, -1)
    if ops.dense_tail_supported(x3, kv[3], kv[4]):
        # what the network runs: the fc1 main loop, then the head summing fc1's split-K slabs
        out.append(("fc1+fc2.fwd(head sums fc1's slabs)", timeit(lambda: ops.dense_tail_forward(
            x3, kv[3], bv[3], "relu", slot.ys[3], kv[4], bv[4], None, slot.ys[4])), 3,
            f(m[3] + m[4]), 0))
    else:
        out.append(("fc1.fwd", timeit(lambda: ops.dense_forward(
            x3, kv[3], bv[3], "relu", slot.ys[3])), 3, f(m[3]), 0))
        out.append(("fc2.fwd", timeit(lambda: ops.dense_forward(
            slot.ys[3], kv[4], bv[4], None, slot.ys[4])), 3, f(m[4]), 0))
    out.append(("fc1.dW(+bias grad)", timeit(lambda: ops.dense_dw(
        x3, dz4, gk[3], bias_grad=gb[3])), 1, f(m[3]), 0))
    out.append(("fc1.dX", timeit(lambda: ops.dense_dx(
        dz4, kv[3], slot.dxs[3].view(S, -1), mask_src=x3, mask_act="relu")), 1, f(m[3]), 0))
    out.append(("conv3.dW(+bias grad)", timeit(lambda: ops.conv_dw(
        slot.ys[1], dz3, tuple(kv[2].shape), 1, gk[2], bias_grad=gb[2])), 1, f(m[2]), 0))
    out.append(("conv3.dX", timeit(lambda: ops.conv_dx(
        dz3, kv[2], tuple(slot.ys[1].shape), 1, slot.dcol, slot.dxs[2], mask_src=slot.ys[1],
        mask_act="relu")), 1, f(m[2]), 0))
    out.append(("conv2.dW(+bias grad)", timeit(lambda: ops.conv_dw(
        slot.ys[0], dz2, tuple(kv[1].shape), 2, gk[1], bias_grad=gb[1])), 1, f(m[1]), 0))
    out.append(("conv2.dX", timeit(lambda: ops.conv_dx(
        dz2, kv[1], tuple(slot.ys[0].shape), 2, slot.dcol, slot.dxs[1], mask_src=slot.ys[0],
        mask_act="relu")), 1, f(m[1]), 0))
    out.append(("conv1.dW(u8,+bias grad)", timeit(lambda: ops.conv_dw(
        obs_t, dz1, tuple(kv[0].shape), 4, gk[0], a_div=255.0, bias_grad=gb[0])), 1, f(m[0]), 0))
    # ---- the small launches of the iteration (heads, loss, rollout tail) -----------------------
    A = NUM_ACTIONS
    dq = torch.randn(S, A, device=obs_t.device)
    dx4 = slot.dxs[4].view(S, -1) if slot.dxs[4] is not None else dz4
    wk = agent._get_work(S, obs_t.device)
    q_on, q_tg = slot.ys[4], torch.randn(S, A, device=obs_t.device)

    def td_loss(head=None):
        ops.dqn_td_loss(
            q_on, q_tg, None, None, exp.action, exp.reward.contiguous(), exp.discount.contiguous(),
            exp.step_type.contiguous(), None, 0.99, 1.0, agent._loss_kind(agent._td_errors_loss_fn),
            float(S), wk.loss, wk.td_loss, wk.td_error, wk.dq, gamma_loss=0.99,
            field_sums_out=wk.field_sums, head=head)

    head = net.fusable_head(S, "train")
    if head is not None:
        # what the agent runs: TD loss and the Q head's backward pass in ONE launch
        out.append(("td_loss + fc2.dX+dW (one launch)", timeit(lambda: td_loss(head)), 1,
                    2 * f(m[4]), 0))
    else:
        if _seq.FUSE_HEAD_BACKWARD and ops.dense_small_backward_ok(slot.ys[3], dq, slot.ys[3]):
            out.append(("fc2.dX+dW(+bias grad)", timeit(lambda: ops.dense_small_backward(
                slot.ys[3], dq, kv[4], dx4, gk[4], mask_src=slot.ys[3], mask_act="relu",
                bias_grad=gb[4])), 1, 2 * f(m[4]), 0))
        else:
            out.append(("fc2.dW(+bias grad)", timeit(lambda: ops.dense_dw(
                slot.ys[3], dq, gk[4], bias_grad=gb[4])), 1, f(m[4]), 0))
            out.append(("fc2.dX", timeit(lambda: ops.dense_dx(
                dq, kv[4], dx4, mask_src=slot.ys[3], mask_act="relu")), 1, f(m[4]), 0))
        out.append(("dqn.td_loss(+dL/dq, field sums)", timeit(td_loss), 1, 0.0, 100.0 * S))
    pol = agent.collect_policy
    qpol = getattr(pol, "_wrapped_policy", pol)
    qpol = getattr(qpol, "_wrapped_policy", qpol)
    sel = getattr(qpol, "select", None)
    if sel is not None:
        act_out = torch.empty((args_envs(w),), dtype=torch.int64, device=obs_t.device)
        out.append(("policy.eps_greedy_select", timeit(lambda: sel(q_on[:args_envs(w)], None, 0.1,
                                                                   out=act_out)),
                    1, 0.0, args_envs(w) * (4.0 * A + 8)))
    env = w["env"]
    act_env = torch.zeros((env.batch_size,), dtype=torch.int64, device=obs_t.device)
    out.append(("env.step(256 envs, synthetic)", timeit(lambda: env.step(act_env)), 1, 0.0,
                env.batch_size * (28224.0 + 12)))
    from agents_amd import _lib as _l
    lib = _l.load()
    cnt = torch.zeros((env.batch_size,), dtype=torch.int32, device=obs_t.device)
    tot_c = torch.zeros((1,), dtype=torch.int64, device=obs_t.device)
    st_ = env.current_time_step().step_type

    def count_steps():
        with torch.cuda.device(obs_t.device):
            _l.check(lib.aa_count_steps(st_.data_ptr(), st_.numel(), cnt.data_ptr(),
                                        tot_c.data_ptr(), None, _l.stream_ptr()), "aa_count_steps")
    out.append(("driver.count_steps", timeit(count_steps), 1, 0.0, env.batch_size * 8.0))
    n_par = net.flat_params.numel()
    opt = agent._optimizer
    scratch = net.flat_params.clone()
    iters = opt.iterations
    out.append(("rmsprop(centered,mom)", timeit(lambda: opt.apply_flat(scratch, net.flat_grads)),
                1, 0.0, 36.0 * n_par))
    opt.iterations = iters

    def products(name):
        if name.startswith("conv1."):
            return 3
        if name == "conv2+conv3.fwd(fused)":
            return 6 if ops.conv_pair_prepare_bytes(tuple(slot.ys[0].shape), kv[1], 2, kv[2],
                                                    1) > 0 else None
        if name in ("conv2.dX", "conv3.dX"):
            i = 1 if name[4] == "2" else 2
            return 6 if ops.conv_dx_prepare_bytes(tuple(slot.ys[i - 1].shape), kv[i],
                                                  2 if i == 1 else 1) > 0 else None
        if name.startswith(("conv2.dW", "conv3.dW")):
            return 6 if ops.CONV_DW_X6 else None
        return None
    return [r + (products(r[0]),) for r in out]


# ---- in-loop kernel durations: a rocprofv3 --kernel-trace pass of this very loop ------------------
def _read_kernel_trace(out_dir):
    """[(kernel name, start ns, end ns)] sorted by start, from whatever rocprofv3 wrote under
    out_dir: a rocpd sqlite database (ROCm 7 default) or a *_kernel_trace.csv."""
    import csv
    import glob
    import sqlite3
    rows = []
    for db in glob.glob(os.path.join(out_dir, "**", "*.db"), recursive=True):
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        if not cols:
            continue
        name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
        rows += c.execute(f"select {name_col}, start, end from kernels").fetchall()
        c.close()
    if not rows:
        for path in glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True):
            with open(path) as fh:
                for r in csv.DictReader(fh):
                    rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]),
                                 int(r["End_Timestamp"])))
    return sorted(rows, key=lambda r: r[1])


def inloop_profile(args, trace_out=None, timeout=420):
    """Re-runs this benchmark's loop under `rocprofv3 --kernel-trace` in a child process and
    returns per-kernel durations INSIDE the loop (three streams, every kernel competing with its
    neighbours) -- what an isolated 20-launch graph flatters by 1.2-1.7x.  The child brackets its
    timed region, and each isolated case of kernel_breakdown, with aa_marker dispatches; this reads
    the trace by position between them.  Returns a dict, or {"error": ...}."""
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="aa_inloop_", dir="/tmp")
    steps = max(args.inloop_steps, 10)
    cmd = [prof, "--kernel-trace", "-d", tmp, "-o", "inloop", "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", str(steps), "--warmup", "10",
           "--max-length", str(args.max_length), "--batch", str(args.batch), "--envs",
           str(args.envs), "--prefill", str(args.prefill), "--no-cpu-baseline", "--no-inloop",
           "--no-other-configs", "--lanes", getattr(args, "lanes_decided", "on")]
    env = dict(os.environ, AA_BENCH_TRACE_CHILD="1", TMPDIR="/tmp")
    try:
        r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True,
                           timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": f"profiling pass exceeded {timeout}s"}
    child = None
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith("{") and "trace_child" in line:
            child = json.loads(line)
            break
    if r.returncode != 0 or child is None:
        return {"error": f"profiling pass failed (rc {r.returncode}): {r.stderr[-400:]}"}
    try:
        trace = _read_kernel_trace(tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    marks = [i for i, k in enumerate(trace) if "aa_marker_kernel" in k[0]]
    labels = child["marks"]
    if len(marks) != len(labels) or not trace:
        return {"error": f"{len(marks)} marker dispatches in the trace for {len(labels)} marks"}

    def region(a, b):
        """per kernel name: [launches, total ns] between marker dispatches a and b."""
        acc = {}
        for name, st, en in trace[marks[a] + 1:marks[b]]:
            e = acc.setdefault(name, [0, 0])
            e[0] += 1
            e[1] += en - st
        return acc

    i0, i1 = labels.index("loop.begin"), labels.index("loop.end")
    loop = region(i0, i1)
    n_steps = child["steps"]
    kernels = {n: {"launches_per_step": c / n_steps, "avg_us": t / c / 1e3}
               for n, (c, t) in loop.items()}
    ops_ = {}
    op_marks = [i for i, l in enumerate(labels) if l == "op.begin"]
    execs = 1 + 2 * child["reps"]            # one eager call + two replays of the `reps`-graph
    for name, ib in zip(child["ops"], op_marks):
        reg = region(ib, ib + 1)
        # (launches that are not part of the case -- a fill from an event / tensor constructor of
        # the harness -- show up a couple of times per region, not once per execution)
        per_launch = {n: c / execs for n, (c, t) in reg.items() if c / execs >= 0.5}
        in_loop = sum(k * kernels[n]["avg_us"] for n, k in per_launch.items() if n in kernels)
        # launches of the isolated case that the loop does not run (a filter pre-pass the loop
        # replaced by optimizer-maintained planes): left out of the in-loop figure and listed
        missing = [n for n in per_launch if n not in kernels]
        alone = sum(t for n, (c, t) in reg.items() if n in per_launch) / execs / 1e3
        ops_[name] = {"device_kernels": {n: round(k, 3) for n, k in per_launch.items()},
                      "inloop_us": in_loop if len(missing) < len(per_launch) else None,
                      "profiled_alone_us": alone, "not_in_loop": missing}
    wall = (trace[marks[i1]][1] - trace[marks[i0]][2]) / n_steps / 1e3
    out = {"steps": n_steps, "ms_per_step_under_profiler": child["ms_per_step"],
           "kernel_time_sum_us_per_step": sum(t for c, t in loop.values()) / n_steps / 1e3,
           "launches_per_step": sum(c for c, t in loop.values()) / n_steps,
           "gpu_wall_us_per_step": wall, "kernels": kernels, "ops": ops_}
    if trace_out:
        with open(trace_out, "w") as fh:
            fh.write("name,launches_per_step,avg_us,us_per_step\n")
            for n, k in sorted(kernels.items(),
                               key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches_per_step"]):
                fh.write(f"\"{n}\",{k['launches_per_step']:.3f},{k['avg_us']:.3f},"
                         f"{k['avg_us'] * k['launches_per_step']:.3f}\n")
    return out


def inloop_other(name, steps, timeout=300):
    """`bench.py --config name` re-run under `rocprofv3 --kernel-trace` in a child process:
    {kernel name: {launches_per_step, avg_us}} of its timed loop (between the child's loop.begin /
    loop.end marker dispatches), or {"error": ...}."""
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix=f"aa_inloop_{name}_", dir="/tmp")
    cmd = [prof, "--kernel-trace", "-d", tmp, "-o", "inloop", "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--config", name, "--steps", str(steps),
           "--no-cpu-baseline", "--no-inloop"]
    env = dict(os.environ, AA_BENCH_TRACE_CHILD="1", TMPDIR="/tmp")
    try:
        r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True,
                           timeout=timeout)
    except subprocess.TimeoutExpired:
        shutil.rmtree(tmp, ignore_errors=True)
        return {"error": f"profiling pass exceeded {timeout}s"}
    child = None
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith("{") and "trace_child" in line:
            child = json.loads(line)
            break
    try:
        if r.returncode != 0 or child is None:
            return {"error": f"profiling pass failed (rc {r.returncode}): {r.stderr[-400:]}"}
        trace = _read_kernel_trace(tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    marks = [i for i, k in enumerate(trace) if "aa_marker_kernel" in k[0]]
    labels = child["marks"]
    if len(marks) != len(labels) or "loop.begin" not in labels:
        return {"error": f"{len(marks)} marker dispatches in the trace for {len(labels)} marks"}
    i0, i1 = marks[labels.index("loop.begin")], marks[labels.index("loop.end")]
    acc = {}
    for kname, st, en in trace[i0 + 1:i1]:
        e = acc.setdefault(kname, [0, 0])
        e[0] += 1
        e[1] += en - st
    n = child["loop_steps"]
    kernels = {k: {"launches_per_step": c / n, "avg_us": t / c / 1e3} for k, (c, t) in acc.items()}
    return {"steps": n, "ms_per_step_under_profiler": child["ms_per_step"],
            "kernel_time_sum_us_per_step": sum(t for c, t in acc.values()) / n / 1e3,
            "launches_per_step": sum(c for c, t in acc.values()) / n,
            "gpu_wall_us_per_step": (trace[i1][1] - trace[i0][2]) / n / 1e3, "kernels": kernels}


def _write_kernel_table(path, kernels):
    with open(path, "w") as fh:
        fh.write("name,launches_per_step,avg_us,us_per_step\n")
        for n, k in sorted(kernels.items(),
                           key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches_per_step"]):
            fh.write(f"\"{n}\",{k['launches_per_step']:.3f},{k['avg_us']:.3f},"
                     f"{k['avg_us'] * k['launches_per_step']:.3f}\n")


def _pmc_other(name):
    """Counter record of a --config ppo / sac kernel from the committed --pmc passes
    (profiles/r06_pmc_other.json, made by tools/profile_r06.sh), or {}."""
    try:
        with open(os.path.join(ROOT, PMC_OTHER_FILE)) as fh:
            return json.load(fh).get(name, {})
    except (OSError, ValueError):
        return {}


def run_other_config(name, steps, timeout=300):
    """`bench.py --config ppo|sac` in a child process: its JSON line (or {"error": ...})."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", name, "--steps", str(steps)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": f"--config {name} exceeded {timeout}s"}
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return {"error": f"--config {name} failed (rc {r.returncode}): {r.stderr[-300:]}"}


def cpu_baseline(S, steps, threads):
    """The oracle (numpy replay + torch-CPU DQN step) timed on this host: a bounded sample of the
    same workload (same shapes, batch 256, replay ring shortened to CPU_RING_FRAMES = 256 frames per
    env: 65,536 rows = 1.85 GB of host memory -- a working set beyond the host's last-level cache,
    where round 4's 8-frame ring (58 MB) sat inside it; the GPU side runs 3,906 frames per env)."""
    from oracle import dqn as odqn
    from oracle import nets as onets
    from oracle import optim as ooptim
    from oracle import replay as oreplay
    torch.set_num_threads(threads)
    B_env, L_ = 256, CPU_RING_FRAMES
    layers = onets.atari_q_layers(NUM_ACTIONS)
    params = onets.init_params(layers, OBS_SHAPE, seed=0)
    agent = odqn.OracleDqnAgent(layers, OBS_SHAPE, NUM_ACTIONS, params,
                                optimizer=ooptim.RMSprop(2.5e-4, 0.95, 0.95, 0.01, True),
                                gamma=0.99, loss="huber", target_update_period=2500)
    shapes = [(), OBS_SHAPE, (), (), (), ()]
    dtypes = [np.int32, np.uint8, np.int64, np.int32, np.float32, np.float32]
    rb = oreplay.OracleReplayBuffer(shapes, dtypes, B_env, L_, seed=1)
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 256, (B_env,) + OBS_SHAPE, dtype=np.uint8) for _ in range(4)]
    for i in range(L_):     # (untimed prefill: four distinct frame batches, cycled)
        rb.add_batch([rng.integers(0, 3, B_env).astype(np.int32),
                      frames[i & 3],
                      rng.integers(0, NUM_ACTIONS, B_env).astype(np.int64),
                      rng.integers(0, 3, B_env).astype(np.int32),
                      rng.choice([-1.0, 0.0, 1.0], B_env).astype(np.float32),
                      np.ones(B_env, np.float32)])

    def one():
        # collect: policy forward on 256 envs + add_batch ; learn: sample + train
        with torch.no_grad():
            obs = torch.from_numpy(rng.integers(0, 256, (B_env,) + OBS_SHAPE, dtype=np.uint8))
            q = agent.q_values(obs)
            act = q.argmax(1).numpy().astype(np.int64)
        rb.add_batch([np.ones(B_env, np.int32), obs.numpy(), act, np.ones(B_env, np.int32),
                      np.zeros(B_env, np.float32), np.ones(B_env, np.float32)])
        data, ids, probs = rb.get_next(S, 2)
        st, ob, ac, nst, rew, disc = data
        agent.train(torch.from_numpy(ob), ac, rew, disc, st)

    one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = (time.perf_counter() - t0) / steps
    return S / dt, dt


def _best_threads(fn):
    """Fastest of a few torch thread counts for a CPU leg (2 probe steps each)."""
    ncpu = max(1, os.cpu_count() or 1)
    cands = sorted({min(ncpu, t) for t in (4, 8, 16, 32)})
    probe = {t: fn(2, t) for t in cands}
    return max(probe, key=probe.get), ncpu, cands


def setup_ranks(args):
    """One process per GPU.  `python bench.py --gpus N` without a launcher re-launches itself under
    torch.distributed.run (one rank per GPU of this node, rendezvous on 127.0.0.1); under a
    launcher the process joins the RCCL process group.  Returns (world, rank, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher -- one rank per GPU under
        # torch.distributed.run on this node (rendezvous on 127.0.0.1), same arguments
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log(f"[bench] --gpus {args.gpus} without WORLD_SIZE: re-launching under "
            f"torch.distributed.run ({args.gpus} ranks, 127.0.0.1:{port})")
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    # AA_BENCH_BACKEND=gloo AA_BENCH_SHARE_GPU=1: development aid -- runs the N-rank control flow
    # (hooks, buckets, barriers, rank-0 breakdown) with every rank on GPU 0 when only one GPU exists
    backend = os.environ.get("AA_BENCH_BACKEND", "nccl")
    if os.environ.get("AA_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus says "
                             f"{args.gpus}")
        if rank == 0:
            log(f"[bench] process group up: backend {dist.get_backend()}, "
                f"{dist.get_world_size()} ranks, one rank per GPU (this rank: {dev})")

    return world, rank, dev


def main_prioritized(args):
    """`--replay prioritized`: configs[1]'s loop (256 envs, 1 M-row uint8 Atari replay, batch 256,
    Mnih-15 Q-net) with proportional prioritized sampling instead of uniform: every iteration is
    collect (add_batch gives the new rows the running maximum priority) -> get_next (two-level
    exact scan over the 1 M priorities, csrc/prio.hip) -> DqnAgent.train -> update_priorities from
    DqnLossInfo.td_error via Learner(after_train_strategy_step_fn=...).  One GPU, for the record;
    the graded line is the uniform one.  `roofline` = the sampler's scan launches."""
    if args.gpus != 1:
        raise SystemExit("--replay prioritized is a single-GPU line")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from agents_amd import _lib
    from agents_amd.utils import common
    _lib.load()
    S = args.batch
    w = build_workload(dev, 0, 1, args.envs, args.max_length, S, seed=1, replay="prioritized")
    prefill = args.max_length if args.prefill < 0 else min(args.prefill, args.max_length)
    w["init_driver"]._num_steps = args.envs * max(prefill, 2)
    w["init_driver"].run()
    torch.cuda.synchronize()
    rb, lrn = w["rb"], w["learner"]
    log(f"[bench] prioritized replay: prefilled {rb.num_frames()} frames")
    it = iter(w["dataset"])
    collect_run = common.function(w["collect_driver"].run)
    time_step = None

    def step():
        nonlocal time_step
        time_step, _ = collect_run(time_step)
        return lrn.run(iterations=1, iterator=it)

    from agents_amd.utils import graph
    # untimed until no HIP graph has been captured for a while: the sampled batches are fresh
    # tensors, and GraphedTrain gives an address set that comes back its own graph
    quiet, seen, primed = 0, graph.capture_count(), 0
    while primed < 1200 and quiet < 100:    # (up to 16 address sets get a graph, then no more)
        step()
        primed += 1
        now = graph.capture_count()
        quiet = quiet + 1 if now == seen else 0
        seen = now
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    captures_before = graph.capture_count()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        li = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    captures_in_timed_region = graph.capture_count() - captures_before
    # the sampler alone: HIP events around back-to-back draws (index sampling only, no row gather)
    reps = 50
    rb._sample_rows(S, 2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        rb._sample_rows(S, 2)
    e1.record()
    torch.cuda.synchronize()
    scan_ms = e0.elapsed_time(e1) / reps
    cap = rb.capacity
    n_blocks = (cap + 1023) // 1024
    # level 1 reads every priority (4 B) and stored id (8 B); level 2, per sample, the block sums
    # (8 B each) and one block's 1,024 priorities + ids
    scan_bytes = cap * 12 + S * (n_blocks * 8 + 1024 * 12)
    pri = rb.priorities()
    out = {"metric": "replay_samples_per_sec trained, DQN Atari b=256, PRIORITIZED replay "
                     "(proportional, priorities from td_error each step)",
           "value": S / dt, "unit": "samples/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "learner_steps_per_sec": 1.0 / dt, "final_loss": float(li.loss),
           "prime_steps": primed, "captures_in_timed_region": captures_in_timed_region,
           "config": {"workload": "configs[1] with TFPrioritizedReplayBuffer: DQN Atari "
                                  f"Pong-shaped, replay {args.envs}x{args.max_length} rows, "
                                  f"batch={S}, num_steps=2, Mnih-15 Q-net; collect -> prioritized "
                                  "get_next -> train -> update_priorities(td_error) per iteration "
                                  "(no sampling ahead, one stream)",
                      "global_batch": S, "envs_per_gpu": args.envs,
                      "replay_rows_per_gpu": cap, "parallelism": "single",
                      "replay": "prioritized"},
           "roofline": {"kernel": "aa_prio_block_sums_kernel + aa_prio_sample_kernel "
                                  "(csrc/prio.hip): exact uint64 two-level scan over all "
                                  "priorities, one draw of 256 window starts",
                        "bound": "hbm", "achieved": scan_bytes / scan_ms / 1e6,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": scan_bytes / scan_ms / 1e6 / HBM_PEAK_GBS, "traffic": None,
                        "avg_launch_ms": scan_ms, "algorithmic_bytes_per_launch": scan_bytes,
                        "duration_source": "HIP events around 50 back-to-back draws"},
           "priorities": {"max": float(pri.max()), "mean_nonzero":
                          float(pri[pri > 0].mean()), "rows_with_priority": int((pri > 0).sum())}}
    print(json.dumps(out), flush=True)


def main_other_config(args):
    """`--config ppo` / `--config sac`: BASELINE.json configs[2] and configs[4] at one GPU -- parity
    configurations, measured for the record with the same JSON contract (the graded metric is the
    DQN line).  A "step" is one iteration of the config's train_eval loop.  `roofline` is the
    DOMINANT KERNEL of the loop with its duration INSIDE the loop (a rocprofv3 --kernel-trace child
    pass, like the DQN line's), algorithmic flop and bytes per launch, and the counter traffic of
    the committed --pmc pass."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    # `--gpus N`: one process per GPU, weak scaling like the DQN line (per-rank environments and
    # replay shard, gradients SUM all-reduced per train step through the Learner's strategy);
    # BASELINE.json configs[4] is an 8-GPU configuration
    world, rank, dev = setup_ranks(args)
    from agents_amd import _lib
    _lib.load()
    par = "single" if world == 1 else f"dp{world}"
    want_inloop = not args.no_inloop and not TRACE_CHILD and world == 1
    if args.config == "ppo":
        import bench_ppo
        a = argparse.Namespace(envs=2048, steps=128, minibatch=4096, epochs=10,
                               iters=max(args.steps, 1) if args.steps_given else 3,
                               mark=mark if TRACE_CHILD else None)
        r = bench_ppo.run(a, dev=dev, rank=rank, world=world)
        r.pop("agent")
        it_s = r["iteration_s"]
        n_mb = r["minibatch_steps_per_iteration"]
        if TRACE_CHILD:
            print(json.dumps({"trace_child": True, "marks": MARKS, "loop_steps": a.iters,
                              "ms_per_step": it_s * 1e3}), flush=True)
            return
        # actor + value MLPs (17-64-64-6, 17-64-64-1): forward + backward = 6 flop per weight
        # and frame
        weights = (17 * 64 + 64 * 64 + 64 * 6) + (17 * 64 + 64 * 64 + 64 * 1)
        flop = 6.0 * weights * 4096
        # K1 (aa_ppo_fused_step_kernel) per launch: gathers 4,096 rows of 10 leaves (172 B: obs
        # 17, action 6, old loc / scale 6 + 6, return, advantage, weight, step type ...), reads
        # the parameters once per workgroup (from L2), writes one gradient slab per workgroup of
        # 16 frames (256 slabs x total_params x 4 B) -- the slabs are the traffic
        total = r.get("total_params", 11085)
        n_wg = 4096 // 16
        k1_bytes = 4096 * 172 + n_wg * total * 4 + total * 4
        mb_ms = r["train_s_per_iteration"] / n_mb * 1e3
        il = inloop_other("ppo", 2) if want_inloop else None
        k1 = None
        if il and "error" not in il:
            k1 = next((v for k, v in il["kernels"].items() if "aa_ppo_fused_step_kernel" in k), None)
            if args.trace_out:
                _write_kernel_table(args.trace_out, il["kernels"])
        elif il:
            log(f"[bench] ppo in-loop pass failed: {il['error']}")
        pmc = _pmc_other("ppo_fused_step")
        k1_ms = k1["avg_us"] / 1e3 if k1 else None
        roof = {"kernel": "aa_ppo_fused_step_kernel (K1 of the 2-launch minibatch step: row "
                          "gather, both (64,64) MLPs forward + backward, loss; csrc/ppo_fused.hip)",
                "bound": "hbm",
                "achieved": k1_bytes / (k1_ms * 1e-3) / 1e9 if k1_ms else None,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": k1_bytes / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k1_ms else None,
                "traffic": pmc.get("bytes_per_launch"),
                "traffic_source": PMC_OTHER_FILE if pmc else None,
                "avg_launch_ms": k1_ms,
                "launches_per_step": k1["launches_per_step"] if k1 else None,
                "duration_source": "in-loop rocprofv3" if k1 else "unmeasured (no profiler pass)",
                "algorithmic_bytes_per_launch": k1_bytes, "algorithmic_flop_per_launch": flop,
                "flop_frac_of_fp32_mfma_peak": (flop / (k1_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS
                                                if k1_ms else None),
                "mfma_busy": pmc.get("mfma_busy"),
                "minibatch_step_us": mb_ms * 1e3,
                "note": "launch / latency bound, as SURVEY.md 8(d) says of the (64,64) MLPs: 0.27 "
                        "GFLOP and 12 MB per launch are 2 us of either pipe; the launch is "
                        "staging + 6 layer steps behind barriers + the slab store (in-kernel "
                        "timeline: DESIGN.md).  The minibatch step is K1 + reduce + apply = "
                        "minibatch_step_us; that figure, not frac, is what the config is "
                        "judged by"}
        out = {"metric": "PPO frames trained per second (minibatch steps/s x 4096), configs[2] "
                         "HalfCheetah-shaped", "value": r["train_frames_per_sec"],
               "unit": "frames/s", "n_gpus": world, "steps": a.iters, "warmup": 1,
               "ms_per_step": it_s * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": r["workload"] + "; reward + observation normalisers on "
                          "(the reference's defaults); step = collect 2048 x 129 env steps + "
                          f"{n_mb} minibatch train steps, per GPU", "parallelism": par},
               "rccl_ranks": world, "collectives": r.get("collectives"),
               "collect_env_steps_per_sec": r["collect_env_steps_per_sec"],
               "train_minibatch_steps_per_sec": r["train_minibatch_steps_per_sec"],
               "replay_rows_per_sec": world * 2048 * 129 / it_s,   # gather_all rows per iteration
               "roofline": roof, "inloop": il}
        if not args.no_cpu_baseline and rank == 0 and world == 1:
            th, ncpu, cands = _best_threads(lambda n, t: bench_ppo.cpu_baseline(4096, n, t))
            sps = bench_ppo.cpu_baseline(4096, 200, th)
            out["cpu_baseline"] = {"value": sps * 4096, "unit": "frames/s", "cores": th,
                                   "kind": "port", "minibatch_steps_per_sec": sps,
                                   "sample": f"200 minibatch steps (4,096 frames, actor + value "
                                             f"MLP (64,64), clipped surrogate, Adam) of the "
                                             f"torch-CPU oracle on {th} of {ncpu} host threads "
                                             f"(fastest of {cands})"}
    else:
        import bench_sac
        a = argparse.Namespace(envs=4096, max_length=args.sac_max_length, batch=256,
                               iters=args.steps if args.steps_given else 200,
                               mark=mark if TRACE_CHILD else None)
        r = bench_sac.run(a, dev=dev, rank=rank, world=world)
        dt = r["ms_per_iteration"]
        if TRACE_CHILD:
            print(json.dumps({"trace_child": True, "marks": MARKS, "loop_steps": a.iters,
                              "ms_per_step": dt}), flush=True)
            return
        # aa_mlp_wide_fwd_kernel: a workgroup takes 4 samples through the WHOLE network and
        # streams every layer's weights from L2 (VALU fp32 FMAs, no MFMA: csrc/mlp_wide.hip).
        # Per launch of (n networks, batch B, widths d0..dn): flop = 2 B n sum(d_i d_{i+1});
        # L2 -> CU stream = n ceil(B/4) x weight bytes; HBM side = weights once + inputs + outputs.
        launches = r.get("wide_fwd_launches") or []
        fl = l2b = hbm = 0.0
        for n_nets, B, dims in launches:
            w_el = sum(dims[i] * dims[i + 1] + dims[i + 1] for i in range(len(dims) - 1))
            fl += 2.0 * B * n_nets * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
            l2b += n_nets * -(-B // 4) * w_el * 4.0
            hbm += n_nets * (w_el * 4.0 + B * 4.0 * sum(dims))
        n_l = max(len(launches), 1)
        il = inloop_other("sac", 100) if want_inloop else None
        kf = None
        if il and "error" not in il:
            kf = next((v for k, v in il["kernels"].items() if "aa_mlp_wide_fwd_kernel" in k), None)
            if args.trace_out:
                _write_kernel_table(args.trace_out, il["kernels"])
        elif il:
            log(f"[bench] sac in-loop pass failed: {il['error']}")
        pmc = _pmc_other("mlp_wide_fwd")
        kf_ms = kf["avg_us"] / 1e3 if kf else None
        roof = {"kernel": f"aa_mlp_wide_fwd_kernel ({len(launches)} launches per iteration: "
                          "actor / twin-critic / 4-critic forwards of whole (256,256) MLPs; "
                          "csrc/mlp_wide.hip)",
                "bound": "l2",
                "achieved": l2b / n_l / (kf_ms * 1e-3) / 1e9 if kf_ms else None,
                "peak": L2_PEAK_GBS, "unit": "GB/s",
                "frac": l2b / n_l / (kf_ms * 1e-3) / 1e9 / L2_PEAK_GBS if kf_ms else None,
                "traffic": pmc.get("bytes_per_launch"),
                "traffic_source": PMC_OTHER_FILE if pmc else None,
                "avg_launch_ms": kf_ms,
                "launches_per_step": kf["launches_per_step"] if kf else None,
                "duration_source": "in-loop rocprofv3" if kf else "unmeasured (no profiler pass)",
                "algorithmic_bytes_per_launch": l2b / n_l,
                "algorithmic_hbm_bytes_per_launch": hbm / n_l,
                "algorithmic_flop_per_launch": fl / n_l,
                "flop_frac_of_fp32_peak": (fl / n_l / (kf_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS
                                           if kf_ms else None),
                "note": "averages over the launches of one iteration (listed in the detail "
                        "record).  The bytes are the L2 -> CU weight stream (every workgroup of 4 "
                        "samples reads its network's 0.66 MB), which is what bounds the kernel; "
                        "the HBM side is the weights once.  256-row launches put 64 workgroups "
                        "per network on 256 CUs: the chip-level fraction is low by shape, the "
                        "per-CU stream (0.66 MB in ~13 us = 50 GB/s of a CU's ~150) is the "
                        "kernel-quality figure"}
        out = {"metric": "SAC learner steps/sec (batch 256) + 4,096-env collect, configs[4] at 1 GPU",
               "value": r["learner_steps_per_sec"], "unit": "steps/s", "n_gpus": world,
               "steps": a.iters, "warmup": 40, "ms_per_step": dt, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": r["workload"] + "; step = 1 collect step (4096 envs) + "
                          "sample 256x2 + 1 SacAgent.train, per GPU (global batch "
                          f"{256 * world})", "parallelism": par,
                          "replay_rows_per_gpu": r["replay_rows"],
                          "replay_bytes_per_gpu": r["replay_bytes"]},
               "env_steps_per_sec": r["env_steps_per_sec"],
               "trained_transitions_per_sec": r["trained_transitions_per_sec"],
               "replay_rows_per_sec": r["replay_rows_per_sec"],
               "rccl_ranks": world, "collectives": r.get("collectives"),
               "roofline": roof, "wide_fwd_launches": launches, "inloop": il}
        if not args.no_cpu_baseline and rank == 0 and world == 1:
            th, ncpu, cands = _best_threads(lambda n, t: bench_sac.cpu_baseline(256, n, t))
            sps = bench_sac.cpu_baseline(256, 300, th)
            out["cpu_baseline"] = {"value": sps, "unit": "steps/s", "cores": th, "kind": "port",
                                   "sample": f"300 OracleSacAgent.train steps (batch 256, actor + "
                                             f"twin critics (256,256), three Adam) on {th} of "
                                             f"{ncpu} host threads (fastest of {cands}); no "
                                             "collect step on the CPU side"}
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        if rank == 0:
            print(json.dumps(out), flush=True)
        dist.destroy_process_group()
        return
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=["dqn", "ppo", "sac"], default="dqn",
                    help="dqn = BASELINE.json's metric (configs[1] / [3]); ppo / sac = configs[2] / "
                         "configs[4] at one GPU, for the record")
    ap.add_argument("--replay", choices=["uniform", "prioritized"], default="uniform",
                    help="prioritized = the DQN loop on TFPrioritizedReplayBuffer (proportional "
                         "sampling, csrc/prio.hip) with priorities fed back from td_error through "
                         "the Learner's after_train_strategy_step_fn; one GPU, for the record")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--max-length", type=int, default=3906,
                    help="replay frames per env (3906 x 256 envs = the 1M-row config)")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--cpu-steps", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="replay the collect / sample / train graphs on ONE stream (= --lanes off)")
    ap.add_argument("--lanes", choices=["auto", "on", "off"], default="on",
                    help="on (default) = collect / sample / train graphs on three HIP streams "
                         "ordered by events; off = one stream; auto = time both for --lane-probe "
                         "untimed iterations after priming and keep the faster one (measured on "
                         "the pool's slow and fast boxes alike: three lanes win, 0.408 vs 0.447 "
                         "and 0.352 vs 0.401 ms)")
    ap.add_argument("--lane-probe", type=int, default=60)
    ap.add_argument("--prefill", type=int, default=-1, help="frames per env to prefill (-1 = all)")
    ap.add_argument("--host-profile", type=int, default=0,
                    help="cProfile this many extra steps after the timed region (stderr)")
    ap.add_argument("--no-inloop", action="store_true",
                    help="skip the rocprofv3 --kernel-trace pass that measures kernel durations "
                         "inside the loop (the roofline then falls back to isolated launches)")
    ap.add_argument("--inloop-steps", type=int, default=100)
    ap.add_argument("--trace-out", default=None,
                    help="write the in-loop per-kernel table (CSV) here")
    ap.add_argument("--sac-max-length", type=int, default=1000,
                    help="--config sac: replay frames per env (1000 x 4096 envs = configs[4]'s "
                         "4 M-row / 6.5 GB replay)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short --config ppo / --config sac runs appended to the line")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="full record (roofline_all, in-loop kernel table, notes); the stdout "
                         "line is the compact one (< 4 KB)")
    ap.add_argument("--steady-steps", type=int, default=300,
                    help="when --steps < 200, time this many further steps after the timed "
                         "region and report them as steady_ms_per_step (0 = off)")
    args = ap.parse_args()
    args.steps_given = any(a == "--steps" or a.startswith("--steps=") for a in sys.argv[1:])
    if args.config != "dqn":
        return main_other_config(args)
    if args.replay == "prioritized":
        return main_prioritized(args)

    world, rank, dev = setup_ranks(args)
    local_rank = dev.index
    backend = os.environ.get("AA_BENCH_BACKEND", "nccl")
    from agents_amd import _lib
    _lib.load()  # fail loudly if the HIP library is missing
    S = args.batch
    w = build_workload(dev, rank, world, args.envs, args.max_length, S, seed=1)
    # ---- prefill the replay shard with the random policy (untimed) ---------------------------
    t0 = time.perf_counter()
    prefill = args.max_length if args.prefill < 0 else min(args.prefill, args.max_length)
    w["init_driver"]._num_steps = args.envs * max(prefill, 2)
    w["init_driver"].run()
    torch.cuda.synchronize()
    if rank == 0:
        log(f"[bench] prefilled {w['rb'].num_frames()} frames "
            f"({w['rb'].num_frames() * ROW_BYTES / 1e9:.1f} GB) in {time.perf_counter() - t0:.1f}s")
    it = iter(w["dataset"])
    lrn, drv = w["learner"], w["collect_driver"]
    # train_eval.py:234-237: `collect_driver.run = common.function(collect_driver.run)`
    from agents_amd.utils import common, graph
    collect_run = common.function(drv.run)
    lanes_mode = "off" if args.no_overlap else args.lanes
    if lanes_mode != "off":
        # collect / sample / train graphs on three HIP streams, ordered by events along the true
        # data dependencies (agents_amd/utils/graph.py: Lanes).  Bit-identical results either way.
        graph.enable_overlap(dev)
    time_step = None

    # A/B aid: AA_BENCH_HOST_DELAY_US=n burns n us of host time per iteration.  If the throughput
    # does not move, the host has that much slack (the loop is GPU-bound); if it drops by n us per
    # iteration, the loop is bound by the host's enqueue work.
    host_delay = float(os.environ.get("AA_BENCH_HOST_DELAY_US", "0")) * 1e-6

    def step():
        nonlocal time_step
        if host_delay:
            t_ = time.perf_counter() + host_delay
            while time.perf_counter() < t_:
                pass
        time_step, _ = collect_run(time_step)
        return lrn.run(iterations=1, iterator=it)

    def sync_all():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # ---- priming (untimed, independent of --warmup) ------------------------------------------
    # The first iterations are not the steady state: two eager calls per program size the buffers,
    # the third records every HIP graph of the loop (both driver bodies, the whole sampler ring,
    # one train graph per ring slot: agents_amd/utils/graph.py).  Iterate until no capture has
    # happened for three consecutive iterations, the way the reference's harness discards its
    # first log window (tf_agents/benchmark/utils.py:89-180).  Same count on every rank.
    # (AA_BENCH_PRIME_MIN=n primes for at least n iterations.  Round 5 checked whether the 3-5 %
    # between the 20 timed steps of the driver's command line and the 300 steady ones behind them
    # -- BENCH_r04: 0.3223 vs 0.3066 ms -- is a warm-up effect: it is not.  Same box, alternating:
    # 6 priming iterations 0.3704 / 0.3706 ms, 300 of them 0.3638 / 0.3707 ms, steady 0.357-0.363
    # either way (profiles/r05_f_prime.txt).  The difference is the protocol's: the timed region
    # ends with a device synchronisation, i.e. with the drain of the last iteration the host had
    # run ahead of -- one iteration's latency spread over 20 steps.)
    prime_min = PRIME_MIN
    prime_steps, quiet, seen = 0, 0, graph.capture_count()
    t_prime = time.perf_counter()
    while (prime_steps < 64 and quiet < 3) or prime_steps < prime_min:
        step()
        prime_steps += 1
        now = graph.capture_count()
        quiet = quiet + 1 if (now == seen and now > 0) else 0
        seen = now
    sync_all()
    t_prime = time.perf_counter() - t_prime
    if rank == 0:
        log(f"[bench] primed in {prime_steps} iterations / {t_prime:.2f}s "
            f"({seen} HIP-graph captures)")
    # ---- lane mode (untimed): the same loop, bit-identical results, on three streams or on one.
    # On most boxes of the pool three lanes win by ~10 %; on some the kernels of concurrent streams
    # stretch each other so much (in-loop kernel time 667 us against 505 on the same tree) that
    # one stream is as fast.  `--lanes auto` measures, like a library's autotuning warm-up.
    lane_probe = None
    if lanes_mode == "auto" and not TRACE_CHILD:
        def timed(n):
            for _ in range(10):
                step()
            graph.join_lanes(dev)
            sync_all()
            t_ = time.perf_counter()
            for _ in range(n):
                step()
            graph.join_lanes(dev)
            sync_all()
            return (time.perf_counter() - t_) / n * 1e3
        t_on = timed(args.lane_probe)
        graph.disable_overlap(dev)
        t_off = timed(args.lane_probe)
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([t_on, t_off], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)      # every rank takes the same decision
            t_on, t_off = float(tt[0]), float(tt[1])
        lanes_mode = "on" if t_on <= t_off else "off"
        if lanes_mode == "on":
            graph.enable_overlap(dev)
        lane_probe = {"three_lanes_ms": t_on, "one_stream_ms": t_off, "steps_each": args.lane_probe}
        if rank == 0:
            log(f"[bench] lane probe: three lanes {t_on:.4f} ms, one stream {t_off:.4f} ms per "
                f"iteration -> lanes {lanes_mode}")
    elif lanes_mode == "auto":
        lanes_mode = "on"
    args.lanes_decided = lanes_mode
    for _ in range(args.warmup):
        step()
    sync_all()
    captures_before = graph.capture_count()
    if hasattr(lrn.strategy, "reset_stats"):
        lrn.strategy.reset_stats()
    mark("loop.begin")
    wait0 = getattr(collect_run, "wait_seconds", 0.0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss_info = step()
    t_enqueue = time.perf_counter() - t0     # host time to issue the steps (the GPU trails behind)
    # ... of which the host WAITED for the GPU (the driver's step-count post of the previous
    # collect step: the one point where the loop throttles the host).  enqueue - wait = host WORK;
    # the loop is host-bound when that equals ms_per_step, GPU-bound when the wait is what fills it
    t_wait = getattr(collect_run, "wait_seconds", 0.0) - wait0
    graph.join_lanes(dev)
    sync_all()
    dt = time.perf_counter() - t0
    mark("loop.end")
    captures_in_timed_region = graph.capture_count() - captures_before
    collectives = None
    if world > 1 and hasattr(lrn.strategy, "stats"):
        # what the data path exchanged per step in the timed region, then an untimed pass with
        # timing events around the collectives / the waits for them: their EXPOSED stream time
        st_ = dict(lrn.strategy.stats)
        lrn.strategy.reset_stats()
        lrn.strategy.profile = True
        n_prof = min(max(args.steps, 10), 50)
        for _ in range(n_prof):
            step()
        graph.join_lanes(dev)
        sync_all()
        collectives = {"allreduce_calls_per_step": st_["calls"] / args.steps,
                       "allreduce_bytes_per_step": st_["bytes"] / args.steps,
                       "allreduce_exposed_ms_per_step": lrn.strategy.exposed_ms() / n_prof,
                       "backend": lrn.strategy.backend,
                       "ranks": lrn.strategy.num_replicas_in_sync}
        lrn.strategy.profile = False
        lrn.strategy.reset_stats()
        # a data-parallel line is only one if every step moved the WHOLE gradient buffer through
        # the communicator of --gpus ranks (+ the 3 LossInfo sums Learner.run reduces): anything
        # else -- a rank on its own, a skipped bucket -- must not print a number
        grad_bytes = w["net"].flat_params.numel() * 4
        got = collectives["allreduce_bytes_per_step"]
        if collectives["ranks"] != args.gpus or not grad_bytes <= got <= grad_bytes + 64:
            raise SystemExit(
                f"[bench] data-parallel check failed: communicator of {collectives['ranks']} ranks "
                f"for --gpus {args.gpus}; {got:.0f} all-reduce bytes per step, expected "
                f"{grad_bytes} gradient bytes (+ <= 64 of LossInfo sums)")
    # `--steps 20` times ~8 ms of wall clock: a second, longer region of the same loop right after
    # it says whether the short one was representative (reported beside it, never instead of it)
    steady = None
    if (not TRACE_CHILD and args.steady_steps > 0 and args.steps < 200
            and not args.host_profile):
        sync_all()
        t1 = time.perf_counter()
        for _ in range(args.steady_steps):
            step()
        graph.join_lanes(dev)
        sync_all()
        steady = (time.perf_counter() - t1) / args.steady_steps * 1e3
    if TRACE_CHILD:
        # child of inloop_profile(): the isolated cases once more, bracketed by markers, so that
        # the parent learns which device kernels each case launches; then one JSON line
        graph.disable_overlap()
        bd = kernel_breakdown(w, S)
        print(json.dumps({"trace_child": True, "marks": MARKS, "ops": [r[0] for r in bd],
                          "reps": 20, "steps": args.steps,
                          "ms_per_step": dt / args.steps * 1e3}), flush=True)
        return
    if args.host_profile and rank == 0:
        # plain timers first (cProfile inflates Python frames): host time inside the two calls
        tc = tt = 0.0
        n_hp = args.host_profile
        wait0 = getattr(collect_run, "wait_seconds", 0.0)
        graph.REPLAY_TIMERS = {}
        for _ in range(n_hp):
            a = time.perf_counter()
            time_step, _ = collect_run(time_step)
            b = time.perf_counter()
            lrn.run(iterations=1, iterator=it)
            c = time.perf_counter()
            tc += b - a
            tt += c - b
        sync_all()
        wait_us = (getattr(collect_run, "wait_seconds", 0.0) - wait0) / n_hp * 1e6
        log(f"[bench] host time per iteration: driver.run {tc / n_hp * 1e6:.1f} us "
            f"(of which waiting for the step-count post {wait_us:.1f} us), "
            f"learner.run {tt / n_hp * 1e6:.1f} us; HIP-graph launches: " +
            ", ".join(f"{k} {v[1] / max(v[0], 1) * 1e6:.1f} us x{v[0] / n_hp:.0f}"
                      for k, v in sorted(graph.REPLAY_TIMERS.items())))
        graph.REPLAY_TIMERS = None
        # GPU-side timeline of the last iterations (timing events between the graph launches):
        # offsets in us from the end of the previous optimizer step
        graph.TIMELINE = []
        for _ in range(40):
            step()
        sync_all()
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
            if "train.apply_done" not in a or "train.apply_done" not in b:
                continue
            t0 = a["train.apply_done"]
            rows.append({k: t0.elapsed_time(v) * 1e3 for k, v in b.items()})
        if rows:
            keys = ["collect.begin", "collect.done", "sample.begin", "sample.done", "train.begin",
                    "train.grads_done", "train.apply_done", "early_target.begin",
                    "early_target.done"]
            log("[bench] GPU timeline of an iteration, us after the previous optimizer step "
                "(mean over %d): " % len(rows) +
                ", ".join(f"{k} {sum(r[k] for r in rows if k in r) / len(rows):.0f}"
                          for k in keys))
        # where the HOST spends an iteration (the loop is enqueue-bound when host_enqueue_ms_per_step
        # equals ms_per_step): cProfile over further steps, top entries by own time
        import cProfile
        import io
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(args.host_profile):
            step()
        pr.disable()
        sync_all()
        buf = io.StringIO()
        st = pstats.Stats(pr, stream=buf)
        st.sort_stats("tottime").print_stats(28)
        st.sort_stats("cumulative").print_stats(28)
        log(buf.getvalue())
    graph.disable_overlap()
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    steps_per_sec = args.steps / dt
    value = steps_per_sec * S * world
    loss_val = float(loss_info.loss.item())

    out = {
        "metric": "replay_samples_per_sec trained (= learner_steps_per_sec x 256 x n_gpus), "
                  "DQN Atari b=256",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "prime_steps": prime_steps, "prime_seconds": t_prime,
        "captures_in_timed_region": captures_in_timed_region,
        "host_enqueue_ms_per_step": t_enqueue / args.steps * 1e3,
        "host_wait_ms_per_step": t_wait / args.steps * 1e3,
        "host_work_ms_per_step": (t_enqueue - t_wait) / args.steps * 1e3,
        "learner_steps_per_sec": steps_per_sec,
        "env_steps_per_sec": steps_per_sec * args.envs * world,
        # SURVEY.md 8(d) reads "replay samples/sec" as the rows get_next returns: S x T per step
        "replay_rows_per_sec": steps_per_sec * S * 2 * world,
        "final_loss": loss_val, "rccl_ranks": world, "collectives": collectives,
        "lanes": lanes_mode,
        "config": {"workload": "configs[1]: DQN Atari Pong-shaped (84x84x4 uint8 stack), replay "
                               f"{args.envs}x{args.max_length} rows/GPU, batch={S}, num_steps=2, "
                               "Mnih-15 Q-net, Huber, centred RMSProp, 1 collect step (256 envs) "
                               "+ 1 train step per iteration",
                   "global_batch": S * world, "envs_per_gpu": args.envs,
                   "replay_rows_per_gpu": args.envs * args.max_length,
                   "parallelism": f"dp{world}" if world > 1 else "single"},
    }
    if lane_probe is not None:
        out["lane_probe"] = lane_probe
    if steady is not None:
        if world > 1:
            import torch.distributed as dist
            tm = torch.tensor([steady], dtype=torch.float64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            steady = float(tm.item())
        out["steady_ms_per_step"] = steady
        out["steady_steps"] = args.steady_steps
    if rank == 0:
        flops_step = train_flops_per_sample() * S + 2.0 * sum(fwd_macs_per_sample()) * args.envs
        out["step_algorithmic_gflop"] = flops_step / 1e9
        out["step_mfma_frac"] = flops_step / (dt / args.steps) / 1e12 / MFMA_F32_PEAK_TFLOPS
        if not args.no_breakdown:
            bd = kernel_breakdown(w, S)
            tot = sum(r[1] * r[2] for r in bd)
            # ---- kernel durations INSIDE the loop (rocprofv3 --kernel-trace pass of this loop) ----
            inloop = None
            if world == 1 and not args.no_inloop:
                t_il = time.perf_counter()
                inloop = inloop_profile(args, trace_out=args.trace_out)
                if "error" in inloop:
                    log(f"[bench] in-loop profiling pass unavailable: {inloop['error']}")
                else:
                    log(f"[bench] in-loop pass: {inloop['steps']} steps under rocprofv3 in "
                        f"{time.perf_counter() - t_il:.0f}s; {inloop['launches_per_step']:.1f} "
                        f"launches / step, kernel time {inloop['kernel_time_sum_us_per_step']:.0f} "
                        f"us inside {inloop['gpu_wall_us_per_step']:.0f} us of GPU wall per step")
            il_ops = inloop.get("ops", {}) if inloop and "error" not in inloop else {}
            log("[bench] per-launch GPU time of the iteration's kernels: isolated (HIP events "
                "around a graph of 20 launches) | in the loop (rocprofv3) x launches per step:")
            for name, ms, n, fl, by, prod in bd:
                il = il_ops.get(name, {}).get("inloop_us")
                t_us = il if il else ms * 1e3
                extra = f"{fl / t_us / 1e6:8.1f} TFLOP/s" if fl else f"{by / t_us / 1e3:8.1f} GB/s"
                log(f"    {name:40s} {ms * 1e3:8.1f} | "
                    f"{(f'{il:8.1f}' if il else '     n/a')} us x{n}  {extra}")
            # HBM bytes and matrix-pipe busy per launch come from COMMITTED rocprofv3 --pmc passes
            # (isolated launches, one counter set per pass: tools/pmc_r02.sh, run by tools/profile_r03.sh); they are not
            # re-measured by this run and are labelled so
            pmc_case = {"conv2+conv3.fwd(fused)": "conv23.fwd",
                        "fc1.fwd": "fc1.fwd",
                        "fc1+fc2.fwd(head sums fc1's slabs)": "fc1.fwd",
                        "fc1.dX": "fc1.dX", "fc1.dW(+bias grad)": "fc1.dW",
                        "conv2.dX": "conv2.dX", "conv3.dX": "conv3.dX",
                        "conv2.dW(+bias grad)": "conv2.dW", "conv3.dW(+bias grad)": "conv3.dW",
                        "conv1.fwd(u8)": "conv1.fwd",
                        "replay.get_next(sample+gather 512 rows)": "replay.get_next"}
            pmc, pmc_src = {}, None
            for cand in (PMC_FILE, os.path.join("profiles", "r05_pmc.json")):
                if os.path.exists(os.path.join(ROOT, cand)):
                    with open(os.path.join(ROOT, cand)) as fh:
                        pmc = json.load(fh).get("cases", {})
                    pmc_src = cand
                    break

            def pmc_of(name):
                c = pmc.get(pmc_case.get(name, ""), None)
                if not c:
                    return None, None, None
                ks = c["kernels"]
                tb = [k.get("hbm_bytes_per_launch") for k in ks.values()]
                traffic_b = sum(t for t in tb if t is not None) if any(
                    t is not None for t in tb) else None
                dom = ks[c["dominant_kernel"]]
                return traffic_b, dom.get("mfma_busy"), c["dominant_kernel"]

            def roof(row):
                """`achieved` / `frac` use the IN-LOOP duration when the profiling pass gave one
                (`duration_source` says which); `isolated_ms` is always the HIP-event figure."""
                name, ms, n, fl, by, prod = row
                traffic_b, busy, dom = pmc_of(name)
                il = il_ops.get(name, {})
                t_ms = il["inloop_us"] / 1e3 if il.get("inloop_us") else ms
                src = "in-loop (rocprofv3 --kernel-trace pass of this loop, run by bench.py)" \
                    if il.get("inloop_us") else "isolated (graph of 20 launches, HIP events)"
                common_ = {"kernel": name, "avg_launch_ms": t_ms, "isolated_ms": ms,
                           "duration_source": src, "launches_per_step": n,
                           "device_kernels": il.get("device_kernels", dom),
                           "traffic": traffic_b,
                           "traffic_source": None if traffic_b is None else
                           f"{pmc_src} (committed --pmc passes of isolated launches; not "
                           "measured in this run)"}
                if fl:
                    ach = fl / t_ms / 1e9
                    r = dict(common_, bound="mfma", achieved=ach, peak=MFMA_F32_PEAK_TFLOPS,
                             unit="TFLOP/s", frac=ach / MFMA_F32_PEAK_TFLOPS,
                             frac_isolated=fl / ms / 1e9 / MFMA_F32_PEAK_TFLOPS,
                             algorithmic_flop_per_launch=fl, mfma_busy=busy,
                             mfma_busy_source=None if busy is None else pmc_src)
                    if prod:
                        # the kernel executes `prod` bf16 MFMA products per fp32 product on the
                        # 2.5 PFLOP/s pipe: against 157.3 TFLOP/s its ceiling is 2500/(prod*157.3)
                        r.update(bf16_products_per_fp32_product=prod,
                                 pipe_frac=prod * fl / t_ms / 1e9 / MFMA_BF16_PEAK_TFLOPS,
                                 frac_ceiling=MFMA_BF16_PEAK_TFLOPS / (prod * MFMA_F32_PEAK_TFLOPS))
                    r["peak_note"] = (
                        "frac = algorithmic fp32 flop / in-loop launch time / 157.3 TFLOP/s (the "
                        "fp32-input MFMA peak: the dtype of the contraction).  Kernels with "
                        "bf16_products_per_fp32_product run on the bf16 pipe, where frac can "
                        "reach frac_ceiling (2.65 for six products); pipe_frac = products x flop "
                        "/ time / 2,500 TFLOP/s is their utilisation of the pipe they use")
                    return r
                ach = by / t_ms / 1e6
                return dict(common_, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=ach / HBM_PEAK_GBS, frac_isolated=by / ms / 1e6 / HBM_PEAK_GBS,
                            algorithmic_bytes_per_launch=by)

            rows = [roof(r) for r in bd]
            # dominant KERNEL = the single device kernel with the largest share of the iteration's
            # GPU time (in-loop average duration x launches per step, from the profiling pass).  A
            # row of the table that is several launches (fc1's GEMM + the head that sums its
            # slabs) competes with its largest kernel, not with the row total; `roofline` is the
            # row that kernel belongs to, `dominant_device_kernel` names it.
            il_k = inloop.get("kernels", {}) if inloop and "error" not in inloop else {}

            def row_weight(r):
                dk = r.get("device_kernels")
                if il_k and isinstance(dk, dict) and dk:
                    return max(il_k[n]["avg_us"] * il_k[n]["launches_per_step"]
                               for n in dk if n in il_k) if any(n in il_k for n in dk) else 0.0
                return r["avg_launch_ms"] * 1e3 * r["launches_per_step"]

            dom = max(rows, key=row_weight)
            out["roofline"] = dom
            dk = dom.get("device_kernels")
            if il_k and isinstance(dk, dict) and any(n in il_k for n in dk):
                kn = max((n for n in dk if n in il_k),
                         key=lambda n: il_k[n]["avg_us"] * il_k[n]["launches_per_step"])
                out["dominant_device_kernel"] = {
                    "name": kn[:100], "avg_us_in_loop": round(il_k[kn]["avg_us"], 2),
                    "launches_per_step": round(il_k[kn]["launches_per_step"], 2)}
            out["roofline_replay_gather"] = rows[0]
            out["roofline_replay_add"] = rows[1]
            out["roofline_all"] = [{k: v for k, v in r.items() if k != "peak_note"} for r in rows]
            out["kernel_time_sum_ms"] = tot
            if inloop is not None:
                out["inloop"] = {k: v for k, v in inloop.items() if k not in ("ops", "kernels")}
                if "kernels" in inloop:
                    top = sorted(inloop["kernels"].items(),
                                 key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches_per_step"])
                    out["inloop"]["kernels"] = [
                        {"name": n_[:96], "launches_per_step": round(k["launches_per_step"], 3),
                         "avg_us": round(k["avg_us"], 2)} for n_, k in top[:40]]
        if world == 1 and not args.no_other_configs:
            # BASELINE.json configs[2] / configs[4] at one GPU, observed by whoever runs this
            # command (short runs: 2 PPO iterations, 150 SAC iterations; own roofline + cpu_baseline)
            out["other_configs"] = {"ppo": run_other_config("ppo", 2),
                                    "sac": run_other_config("sac", 150)}
        if not args.no_cpu_baseline and world == 1:
            # torch-CPU convolutions at batch 256 stop scaling (and collapse when every hardware
            # thread of a 256-core host is used): probe a few thread counts, keep the fastest
            ncpu = max(1, os.cpu_count() or 1)
            cands = sorted({min(ncpu, t) for t in (8, 16, 32, 64)})
            probe = {t: cpu_baseline(S, 2, t)[1] for t in cands}
            threads = min(probe, key=probe.get)
            v, spstep = cpu_baseline(S, args.cpu_steps, threads)
            out["cpu_baseline"] = {
                "value": v, "unit": "samples/s", "cores": threads, "kind": "port",
                "learner_steps_per_sec": v / S,
                "sample": f"{args.cpu_steps} iterations (collect 256 envs + sample 256x2 + train "
                          f"batch 256) of the numpy/torch-CPU oracle on {threads} of {ncpu} host "
                          f"threads (fastest of {cands}), {spstep:.3f} s/iteration, replay ring "
                          f"shortened to {CPU_RING_FRAMES} frames/env (1.85 GB)"}
        emit(out, args.detail_out)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
