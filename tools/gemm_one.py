#!/usr/bin/env python
"""Runs one contraction of the DQN-Atari step N times (for rocprofv3 --pmc / --kernel-trace).
  python tools/gemm_one.py conv2.fwd [--reps 20] [--cfg 0] [--splits 0] [--no-dma]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from agents_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cfg", type=int, default=0)
    ap.add_argument("--splits", type=int, default=0)
    ap.add_argument("--no-dma", action="store_true")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--no-time", action="store_true", help="skip the graph timing (profiler runs)")
    args = ap.parse_args()
    ops.FORCE_NO_DMA = args.no_dma
    S = args.batch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    c, s = args.cfg, args.splits
    if args.name == "conv1.fwd":
        obs = torch.randint(0, 256, (S, 84, 84, 4), dtype=torch.uint8, generator=g).to(dev)
        w, b, y = r(8, 8, 4, 32), r(32), r(S, 20, 20, 32)
        fn = lambda: ops.conv_forward(obs, w, b, 4, "relu", y, a_div=255.0, force_cfg=c,
                                      force_splits=s)
    elif args.name == "conv2.fwd":
        x, w, b, y = r(S, 20, 20, 32), r(4, 4, 32, 64), r(64), r(S, 9, 9, 64)
        fn = lambda: ops.conv_forward(x, w, b, 2, "relu", y, force_cfg=c, force_splits=s)
    elif args.name == "conv3.fwd":
        x, w, b, y = r(S, 9, 9, 64), r(3, 3, 64, 64), r(64), r(S, 7, 7, 64)
        fn = lambda: ops.conv_forward(x, w, b, 1, "relu", y, force_cfg=c, force_splits=s)
    elif args.name == "conv23.fwd":
        x, w2, b2, y2 = r(S, 20, 20, 32), r(4, 4, 32, 64), r(64), r(S, 9, 9, 64)
        w3, b3, y3 = r(3, 3, 64, 64), r(64), r(S, 7, 7, 64)
        fn = lambda: ops.conv_pair_forward(x, w2, b2, 2, "relu", y2, w3, b3, 1, "relu", y3)
    elif args.name in ("conv2.dX", "conv3.dX"):
        if args.name == "conv2.dX":
            xs, wsh, strd, npix = (S, 20, 20, 32), (4, 4, 32, 64), 2, 81
        else:
            xs, wsh, strd, npix = (S, 9, 9, 64), (3, 3, 64, 64), 1, 49
        y, w, dz, dx = r(*xs), r(*wsh), r(S * npix, 64), r(*xs)
        dcol = torch.empty(S * npix * wsh[0] * wsh[1] * wsh[2], device=dev)
        ops.CONV_DX_FRAME = not args.no_dma      # --no-dma selects the GEMM + col2im path here
        fn = lambda: ops.conv_dx(dz, w, xs, strd, dcol, dx, mask_src=y, mask_act="relu")
    elif args.name == "fc1.fwd":
        x, w, b, y = r(S, 3136), r(3136, 512), r(512), r(S, 512)
        w2, b2, q = r(512, 6), r(6), r(S, 6)
        if c or s:
            fn = lambda: ops.dense_forward(x, w, b, "relu", y, force_cfg=c, force_splits=s)
        else:   # what the network runs: fc1's main loop + the head summing its split-K slabs
            fn = lambda: ops.dense_tail_forward(x, w, b, "relu", y, w2, b2, None, q)
    elif args.name == "fc1.dW":
        x, dz, gk = r(S, 3136), r(S, 512), r(3136, 512)
        fn = lambda: ops.dense_dw(x, dz, gk, force_cfg=c, force_splits=s)
    elif args.name in ("conv2.dW", "conv3.dW"):
        if args.name == "conv2.dW":
            xs, wsh, strd, npix = (S, 20, 20, 32), (4, 4, 32, 64), 2, 81
        else:
            xs, wsh, strd, npix = (S, 9, 9, 64), (3, 3, 64, 64), 1, 49
        x, dz, gk, gb = torch.relu(r(*xs)), r(S * npix, 64), r(*wsh), r(64)
        fn = lambda: ops.conv_dw(x, dz, wsh, strd, gk, a_div=1.0, force_cfg=c, force_splits=s,
                                 bias_grad=gb)
    elif args.name == "conv1.dW":
        obs = torch.randint(0, 256, (S, 84, 84, 4), dtype=torch.uint8, generator=g).to(dev)
        dz, gk, gb = r(S * 400, 32), r(8, 8, 4, 32), r(32)
        fn = lambda: ops.conv_dw(obs, dz, (8, 8, 4, 32), 4, gk, a_div=255.0, force_cfg=c,
                                 force_splits=s, bias_grad=gb)
    elif args.name == "fc1.dX":
        x, w, dz, dx = r(S, 3136), r(3136, 512), r(S, 512), r(S, 3136)
        fn = lambda: ops.dense_dx(dz, w, dx, mask_src=x, mask_act="relu", force_cfg=c,
                                  force_splits=s)
    elif args.name == "replay.get_next":
        # the fused draw + gather launch of TFUniformReplayBuffer.get_next on Atari rows
        from agents_amd.replay_buffers import tf_uniform_replay_buffer as rb_lib
        from agents_amd.specs import tensor_spec
        from agents_amd.trajectories import trajectory
        spec = trajectory.Trajectory(
            step_type=tensor_spec.TensorSpec((), torch.int32),
            observation=tensor_spec.TensorSpec((84, 84, 4), torch.uint8),
            action=tensor_spec.TensorSpec((), torch.int64), policy_info=(),
            next_step_type=tensor_spec.TensorSpec((), torch.int32),
            reward=tensor_spec.TensorSpec((), torch.float32),
            discount=tensor_spec.TensorSpec((), torch.float32))
        rb = rb_lib.TFUniformReplayBuffer(spec, batch_size=256, max_length=256, device=dev)
        for v in rb._data_table.variables():
            if v.dtype == torch.uint8:
                v.random_(0, 256)
        rb._last_id.fill_(255)
        rb._last_id_host = 255
        if rb.supports_stamped_draws() and os.environ.get("AA_PMC_DEVICE_DRAW") != "1":
            # what a graphed dataset runs: the stamped launch, rows drawn by the host library
            # (output buffers allocated by hand: a get_next here would put the device-draw kernel
            # into the trace as well)
            from agents_amd.utils import nest_utils
            flat = nest_utils.flatten(spec)
            outs = [torch.empty((S, 2) + tuple(sp.shape), dtype=sp.dtype, device=dev)
                    for sp in flat]
            info = rb_lib.BufferInfo(ids=torch.empty((S, 2), dtype=torch.int64, device=dev),
                                     probabilities=torch.empty((S,), dtype=torch.float32,
                                                               device=dev))
            stamped = rb.stamped_slot((nest_utils.pack_sequence_as(spec, outs), info))
            fn = lambda: rb.draw_into(stamped)
        else:
            fn = lambda: rb.get_next(S, 2)
    elif args.name == "replay.gather":
        # 512 random rows of the Atari trajectory table (28,248 B per row), 4096-row table
        from agents_amd.replay_buffers import table
        from agents_amd.specs import tensor_spec
        specs = [tensor_spec.TensorSpec((), torch.int32), tensor_spec.TensorSpec((84, 84, 4), torch.uint8),
                 tensor_spec.TensorSpec((), torch.int64), tensor_spec.TensorSpec((), torch.int32),
                 tensor_spec.TensorSpec((), torch.float32), tensor_spec.TensorSpec((), torch.float32)]
        tab = table.Table(specs, 65536, device=dev)
        for v in tab.variables():
            if v.dtype == torch.uint8:
                v.random_(0, 256)
        idt = torch.arange(65536, dtype=torch.int64, device=dev)
        rows = torch.randint(0, 65536, (256, 2), generator=g).to(dev)
        ids = torch.empty((256, 2), dtype=torch.int64, device=dev)
        fn = lambda: tab.read(rows, idt, ids)
    else:
        raise SystemExit("unknown case")
    for _ in range(args.reps):
        fn()
    torch.cuda.synchronize()
    if args.no_time:
        return
    # graph-timed (what the launch costs inside the captured step)
    g2 = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        with torch.cuda.graph(g2, stream=st):
            for _ in range(20):
                fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record(); g2.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{args.name} B={S} cfg={c} splits={s}: {best:.2f} us per launch (graph of 20)")


if __name__ == "__main__":
    main()
