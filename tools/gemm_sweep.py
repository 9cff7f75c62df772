#!/usr/bin/env python
"""Sweeps tile config x split-K for every contraction of the DQN-Atari train step (batch S) and
prints the time of each variant (HIP events on torch's current stream, which the kernels use).
Development tool for the plan heuristics in csrc/gemm.hip; needs a GPU.

  python tools/gemm_sweep.py [--batch 256] [--reps 30] [--only conv2]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from agents_amd import ops  # noqa: E402


def timeit(fn, reps):
    """GPU time per call: `reps` calls are captured into one HIP graph and the replay is timed, so
    the Python / ctypes launch cost (10-20 us per op) does not hide the kernel time."""
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (2 * reps) * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--only", default="")
    ap.add_argument("--cfgs", default="0,1,2,3,4,5,6,7")
    ap.add_argument("--splits", default="0,1,2,3,4,6,8,12,16,24,32")
    ap.add_argument("--no-dma", action="store_true", help="register-staged main loop")
    args = ap.parse_args()
    ops.FORCE_NO_DMA = args.no_dma
    S = args.batch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    obs = torch.randint(0, 256, (S, 84, 84, 4), dtype=torch.uint8, generator=g).to(dev)
    w1, w2, w3 = r(8, 8, 4, 32), r(4, 4, 32, 64), r(3, 3, 64, 64)
    w4, w5 = r(3136, 512), r(512, 6)
    b1, b2, b3, b4, b5 = r(32), r(64), r(64), r(512), r(6)
    y1, y2, y3 = r(S, 20, 20, 32), r(S, 9, 9, 64), r(S, 7, 7, 64)
    y4, y5 = r(S, 512), r(S, 6)
    dz1, dz2, dz3 = r(S * 400, 32), r(S * 81, 64), r(S * 49, 64)
    dz4, dz5 = r(S, 512), r(S, 6)
    g1, g2, g3 = torch.empty_like(w1), torch.empty_like(w2), torch.empty_like(w3)
    g4, g5 = torch.empty_like(w4), torch.empty_like(w5)
    dx4, dx5 = r(S, 3136), r(S, 512)
    dcol = torch.empty(S * 81 * 512, device=dev)
    x3 = y3.view(S, -1)
    M = 1e6
    cases = {
        "conv1.fwd": (lambda c, s: ops.conv_forward(obs, w1, b1, 4, "relu", y1, a_div=255.0,
                                                    force_cfg=c, force_splits=s),
                      2 * S * 400 * 32 * 256 / M),
        "conv2.fwd": (lambda c, s: ops.conv_forward(y1, w2, b2, 2, "relu", y2, force_cfg=c,
                                                    force_splits=s), 2 * S * 81 * 64 * 512 / M),
        "conv3.fwd": (lambda c, s: ops.conv_forward(y2, w3, b3, 1, "relu", y3, force_cfg=c,
                                                    force_splits=s), 2 * S * 49 * 64 * 576 / M),
        "fc1.fwd": (lambda c, s: ops.dense_forward(x3, w4, b4, "relu", y4, force_cfg=c,
                                                   force_splits=s), 2 * S * 3136 * 512 / M),
        "fc2.fwd": (lambda c, s: ops.dense_forward(y4, w5, b5, None, y5, force_cfg=c,
                                                   force_splits=s), 2 * S * 512 * 6 / M),
        "fc1.dW": (lambda c, s: ops.dense_dw(x3, dz4, g4, force_cfg=c, force_splits=s),
                   2 * S * 3136 * 512 / M),
        "fc1.dX": (lambda c, s: ops.dense_dx(dz4, w4, dx4, mask_src=x3, mask_act="relu",
                                             force_cfg=c, force_splits=s),
                   2 * S * 3136 * 512 / M),
        "fc2.dW": (lambda c, s: ops.dense_dw(y4, dz5, g5, force_cfg=c, force_splits=s),
                   2 * S * 512 * 6 / M),
        "fc2.dX": (lambda c, s: ops.dense_dx(dz5, w5, dx5, mask_src=y4, mask_act="relu",
                                             force_cfg=c, force_splits=s), 2 * S * 512 * 6 / M),
        "conv3.dW": (lambda c, s: ops.conv_dw(y2, dz3, (3, 3, 64, 64), 1, g3, force_cfg=c,
                                              force_splits=s), 2 * S * 49 * 64 * 576 / M),
        "conv2.dW": (lambda c, s: ops.conv_dw(y1, dz2, (4, 4, 32, 64), 2, g2, force_cfg=c,
                                              force_splits=s), 2 * S * 81 * 64 * 512 / M),
        "conv1.dW": (lambda c, s: ops.conv_dw(obs, dz1, (8, 8, 4, 32), 4, g1, a_div=255.0,
                                              force_cfg=c, force_splits=s),
                     2 * S * 400 * 32 * 256 / M),
        "conv3.dcol": (lambda c, s: ops.dense_dx(dz3, w3.view(576, 64), dcol[:S * 49 * 576].view(
            S * 49, 576), force_cfg=c, force_splits=s), 2 * S * 49 * 64 * 576 / M),
        "conv2.dcol": (lambda c, s: ops.dense_dx(dz2, w2.view(512, 64), dcol.view(S * 81, 512),
                                                 force_cfg=c, force_splits=s),
                       2 * S * 81 * 64 * 512 / M),
    }
    cfgs = [int(x) for x in args.cfgs.split(",")]
    splits = [int(x) for x in args.splits.split(",")]
    for name, (fn, mflop) in cases.items():
        if args.only and args.only not in name:
            continue
        res = []
        for c in cfgs:
            for s in splits:
                if (c == 0) != (s == 0):
                    continue
                try:
                    us = timeit(lambda: fn(c, s), args.reps)
                except Exception as e:  # invalid combination for this shape
                    continue
                res.append((us, c, s))
        if not res:
            continue
        auto = [x for x in res if x[1] == 0]
        res.sort()
        best = res[0]
        line = f"{name:11s} auto {auto[0][0]:7.1f} us | best cfg={best[1]} splits={best[2]} " \
               f"{best[0]:7.1f} us ({mflop / best[0]:6.1f} TF) |"
        for us, c, s in res[:6]:
            line += f" c{c}s{s}:{us:.1f}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
