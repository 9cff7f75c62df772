#!/usr/bin/env python
"""In-kernel timeline of the fused PPO step (csrc/ppo_fused.hip): wall_clock64 stamps at the phase
boundaries of every workgroup, averaged.   python tools/ppo_fused_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import bench_ppo
    from agents_amd import _lib
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    w = bench_ppo.build(dev, 2048, 128, 4096, epochs=1)
    w["collect_driver"].run()
    lrn = w["learner"]
    lrn.run()                              # warm-up, graphs
    lrn._train_iter = lrn._norm_iter = None
    n_wg = 4096 // 16
    buf = torch.zeros((n_wg, 32), dtype=torch.int64, device=dev)
    lib = _lib.load()
    # stamps only show up in eager launches (the captured graphs hold the old kernel arguments)
    agent = w["agent"]
    from agents_amd.utils import graph
    lib.aa_ppo_fused_debug_stamps(buf.data_ptr())
    gt = graph.graphed_train(agent)
    gt.enabled = False
    lrn._generic_learner._train_fn = agent.train
    lrn.run()
    torch.cuda.synchronize()
    lib.aa_ppo_fused_debug_stamps(None)
    t = buf.cpu().double() * 0.01          # us
    names = {0: "start", 1: "adv moments", 2: "scalars + obs tile", 3: "fwd layer 0 (both nets)",
             4: "fwd layer 1", 5: "fwd layer 2", 9: "fwd done", 10: "loss", 11: "bwd top layer",
             12: "bwd middle layer", 13: "bwd first layer", 17: "end"}
    t0 = t[:, 0:1]
    rel = (t - t0)
    print("phase                 mean us since start   delta")
    prev = 0.0
    for i, n in sorted(names.items()):
        m = float(rel[:, i].mean())
        print(f"{n:22s} {m:10.2f} {m - prev:10.2f}")
        prev = m
    starts = t[:, 0] - t[:, 0].min()
    print(f"workgroup start spread: {float(starts.max()):.2f} us; last end - first start: "
          f"{float((t[:, 17].max() - t[:, 0].min())):.2f} us")


if __name__ == "__main__":
    main()
