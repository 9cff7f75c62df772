#!/usr/bin/env python
"""Where PPOLearner.run() spends HOST time at configs[2] (cProfile + wall clock per phase)."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import bench_ppo
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    w = bench_ppo.build(dev, 2048, 128, 4096, epochs=10)
    lrn, rb, drv = w["learner"], w["rb"], w["collect_driver"]

    def collect():
        rb.clear()
        drv.run()
        torch.cuda.synchronize()
        lrn._train_iter = lrn._norm_iter = None

    collect()
    lrn.run()
    torch.cuda.synchronize()
    collect()
    # phase timers (each followed by a device sync: the GPU time of the phase is inside)
    t0 = time.perf_counter()
    frames = lrn._update_normalizers()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    samples = lrn._take_samples()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    li = lrn._run_fused(samples, (frames // 4096) * 10)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    lrn._generic_learner.finish_run(li)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"update_normalizers {1e3 * (t1 - t0):.2f} ms, take_samples {1e3 * (t2 - t1):.2f} ms, "
          f"fused epochs {1e3 * (t3 - t2):.2f} ms ({(frames // 4096) * 10} steps), finish "
          f"{1e3 * (t4 - t3):.2f} ms")
    collect()
    pr = cProfile.Profile()
    pr.enable()
    lrn.run()
    torch.cuda.synchronize()
    pr.disable()
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(25)
    print(buf.getvalue()[:6000])


if __name__ == "__main__":
    main()
