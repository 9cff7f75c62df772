"""Train-graph time under graph-structure knobs of Sequential.backward."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from agents_amd.networks import sequential

def run(**kw):
    for k, v in kw.items():
        setattr(sequential, k, v)
    dev = torch.device("cuda", 0)
    w = bench.build_workload(dev, 0, 1, 256, 64, 256, seed=1)
    w["init_driver"]._num_steps = 256 * 64
    w["init_driver"].run()
    lrn = w["learner"]
    exp, _ = w["rb"].get_next(256, 2)
    for _ in range(10):
        lrn._train_fn(exp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        lrn._train_fn(exp)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 300 * 1e6

for kw in (dict(SMALL_HEAD_ON_MAIN=True), dict(SMALL_HEAD_ON_MAIN=False), dict(SMALL_HEAD_ON_MAIN=True)):
    print(kw, f"train graphs {run(**kw):7.1f} us")
