import cProfile, pstats, sys, time, io
sys.path.insert(0, ".")
import torch, bench
from agents_amd.utils import common
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
w = bench.build_workload(dev, 0, 1, 256, 3906, 256, seed=1, replay="prioritized")
w["init_driver"]._num_steps = 256 * 3906
w["init_driver"].run(); torch.cuda.synchronize()
rb, lrn = w["rb"], w["learner"]
it = iter(w["dataset"])
collect_run = common.function(w["collect_driver"].run)
ts = None
def step():
    global ts
    ts, _ = collect_run(ts)
    return lrn.run(iterations=1, iterator=it)
for _ in range(400): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue ms/step", (t1 - t0) / 300 * 1e3, "incl drain", (t2 - t0) / 300 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
