"""Where does one bench iteration go?  Times the pieces separately (host wall clock around a
synchronised loop): full step, train graphs only, collect only, get_next only."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

def timed(fn, n=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    w = bench.build_workload(dev, 0, 1, 256, 64, 256, seed=1)
    w["init_driver"]._num_steps = 256 * 64
    w["init_driver"].run()
    it = iter(w["dataset"])
    lrn, drv = w["learner"], w["collect_driver"]
    state = {"ts": None}
    def full():
        state["ts"], _ = drv.run(state["ts"])
        lrn.run(iterations=1, iterator=it)
    print("full step (eager drv)%8.1f us" % timed(full))
    from agents_amd.utils import common
    grun = common.function(drv.run)
    def fullg():
        state["ts"], _ = grun(state["ts"])
        lrn.run(iterations=1, iterator=it)
    print("full step (graphs)   %8.1f us" % timed(fullg, n=200, warm=40))
    from agents_amd.utils import graph
    graph.enable_overlap(dev)
    print("full step (overlap)  %8.1f us" % timed(fullg, n=200, warm=40))
    graph.disable_overlap()
    def collectg():
        state["ts"], _ = grun(state["ts"])
    print("collect (graph)      %8.1f us" % timed(collectg))
    exp, _ = w["rb"].get_next(256, 2)
    print("train graphs only    %8.1f us" % timed(lambda: lrn._train_fn(exp)))
    def collect():
        state["ts"], _ = drv.run(state["ts"])
    print("collect (sync each)  %8.1f us" % timed(collect))
    print("get_next eager       %8.1f us" % timed(lambda: w["rb"].get_next(256, 2)))
    print("next(iterator)       %8.1f us" % timed(lambda: next(it)))
    def lrn_only():
        lrn.run(iterations=1, iterator=it)
    print("learner.run          %8.1f us" % timed(lrn_only))
    pol = w["agent"].collect_policy
    ts0 = state["ts"]
    print("policy.action        %8.1f us" % timed(lambda: pol.action(ts0)))
    env = w["env"]
    a = pol.action(ts0).action
    print("env.step             %8.1f us" % timed(lambda: env.step(a)))
main()
