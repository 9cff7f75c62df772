"""Which objects of a graphed DQN / PPO / SAC loop are only released by Python's cyclic collector?

Builds a small loop (driver + replay + dataset + Learner through the HIP graphs), runs it until
every graph is recorded, drops every name with the collector DISABLED and reports:
  * which of the tracked objects (agent, networks, driver, replay buffer, GraphedTrain,
    GraphedDriverRun, GraphedSampler, every torch CUDAGraph) are still alive,
  * the types the cyclic collector then finds (gc.DEBUG_SAVEALL), i.e. the members of the cycles,
  * device memory (hipMemGetInfo) before / after.

    python tools/lifetime_probe.py [dqn|ppo|sac] [rounds]
"""
import collections
import gc
import os
import sys
import weakref

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from agents_amd import optimizers  # noqa: E402
from agents_amd.agents.dqn import dqn_agent  # noqa: E402
from agents_amd.drivers import dynamic_step_driver  # noqa: E402
from agents_amd.environments import random_tf_environment  # noqa: E402
from agents_amd.networks import layers as L  # noqa: E402
from agents_amd.networks import sequential  # noqa: E402
from agents_amd.replay_buffers import tf_uniform_replay_buffer as rb_lib  # noqa: E402
from agents_amd.specs import tensor_spec  # noqa: E402
from agents_amd.train import learner  # noqa: E402
from agents_amd.trajectories import time_step as ts  # noqa: E402
from agents_amd.utils import common, graph  # noqa: E402

A = 4


def dqn_loop(dev, iters=8, overlap=True):
    B = 8
    obs_spec = tensor_spec.TensorSpec((12, 12, 4), torch.uint8, "observation")
    aspec = tensor_spec.BoundedTensorSpec((), torch.int64, 0, A - 1, "action")
    tss = ts.time_step_spec(obs_spec)
    env = random_tf_environment.RandomTFEnvironment(tss, aspec, batch_size=B,
                                                    episode_end_probability=0.1, seed=11,
                                                    device=dev)
    net = sequential.Sequential([L.Rescale(255.0), L.Conv2D(8, 4, 4, "relu"), L.Flatten(),
                                 L.Dense(32, "relu"), L.Dense(A)], seed=3)
    agent = dqn_agent.DqnAgent(tss, aspec, q_network=net, optimizer=optimizers.Adam(1e-3),
                               td_errors_loss_fn=common.element_wise_huber_loss, gamma=0.9,
                               epsilon_greedy=0.3, target_update_period=3, seed=5)
    agent.initialize()
    rb = rb_lib.TFUniformReplayBuffer(agent.collect_data_spec, batch_size=B, max_length=64,
                                      device=dev, seed=9)
    drv = dynamic_step_driver.DynamicStepDriver(env, agent.collect_policy,
                                                observers=[rb.add_batch], num_steps=B)
    if overlap:
        graph.enable_overlap(dev)
    run = common.function(drv.run)
    t = None
    for _ in range(4):
        t, _ = run(t)
    ds = rb.as_dataset(sample_batch_size=16, num_steps=2).prefetch(3)
    it = iter(ds)
    import tempfile
    lrn = learner.Learner(tempfile.mkdtemp(), common.Variable(0), agent,
                          experience_dataset_fn=None, checkpoint_interval=10 ** 9)
    for _ in range(iters):
        t, _ = run(t)
        lrn.run(iterations=1, iterator=it)
    torch.cuda.synchronize()
    tracked = {"agent": agent, "net": net, "env": env, "rb": rb, "driver": drv, "run": run,
               "learner": lrn, "graphed_train": getattr(agent, "_graphed_train", None),
               "iterator": it, "dataset": ds}
    return {k: weakref.ref(v) for k, v in tracked.items() if v is not None}


def _refs(w):
    out = {}
    for k, v in w.items():
        try:
            out[k] = weakref.ref(v)
        except TypeError:
            pass
    return out


def sac_loop(dev, iters=8):
    import bench_sac
    w = bench_sac.build(dev, envs=16, max_length=16, batch=8)
    graph.enable_overlap(dev)
    it = iter(w["dataset"])
    t = None
    for _ in range(iters):
        t, _ = w["collect"](t)
        w["learner"].run(iterations=1, iterator=it)
    torch.cuda.synchronize()
    w["iterator"] = it
    w["graphed_train"] = getattr(w["agent"], "_graphed_train", None)
    return _refs(w)


def ppo_loop(dev, iters=4):
    import bench_ppo
    w = bench_ppo.build(dev, envs=16, steps=8, minibatch=32, epochs=2)
    lrn, rb = w["learner"], w["rb"]
    t = None
    for k in range(iters):
        rb.clear()
        t, _ = (w["collect"] if k % 2 else w["collect_driver"].run)(t)
        lrn._train_iter = lrn._norm_iter = None
        lrn.run()
    torch.cuda.synchronize()
    w["graphed_train"] = getattr(w["agent"], "_graphed_train", None)
    w.pop("raw_dataset_fn", None)
    return _refs(w)


def main():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    kind = sys.argv[1] if len(sys.argv) > 1 else "dqn"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    build = {"dqn": dqn_loop, "sac": sac_loop, "ppo": ppo_loop}[kind]
    n_graphs = lambda: sum(1 for o in gc.get_objects() if isinstance(o, torch.cuda.CUDAGraph))
    build(dev)          # first use: module-level caches, workspaces
    gc.collect()
    torch.cuda.synchronize()
    gc.disable()
    free0 = torch.cuda.mem_get_info()[0]
    g0 = n_graphs()
    for r in range(rounds):
        refs = build(dev)
        torch.cuda.synchronize()
        alive = sorted(k for k, w in refs.items() if w() is not None)
        if "agent" in alive and r == 0:
            for ref in gc.get_referrers(refs["agent"]()):
                d = type(ref).__module__ + "." + type(ref).__qualname__
                if isinstance(ref, dict):
                    d += " keys=" + str(list(ref)[:10])
                    for r2 in gc.get_referrers(ref):
                        d += "\n            <- " + type(r2).__module__ + "." + type(r2).__qualname__
                        if isinstance(r2, dict):
                            d += " keys=" + str(list(r2)[:10])
                elif hasattr(ref, "__func__"):
                    d += " method " + ref.__func__.__qualname__
                    for r2 in gc.get_referrers(ref):
                        d += "\n            <- " + type(r2).__module__ + "." + type(r2).__qualname__
                        if isinstance(r2, dict):
                            d += " keys=" + str(list(r2)[:10])
                        if isinstance(r2, (list, tuple)):
                            d += " len=" + str(len(r2))
                            for r3 in gc.get_referrers(r2):
                                d += "\n                  <- " + type(r3).__module__ + "." + type(r3).__qualname__ + (" keys=" + str(list(r3)[:10]) if isinstance(r3, dict) else "")
                elif type(ref).__name__ == "frame":
                    d += f" {ref.f_code.co_filename}:{ref.f_lineno} {ref.f_code.co_name}"
                print("    referrer of the live agent:", d)
        free1 = torch.cuda.mem_get_info()[0]
        print(f"[round {r}] alive without the collector: {alive or 'none'};  CUDAGraph objects "
              f"{g0} -> {n_graphs()};  device free {free0 >> 20} -> {free1 >> 20} MiB;  torch "
              f"allocated {torch.cuda.memory_allocated() >> 10} KiB reserved "
              f"{torch.cuda.memory_reserved() >> 20} MiB;  live graphs {graph.live_graphs()}",
              flush=True)
    gc.set_debug(gc.DEBUG_SAVEALL)
    n = gc.collect()
    gc.set_debug(0)
    types = collections.Counter(type(o).__module__ + "." + type(o).__qualname__ for o in gc.garbage)
    print(f"cyclic collector found {n} objects; by type:")
    for t, c in types.most_common(40):
        print(f"   {c:6d}  {t}")
    fns = collections.Counter(o.__module__ + ":" + o.__qualname__ for o in gc.garbage
                              if type(o).__name__ == "function")
    print("garbage functions:")
    for t, c in fns.most_common(20):
        print(f"   {c:6d}  {t}")
    # who points at the objects that matter
    names = ("GraphedTrain", "GraphedDriverRun", "GraphedSampler", "DqnAgent", "Sequential",
             "TFUniformReplayBuffer", "DynamicStepDriver", "Learner", "_Captured", "SacAgent",
             "PPOClipAgent", "PPOLearner", "CriticNetwork", "ActorDistributionNetwork",
             "RandomTFEnvironment", "PPOPolicy", "SacPolicy", "Checkpointer")
    garbage_ids = {id(o) for o in gc.garbage}
    seen = set()
    for o in gc.garbage:
        tn = type(o).__qualname__
        if tn in names and tn not in seen:
            seen.add(tn)
            print(f"--- referrers (inside the garbage) of one {tn}:")
            for ref in gc.get_referrers(o):
                if id(ref) in garbage_ids and ref is not gc.garbage:
                    d = type(ref).__qualname__
                    extra = ""
                    if isinstance(ref, dict):
                        keys = [k for k, v in ref.items() if v is o]
                        own = [type(x).__qualname__ for x in gc.get_referrers(ref)
                               if id(x) in garbage_ids and hasattr(x, "__dict__") and
                               x.__dict__ is ref]
                        extra = f" keys={keys} owner={own}"
                    elif hasattr(ref, "__func__"):
                        extra = f" method {ref.__func__.__qualname__}"
                    elif type(ref).__name__ == "cell":
                        extra = " (closure cell)"
                    elif type(ref).__name__ == "function":
                        extra = f" {ref.__qualname__}"
                    print(f"      {d}{extra}")
    gc.garbage.clear()
    gc.collect()
    still = {k: w() for k, w in refs.items() if w() is not None}
    if still:
        print(f"STILL alive after a full collection (a strong reference from a live object): "
              f"{sorted(still)}")

        def describe(r):
            d = type(r).__module__ + "." + type(r).__qualname__
            if isinstance(r, dict):
                d += " keys=" + str([k for k in list(r)[:8]])
            elif hasattr(r, "__func__"):
                d += " method " + r.__func__.__qualname__
            elif type(r).__name__ == "function":
                d += " " + r.__qualname__
            elif type(r).__name__ == "frame":
                d += f" {r.f_code.co_filename}:{r.f_lineno} {r.f_code.co_name}"
            return d
        for k in ("agent", "graphed_train", "driver", "rb"):
            o = still.get(k)
            if o is None:
                continue
            print(f"--- referrers of {k}:")
            for r in gc.get_referrers(o):
                if r is still or type(r).__name__ == "frame" and r.f_code.co_name == "main":
                    continue
                print("     ", describe(r))
                for r2 in gc.get_referrers(r):
                    if r2 is still or type(r2).__name__ in ("frame", "list") and False:
                        continue
                    print("           <-", describe(r2)[:200])
    print(f"after collection: CUDAGraph objects {n_graphs()}, device free "
          f"{torch.cuda.mem_get_info()[0] >> 20} MiB")


if __name__ == "__main__":
    main()
