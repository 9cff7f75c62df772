"""Upper bound of a DUAL forward launch (online + target network of the DQN step in one launch per
layer) before building it: the Atari Q-network forward on ONE batch of 512 (what a dual launch
costs, give or take the second weight set) against two batch-256 forwards forked on two streams
(what the train step does now), each captured in a HIP graph and timed with HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import bench
    from agents_amd import ops
    from agents_amd.networks import layers as L
    from agents_amd.networks import sequential
    from agents_amd.utils import graph

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    net_a = sequential.Sequential(bench.atari_layers(L, bench.NUM_ACTIONS), seed=2)
    net_b = sequential.Sequential(bench.atari_layers(L, bench.NUM_ACTIONS), seed=3)
    g = torch.Generator().manual_seed(0)
    x256a = torch.randint(0, 256, (256, 84, 84, 4), dtype=torch.uint8, generator=g).to(dev)
    x256b = torch.randint(0, 256, (256, 84, 84, 4), dtype=torch.uint8, generator=g).to(dev)
    x512 = torch.cat([x256a, x256b], 0)
    from agents_amd.specs import tensor_spec
    spec = tensor_spec.TensorSpec(tuple(x256a.shape[1:]), torch.uint8, "observation")
    for n in (net_a, net_b):
        n.create_variables(spec, device=dev)
    side = ops.new_side_stream(dev)

    def forked():
        main = torch.cuda.current_stream(dev)
        net_b.prepare_forward(256, slot="train")
        side.wait_stream(main)
        with ops.side_line(side):
            net_b.forward(x256b, slot="train")
        net_a.forward(x256a, slot="train", need_grad=True)
        main.wait_stream(side)

    def serial():
        net_b.forward(x256b, slot="train")
        net_a.forward(x256a, slot="train", need_grad=True)

    def big():
        net_a.forward(x512, slot="big", need_grad=True)

    def timeit(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        c = graph._Captured()
        c.capture(lambda: [fn() for _ in range(reps)] and None)
        best = 1e9
        for _ in range(5):
            c.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            c.replay()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / reps)
        return best * 1e3

    print(f"two batch-256 forwards, forked on two streams : {timeit(forked):7.1f} us")
    print(f"two batch-256 forwards, one stream            : {timeit(serial):7.1f} us")
    print(f"one batch-512 forward (dual-launch upper bound): {timeit(big):7.1f} us")
    print(f"one batch-256 forward                          : "
          f"{timeit(lambda: net_a.forward(x256a, slot='train', need_grad=True)):7.1f} us")


if __name__ == "__main__":
    main()
