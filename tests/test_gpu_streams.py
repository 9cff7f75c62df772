"""GPU: the fork/join structure of Sequential.backward.  The weight-gradient GEMMs run on a side
stream next to the input-gradient chain; if a fork misses the producer of a layer's dZ the
side-stream kernel reads a buffer that is still being written.  Graph-vs-eager comparisons cannot
see that (both share the code), so this compares the two-stream backward with the single-stream one
on the Atari network at batch 256, where conv1's weight gradient would otherwise start ~20 us
before conv2's input gradient has finished."""
import pytest
import torch

from agents_amd import ops
from agents_amd.networks import layers as L
from agents_amd.networks import sequential
from agents_amd.specs import tensor_spec

pytestmark = pytest.mark.gpu


def atari_net():
    vs = lambda: L.VarianceScaling(2.0)
    net = sequential.Sequential(
        [L.Rescale(255.0), L.Conv2D(32, (8, 8), 4, "relu", kernel_initializer=vs()),
         L.Conv2D(64, (4, 4), 2, "relu", kernel_initializer=vs()),
         L.Conv2D(64, (3, 3), 1, "relu", kernel_initializer=vs()), L.Flatten(),
         L.Dense(512, "relu", kernel_initializer=vs()), L.Dense(6, None, kernel_initializer=vs())],
        input_spec=tensor_spec.TensorSpec((84, 84, 4), torch.uint8), seed=3)
    net.create_variables()
    return net


@pytest.mark.parametrize("dx_first", [True, False])
def test_two_stream_backward_equals_single_stream(dev, dx_first, monkeypatch):
    old = sequential.DX_FIRST
    sequential.DX_FIRST = dx_first
    # every input-gradient launch is preceded by ~100 us of spinning on ITS stream: a weight
    # gradient that does not wait for the producer of its dZ now certainly runs too early
    for name in ("conv_dx", "dense_dx"):
        real = getattr(ops, name)

        def slow(*a, _real=real, **k):
            torch.cuda._sleep(200000)
            return _real(*a, **k)
        monkeypatch.setattr(ops, name, slow)
    try:
        net = atari_net()
        side = ops.new_side_stream(dev)
        g = torch.Generator().manual_seed(0)
        obs = torch.randint(0, 256, (256, 84, 84, 4), dtype=torch.uint8, generator=g).to(dev)
        k = net.dense_tail_start()
        for rep in range(4):
            # fresh upstream gradients every time: a kernel that ran early would pick up the
            # PREVIOUS repetition's intermediate buffers and give a different answer
            dq = torch.randn(256, 6, generator=g).to(dev)
            net.forward(obs, slot="t", need_grad=True)
            net.backward(dq, slot="t", side_stream=None)
            torch.cuda.synchronize()
            ref = [g_.clone() for g_ in net.gradients]   # (alignment padding is never written)
            dq2 = torch.randn(256, 6, generator=g).to(dev)   # scramble the intermediates
            net.forward(obs, slot="t", need_grad=True)
            net.backward(dq2, slot="t", side_stream=None)
            net.flat_grads.fill_(float("nan"))
            net.forward(obs, slot="t", need_grad=True)
            if rep % 2 == 0:
                net.backward(dq, slot="t", side_stream=side)
            else:   # partial backward + resume (the data-parallel bucket mode)
                net.backward(dq, slot="t", side_stream=side, stop_layer=k)
                net.backward_resume(256, slot="t", side_stream=side, from_layer=k)
            torch.cuda.synchronize()
            assert all(torch.equal(a, b) for a, b in zip(net.gradients, ref)), f"repetition {rep}"
    finally:
        sequential.DX_FIRST = old
