"""Two consecutive Conv2D layers in one launch (csrc/conv_pair.hip) against float64 and against the
layer-by-layer kernels; the Sequential integration (forward values and the backward pass that
consumes the middle activation)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from agents_amd import ops

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


def rnd(rng, *shape):
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


def act_ref(x, act):
    return torch.relu(x) if act == "relu" else torch.tanh(x) if act == "tanh" else x


def conv_ref(x, w, b, stride):
    y = F.conv2d(x.permute(0, 3, 1, 2), w.double().permute(3, 2, 0, 1),
                 None if b is None else b.double(), stride=stride)
    return y.permute(0, 2, 3, 1).contiguous()


def close(got, ref, tol=TOL):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape
    scale = max(ref.abs().max().item(), 1e-30)
    err = (got - ref).abs().max().item()
    assert err <= tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"


PAIRS = [  # (B, H, W, C, (KH1,KW1,s1,F1,act1), (KH2,KW2,s2,F2,act2))
    (8, 20, 20, 32, (4, 4, 2, 64, "relu"), (3, 3, 1, 64, "relu")),     # Atari conv2 -> conv3
    (300, 20, 20, 32, (4, 4, 2, 64, "relu"), (3, 3, 1, 64, "relu")),   # more frames than CUs
    (600, 9, 9, 32, (3, 3, 1, 32, "tanh"), (2, 2, 1, 32, None)),       # frames looped per workgroup
    (3, 12, 10, 32, (3, 2, 1, 64, None), (2, 3, 2, 48, "relu")),       # rectangular, three column tiles
    (2, 23, 23, 32, (5, 5, 2, 32, "relu"), (3, 3, 3, 16, "tanh")),     # 100 -> 9 pixels, stride 3
    (1, 4, 4, 32, (4, 4, 1, 32, None), (1, 1, 1, 16, None)),           # single pixel
]


@pytest.fixture(params=[True, False], ids=["bf16x6", "fp32mfma"])
def x6(request):
    """Both implementations of the pair: csrc/conv_pair_x6.hip (default where the shape fits) and
    csrc/conv_pair.hip."""
    prev = ops.CONV_PAIR_X6
    ops.CONV_PAIR_X6 = request.param
    ops._PAIR_X6_WS.clear()
    yield request.param
    ops.CONV_PAIR_X6 = prev
    ops._PAIR_X6_WS.clear()


def test_atari_pair_takes_the_bf16_kernel(dev):
    import ctypes
    from agents_amd import _lib
    d = [_lib.ConvLayerDesc(w=None, bias=None, y=None, KH=k, KW=k, stride=st, Cout=64, act=1)
         for k, st in ((4, 2), (3, 1))]
    lib = _lib.load()
    ws = lib.aa_conv_pair_x6_workspace_bytes(256, 20, 20, 32, ctypes.byref(d[0]), ctypes.byref(d[1]))
    assert ws == (16 * 4 + 18 * 4) * 3 * 64 * 16       # k-steps x column tiles x planes x 1 KiB
    assert ops.CONV_PAIR_X6


@pytest.mark.parametrize("cfg", PAIRS)
@pytest.mark.parametrize("bias", [True, False])
def test_conv_pair_forward(dev, cfg, bias, x6):
    B, H, W, C, (KH1, KW1, s1, F1, a1), (KH2, KW2, s2, F2, a2) = cfg
    rng = np.random.default_rng(B + H + C + F1 + F2)
    x = rnd(rng, B, H, W, C)
    w1, w2 = rnd(rng, KH1, KW1, C, F1) * 0.2, rnd(rng, KH2, KW2, F1, F2) * 0.2
    b1 = rnd(rng, F1) if bias else None
    b2 = rnd(rng, F2) if bias else None
    OH1, OW1 = ops.conv_out_hw(H, W, KH1, KW1, s1)
    OH2, OW2 = ops.conv_out_hw(OH1, OW1, KH2, KW2, s2)
    assert ops.conv_pair_supported((B, H, W, C), w1, s1, w2, s2)
    y1 = torch.full((B, OH1, OW1, F1), float("nan"), device=dev)
    y2 = torch.full((B, OH2, OW2, F2), float("nan"), device=dev)
    d = lambda t: None if t is None else t.to(dev)
    ops.conv_pair_forward(d(x), d(w1), d(b1), s1, a1, y1, d(w2), d(b2), s2, a2, y2)
    r1 = act_ref(conv_ref(x.double(), w1, b1, s1), a1)
    r2 = act_ref(conv_ref(r1, w2, b2, s2), a2)
    close(y1, r1)
    close(y2, r2)
    # the layer-by-layer kernels agree to the same tolerance
    z1 = torch.empty_like(y1)
    z2 = torch.empty_like(y2)
    ops.conv_forward(d(x), d(w1), d(b1), s1, a1, z1)
    ops.conv_forward(z1, d(w2), d(b2), s2, a2, z2)
    close(y1, z1.cpu())
    close(y2, z2.cpu(), tol=5e-5)


def test_conv_pair_x6_exact_on_bf16_representable_operands(dev):
    """Integer-valued operands below 2^8 are single bf16 pieces: every product and (for these
    magnitudes) every partial sum is exact in fp32, so the kernel must reproduce the float64
    reference bit for bit -- any fragment / K-order mismatch between the A and B sides shows."""
    rng = np.random.default_rng(11)
    x = torch.from_numpy(rng.integers(-8, 9, (5, 20, 20, 32)).astype(np.float32))
    w1 = torch.from_numpy(rng.integers(-3, 4, (4, 4, 32, 64)).astype(np.float32))
    w2 = torch.from_numpy(rng.integers(-2, 3, (3, 3, 64, 64)).astype(np.float32))
    b1 = torch.from_numpy(rng.integers(-5, 6, (64,)).astype(np.float32))
    y1 = torch.empty(5, 9, 9, 64, device=dev)
    y2 = torch.empty(5, 7, 7, 64, device=dev)
    ops.conv_pair_forward(x.to(dev), w1.to(dev), b1.to(dev), 2, "relu", y1, w2.to(dev), None, 1,
                          None, y2)
    r1 = torch.relu(conv_ref(x.double(), w1, b1, 2))
    r2 = conv_ref(r1, w2, None, 1)
    assert float(r2.abs().max()) < 2 ** 24
    assert torch.equal(y1.cpu().double(), r1) and torch.equal(y2.cpu().double(), r2)


def test_conv_pair_strided_batch_and_determinism(dev):
    rng = np.random.default_rng(4)
    x5 = rnd(rng, 6, 2, 20, 20, 32).to(dev)
    w1, w2 = (rnd(rng, 4, 4, 32, 64) * 0.1).to(dev), (rnd(rng, 3, 3, 64, 64) * 0.1).to(dev)
    b1, b2 = rnd(rng, 64).to(dev), rnd(rng, 64).to(dev)
    outs = []
    for _ in range(2):
        y1 = torch.empty(6, 9, 9, 64, device=dev)
        y2 = torch.empty(6, 7, 7, 64, device=dev)
        ops.conv_pair_forward(x5[:, 1], w1, b1, 2, "relu", y1, w2, b2, 1, "relu", y2)
        outs.append((y1.cpu(), y2.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    r1 = torch.relu(conv_ref(x5[:, 1].cpu().double(), w1.cpu(), b1.cpu(), 2))
    close(outs[0][1], torch.relu(conv_ref(r1, w2.cpu(), b2.cpu(), 1)))


def test_conv_pair_unsupported_shapes(dev):
    rng = np.random.default_rng(5)
    w1, w2 = rnd(rng, 8, 8, 4, 32), rnd(rng, 4, 4, 32, 64)
    assert not ops.conv_pair_supported((4, 84, 84, 4), w1, 4, w2, 2)       # 400 pixels per frame
    w3, w4 = rnd(rng, 3, 3, 16, 32), rnd(rng, 1, 1, 32, 16)
    assert not ops.conv_pair_supported((4, 8, 8, 16), w3, 1, w4, 1)        # Cin % 32 != 0
    w5 = rnd(rng, 1, 1, 32, 20)
    assert not ops.conv_pair_supported((4, 8, 8, 32), rnd(rng, 3, 3, 32, 32), 1, w5, 1)  # Cout % 16
    with pytest.raises(Exception):
        ops.conv_pair_forward(torch.zeros(4, 84, 84, 4, device=dev), w1.to(dev), None, 4, None,
                              torch.empty(4, 20, 20, 32, device=dev), w2.to(dev), None, 2, None,
                              torch.empty(4, 9, 9, 64, device=dev))


def test_sequential_uses_pair_and_backward_matches(dev):
    """The Atari Q-network forward/backward with the fused pair equals the layer-by-layer path
    (same kernels for everything else) to fp32 tolerance."""
    from agents_amd.networks import sequential, layers as L
    from agents_amd.specs import tensor_spec
    spec = tensor_spec.TensorSpec((84, 84, 4), torch.uint8)

    def build():
        net = sequential.Sequential([
            L.Rescale(255.0), L.Conv2D(32, 8, 4, activation="relu"),
            L.Conv2D(64, 4, 2, activation="relu"), L.Conv2D(64, 3, 1, activation="relu"),
            L.Flatten(), L.Dense(512, activation="relu"), L.Dense(6)], input_spec=spec, seed=3)
        net.create_variables(spec, device=dev)
        return net

    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, 256, (16, 84, 84, 4), dtype=torch.uint8, generator=g).to(dev)
    dq = torch.randn(16, 6, generator=g).to(dev)
    res = {}
    for fuse in (True, False):
        sequential.FUSE_CONV_PAIRS = fuse
        try:
            net = build()
            q = net.forward(x, slot="t", need_grad=True).clone()
            net.backward(dq, slot="t")
            torch.cuda.synchronize()
            res[fuse] = (q.cpu(), net.flat_grads.clone().cpu())
        finally:
            sequential.FUSE_CONV_PAIRS = True
    close(res[True][0], res[False][0])
    close(res[True][1], res[False][1], tol=5e-5)


def test_hoisted_filter_prepasses_are_bit_identical(dev):
    """networks/sequential.py issues the weights-only pre-passes of the bf16x6 kernels (filter split
    of the conv pair; fragments + tables of the conv input gradients) early, on the network's own
    stream, into per-slot scratch (aa_conv_pair_x6_phase / aa_conv_dx_frame_x6_phase): outputs and
    gradients must equal the in-place pre-pass bit for bit, also when the weights change between
    calls (a stale scratch would show) and when forward and backward run on a side stream."""
    from agents_amd.networks import sequential, layers as L
    from agents_amd.specs import tensor_spec
    spec = tensor_spec.TensorSpec((84, 84, 4), torch.uint8)

    def build():
        net = sequential.Sequential([
            L.Rescale(255.0), L.Conv2D(32, 8, 4, activation="relu"),
            L.Conv2D(64, 4, 2, activation="relu"), L.Conv2D(64, 3, 1, activation="relu"),
            L.Flatten(), L.Dense(512, activation="relu"), L.Dense(6)], input_spec=spec, seed=3)
        net.create_variables(spec, device=dev)
        return net

    g = torch.Generator().manual_seed(4)
    x = torch.randint(0, 256, (32, 84, 84, 4), dtype=torch.uint8, generator=g).to(dev)
    dq = torch.randn(32, 6, generator=g).to(dev)
    bump = torch.randn(1, generator=g).item() * 1e-3
    res = {}
    saved = (sequential.HOIST_PREP, sequential._HOIST_FWD, sequential._HOIST_BWD)
    for hoist in (True, False):
        # both halves on (the default hoists the backward pre-passes only)
        sequential.HOIST_PREP = sequential._HOIST_FWD = sequential._HOIST_BWD = hoist
        try:
            net = build()
            side = ops.new_side_stream(dev)
            outs = []
            for rep in range(3):
                if rep == 2:   # new weights: the scratch of the previous call is stale now
                    net.flat_params.mul_(1.0 + bump)
                if rep == 1:
                    side.wait_stream(torch.cuda.current_stream())
                    with ops.side_line(side):
                        q = net.forward(x, slot="t", need_grad=True).clone()
                        net.backward(dq, slot="t")
                    torch.cuda.current_stream().wait_stream(side)
                else:
                    q = net.forward(x, slot="t", need_grad=True).clone()
                    net.backward(dq, slot="t", side_stream=side)
                torch.cuda.synchronize()
                outs.append((q, net.flat_grads.clone()))
            s = net._slots[("t", 32)]
            assert bool(s.pair_prep) == hoist and bool(s.dx_prep) == hoist
            res[hoist] = outs
        finally:
            sequential.HOIST_PREP, sequential._HOIST_FWD, sequential._HOIST_BWD = saved
    for (qa, ga), (qb, gb) in zip(res[True], res[False]):
        assert torch.equal(qa, qb)
        assert torch.equal(ga, gb)
    assert not torch.equal(res[True][0][0], res[True][2][0])


def test_prepared_weights_follow_every_kind_of_write(dev):
    """DqnAgent opts its networks into prepared weights (networks/sequential.py): the filter
    pre-passes run when the weights are written -- optimizer step, target update, restore -- and a
    torch in-place write behind the agent's back is noticed through the tensor version.  Trains two
    identically seeded agents, one with the mechanism off, through eager steps, a `set_weights`, a
    checkpoint round trip and target updates: parameters stay bit-identical."""
    from agents_amd import optimizers
    from agents_amd.agents.dqn import dqn_agent
    from agents_amd.networks import sequential, layers as L
    from agents_amd.specs import tensor_spec
    from agents_amd.trajectories import time_step as ts
    from agents_amd.trajectories import trajectory
    from agents_amd.utils import common

    obs_spec = tensor_spec.TensorSpec((84, 84, 4), torch.uint8, "observation")
    aspec = tensor_spec.BoundedTensorSpec((), torch.int64, 0, 5, "action")
    tss = ts.time_step_spec(obs_spec)

    def build(prepared):
        old = sequential.PREPARED_WEIGHTS
        sequential.PREPARED_WEIGHTS = prepared
        try:
            net = sequential.Sequential([
                L.Rescale(255.0), L.Conv2D(32, 8, 4, activation="relu"),
                L.Conv2D(64, 4, 2, activation="relu"), L.Conv2D(64, 3, 1, activation="relu"),
                L.Flatten(), L.Dense(512, activation="relu"), L.Dense(6)], seed=3)
            agent = dqn_agent.DqnAgent(tss, aspec, q_network=net,
                                       optimizer=optimizers.RMSprop(2.5e-4, rho=0.95, momentum=0.0,
                                                                    epsilon=0.01, centered=True),
                                       td_errors_loss_fn=common.element_wise_huber_loss, gamma=0.99,
                                       epsilon_greedy=0.1, target_update_period=2, seed=5)
            agent.initialize()
            return agent
        finally:
            sequential.PREPARED_WEIGHTS = old

    g = torch.Generator().manual_seed(11)

    def batch():
        B = 16
        obs = torch.randint(0, 256, (B, 2, 84, 84, 4), dtype=torch.uint8, generator=g).to(dev)
        st = torch.ones(B, 2, dtype=torch.int32, device=dev)
        return trajectory.Trajectory(
            step_type=st, observation=obs,
            action=torch.randint(0, 6, (B, 2), generator=g).to(dev), policy_info=(),
            next_step_type=st.clone(), reward=torch.randn(B, 2, generator=g).to(dev),
            discount=torch.ones(B, 2, device=dev))

    a_on, a_off = build(True), build(False)
    assert a_on._q_network._pw is not None and a_off._q_network._pw is None
    assert a_on._target_q_network._pw is not None
    for step in range(7):
        exp = batch()
        if step == 3:     # a write behind the agent's back (torch op: the version tells)
            w = [v * 1.01 for v in a_on._q_network.get_weights()]
            a_on._q_network.set_weights(w)
            a_off._q_network.set_weights(w)
            assert not a_on._q_network._prepared_ok()
        if step == 5:     # restore: parameters written by copy_, pre-passes re-run by the agent
            sd = a_on.state_dict()
            a_on.load_state_dict(sd)
            a_off.load_state_dict(a_off.state_dict())
            assert a_on._q_network._prepared_ok()
        la, lb = a_on.train(exp), a_off.train(exp)
        assert a_on._q_network._prepared_ok() and a_on._target_q_network._prepared_ok()
        assert torch.equal(la.loss, lb.loss), f"step {step}"
        assert torch.equal(a_on._q_network.flat_params, a_off._q_network.flat_params)
        assert torch.equal(a_on._target_q_network.flat_params, a_off._target_q_network.flat_params)


def test_pair_without_the_middle_output(dev):
    """y1 = None on the bf16x6 kernel (a forward no backward pass follows: the policy's, the target
    network's): the second output is bit-identical and nothing is written for the first."""
    rng = np.random.default_rng(21)
    B = 40
    x = rnd(rng, B, 20, 20, 32).to(dev)
    w1, b1 = (rnd(rng, 4, 4, 32, 64) * 0.05).to(dev), rnd(rng, 64).to(dev)
    w2, b2 = (rnd(rng, 3, 3, 64, 64) * 0.05).to(dev), rnd(rng, 64).to(dev)
    y1, y2 = torch.empty(B, 9, 9, 64, device=dev), torch.empty(B, 7, 7, 64, device=dev)
    ops.conv_pair_forward(x, w1, b1, 2, "relu", y1, w2, b2, 1, "relu", y2)
    y2b = torch.full_like(y2, float("nan"))
    ops.conv_pair_forward(x, w1, b1, 2, "relu", None, w2, b2, 1, "relu", y2b)
    assert torch.equal(y2, y2b)
