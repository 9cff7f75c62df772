"""conv1 -> conv2 -> conv3 on uint8 frames in one launch (csrc/conv_triple_x6.h) against float64,
against the two-launch path it replaces (aa_gemm_f32's uint8 conv + the fused pair), with optional
intermediate outputs, prepared filter planes, strided batches -- and the Sequential integration
(QNetwork forward, tf_agents/networks/q_network.py:46-158 over the Atari stack of
examples/dqn/mnih15/dqn_train_eval_atari.py:80-112): forward values, the backward pass that reads
the stored intermediates, and the planes the optimizer keeps current."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from agents_amd import ops

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(autouse=True)
def triple_on(monkeypatch):
    """The kernel is opt-in (AA_FUSE_CONV_TRIPLE=1: it loses inside the DQN iteration, ops.py); these
    tests switch it on."""
    monkeypatch.setattr(ops, "CONV_TRIPLE", True)
    ops._TRIPLE_WS.clear()
    yield
    ops._TRIPLE_WS.clear()

ATARI = ((8, 8, 4, 32, "relu"), (4, 4, 2, 64, "relu"), (3, 3, 1, 64, "relu"))


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


def make(rng, B, H, W, C, layers, bias=True):
    x = torch.from_numpy(rng.integers(0, 256, (B, H, W, C), dtype=np.uint8))
    ws, bs, cin = [], [], C
    for KH, KW, st, Fo, act in layers:
        ws.append(rnd(rng, KH, KW, cin, Fo) * (1.5 / np.sqrt(KH * KW * cin)))
        bs.append(rnd(rng, Fo) * 0.1 if bias else None)
        cin = Fo
    return x, ws, bs


def reference(x, ws, bs, layers, a_div):
    cur = x.double() / a_div
    outs = []
    for w, b, (KH, KW, st, Fo, act) in zip(ws, bs, layers):
        cur = act_ref(conv_ref(cur, w, b, st), act)
        outs.append(cur)
    return outs


def run_triple(dev, x, ws, bs, layers, a_div=255.0, keep=(True, True), prepared=None):
    B, H, W, C = x.shape
    ys, h, w_, strides, acts = [], H, W, [], []
    for i, (KH, KW, st, Fo, act) in enumerate(layers):
        h, w_ = ops.conv_out_hw(h, w_, KH, KW, st)
        y = torch.full((B, h, w_, Fo), float("nan"), device=dev)
        ys.append(y if (i == 2 or keep[i]) else None)
        strides.append(st)
        acts.append(act)
    wd = [w.to(dev) for w in ws]
    bd = [None if b is None else b.to(dev) for b in bs]
    ops.conv_triple_forward(x if x.is_cuda else x.to(dev), wd, bd, strides, acts, ys, a_div=a_div,
                            prepared=prepared)
    return ys, wd, bd, strides


CASES = [   # (B, H, W, C, layers)
    (8, 84, 84, 4, ATARI),                                   # the Atari stack
    (300, 84, 84, 4, ATARI),                                 # more frames than CUs
    (1100, 84, 84, 4, ATARI),                                # frames looped per workgroup
    (5, 44, 44, 4, ((8, 8, 4, 32, "tanh"), (4, 4, 2, 32, None), (2, 2, 1, 16, "relu"))),
    (3, 36, 52, 8, ((4, 4, 4, 32, "relu"), (3, 3, 1, 32, "relu"), (3, 3, 2, 16, None))),
]


def test_atari_stack_qualifies(dev):
    ws = [torch.empty(KH, KW, ci, Fo) for (KH, KW, st, Fo, a), ci in zip(ATARI, (4, 32, 64))]
    n = ops.conv_triple_prepare_bytes((256, 84, 84, 4), ws, (4, 2, 1))
    # k-steps x column tiles x planes x 1 KiB for conv1, conv2, conv3
    assert n == (8 * 2 + 16 * 4 + 18 * 4) * 3 * 64 * 16
    # a first layer whose patch rows are not whole 32-byte steps keeps the two-launch path
    assert ops.conv_triple_prepare_bytes((8, 84, 84, 3), [torch.empty(8, 8, 3, 32)] + ws[1:],
                                         (4, 2, 1)) == 0


@pytest.mark.parametrize("cfg", CASES)
@pytest.mark.parametrize("bias", [True, False])
def test_conv_triple_forward_vs_float64(dev, cfg, bias):
    B, H, W, C, layers = cfg
    rng = np.random.default_rng(B + H + C)
    x, ws, bs = make(rng, B, H, W, C, layers, bias)
    ys, *_ = run_triple(dev, x, ws, bs, layers)
    refs = reference(x, ws, bs, layers, 255.0)
    for y, r in zip(ys, refs):
        close(y, r)


def test_first_layer_exact_on_integer_weights(dev):
    """Bytes x small-integer weights without the /255: every partial sum is an exact integer in
    fp32, so conv1's output equals the float64 convolution bit for bit (an operand-order or
    fragment-layout slip shows as a wrong integer, not as rounding noise)."""
    rng = np.random.default_rng(3)
    B = 6
    x = torch.from_numpy(rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8))
    ws = [torch.from_numpy(rng.integers(-3, 4, (8, 8, 4, 32)).astype(np.float32)),
          rnd(rng, 4, 4, 32, 64) * 1e-4, rnd(rng, 3, 3, 64, 64) * 0.05]
    bs = [torch.from_numpy(rng.integers(-5, 6, (32,)).astype(np.float32)), None, None]
    layers = ((8, 8, 4, 32, None), (4, 4, 2, 64, "relu"), (3, 3, 1, 64, None))
    ys, *_ = run_triple(dev, x, ws, bs, layers, a_div=1.0)
    refs = reference(x, ws, bs, layers, 1.0)
    assert torch.equal(ys[0].cpu().double(), refs[0])
    close(ys[1], refs[1])
    close(ys[2], refs[2])


def test_optional_outputs_prepared_planes_and_strided_batches(dev):
    """ys[0] / ys[1] = None leaves the last output bit-identical; so do planes prepared ahead of the
    call and a batch read in place from a [B, T, ...] sample (obs[:, 0], the train step's view)."""
    rng = np.random.default_rng(11)
    B = 37
    x, ws, bs = make(rng, 2 * B, 84, 84, 4, ATARI)
    xd = x.to(dev).view(B, 2, 84, 84, 4)
    full, wd, bd, strides = run_triple(dev, xd[:, 0], ws, bs, ATARI)
    dense, *_ = run_triple(dev, xd[:, 0].contiguous(), ws, bs, ATARI)
    for a, b in zip(full, dense):
        assert torch.equal(a, b)
    last_only, *_ = run_triple(dev, xd[:, 0], ws, bs, ATARI, keep=(False, False))
    assert last_only[0] is None and last_only[1] is None and torch.equal(last_only[2], full[2])
    n = ops.conv_triple_prepare_bytes((B, 84, 84, 4), wd, strides)
    planes = torch.empty((n,), dtype=torch.uint8, device=dev)
    ops.conv_triple_prepare((B, 84, 84, 4), wd, strides, planes)
    prep, *_ = run_triple(dev, xd[:, 1], ws, bs, ATARI, prepared=planes)
    ref, *_ = run_triple(dev, xd[:, 1], ws, bs, ATARI)
    for a, b in zip(prep, ref):
        assert torch.equal(a, b)


def test_matches_the_two_launch_path(dev):
    """Same arithmetic class as conv_forward (uint8, bf16x3) + conv_pair_forward (bf16x6): the
    outputs agree to fp32 rounding (the MFMA shapes differ, so not bit for bit)."""
    rng = np.random.default_rng(5)
    B = 64
    x, ws, bs = make(rng, B, 84, 84, 4, ATARI)
    ys, wd, bd, strides = run_triple(dev, x, ws, bs, ATARI)
    xd = x.to(dev)
    y1 = torch.empty_like(ys[0])
    ops.conv_forward(xd, wd[0], bd[0], 4, "relu", y1, a_div=255.0)
    y2, y3 = torch.empty_like(ys[1]), torch.empty_like(ys[2])
    ops.conv_pair_forward(y1, wd[1], bd[1], 2, "relu", y2, wd[2], bd[2], 1, "relu", y3)
    for a, b in zip(ys, (y1, y2, y3)):
        close(a, b.cpu(), tol=3e-6)


def _atari_net(dev, seed=0):
    from agents_amd.networks import layers as L
    from agents_amd.networks import sequential
    from agents_amd.specs import tensor_spec
    vs = lambda: L.VarianceScaling(2.0)
    net = sequential.Sequential([
        L.Rescale(255.0), L.Conv2D(32, 8, 4, "relu", kernel_initializer=vs()),
        L.Conv2D(64, 4, 2, "relu", kernel_initializer=vs()),
        L.Conv2D(64, 3, 1, "relu", kernel_initializer=vs()), L.Flatten(),
        L.Dense(512, "relu", kernel_initializer=vs()), L.Dense(6, kernel_initializer=vs())],
        seed=seed)
    net.create_variables(tensor_spec.TensorSpec((84, 84, 4), torch.uint8), device=dev)
    return net


def test_sequential_takes_the_triple_and_backward_reads_its_intermediates(dev, monkeypatch):
    """Sequential.forward on uint8 frames: q values and -- after backward -- every gradient equal
    those of the two-launch forward to fp32 rounding; a forward without need_grad on a fresh slot
    does not store the intermediates."""
    rng = np.random.default_rng(9)
    B = 32
    obs = torch.from_numpy(rng.integers(0, 256, (B, 84, 84, 4), dtype=np.uint8)).to(dev)
    dq = rnd(rng, B, 6).to(dev)
    calls = []
    real = ops.conv_triple_forward
    monkeypatch.setattr(ops, "conv_triple_forward",
                        lambda *a, **k: calls.append(a[5]) or real(*a, **k))
    net = _atari_net(dev)
    q = net.forward(obs, slot="t", need_grad=True).clone()
    assert len(calls) == 1 and all(y is not None for y in calls[0])
    net.backward(dq, slot="t")
    g = net.flat_grads.clone()
    net.forward(obs, slot="fresh")
    assert calls[1][0] is None and calls[1][1] is None and calls[1][2] is not None
    monkeypatch.setattr(ops, "CONV_TRIPLE", False)
    ops._TRIPLE_WS.clear()
    try:
        ref_net = _atari_net(dev)
        ref_net.flat_params.copy_(net.flat_params)
        q_ref = ref_net.forward(obs, slot="t", need_grad=True).clone()
        assert len(calls) == 2                       # the two-launch path ran
        ref_net.backward(dq, slot="t")
        close(q, q_ref.cpu(), tol=3e-6)
        close(g, ref_net.flat_grads.cpu(), tol=2e-5)
    finally:
        ops._TRIPLE_WS.clear()


def test_optimizer_keeps_the_triple_planes_current(dev):
    """DqnAgent's prepared weights: after optimizer steps through the planes-writing kernel the
    network's three split banks equal a fresh pre-pass of the new weights bit for bit, and a
    forward over them equals a forward that splits for itself."""
    from agents_amd import optimizers
    net = _atari_net(dev, seed=4)
    assert net.enable_prepared_weights()
    assert 0 in net._pw["triple"] and not net._pw["pair"]
    assert net.plane_scatter() is not None
    opt = optimizers.RMSprop(2.5e-4, 0.95, 0.95, 0.01, True)
    rng = np.random.default_rng(2)
    for _ in range(3):
        net.flat_grads.copy_(torch.from_numpy(
            rng.standard_normal(net.flat_grads.numel()).astype(np.float32)).to(dev))
        opt.apply_flat(net.flat_params, net.flat_grads, planes=net.plane_scatter())
    kept = net._pw["triple"][0].clone()
    net.refresh_prepared()
    assert torch.equal(kept, net._pw["triple"][0])
    obs = torch.from_numpy(rng.integers(0, 256, (16, 84, 84, 4), dtype=np.uint8)).to(dev)
    q_prepared = net.forward(obs, slot="a").clone()
    fresh = _atari_net(dev, seed=4)
    fresh.flat_params.copy_(net.flat_params)
    assert torch.equal(q_prepared, fresh.forward(obs, slot="a"))
