"""GPU parity of the elementwise / scan / loss kernels against the numpy oracle.

Bit-exact where the arithmetic is elementwise fp32 in reference op order (TD loss per element,
scans, env stream, optimizers are compared to tight tolerance because torch-CPU may fuse
differently); scalar reductions to 1e-6 relative."""
import numpy as np
import pytest
import torch

from agents_amd import _lib, ops
from oracle import dqn as odqn
from oracle import env as oenv
from oracle import optim as ooptim
from oracle import value_ops as ovo

pytestmark = pytest.mark.gpu


def dv(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a), device=dev)


# ---- DQN TD loss -------------------------------------------------------------------------------
@pytest.mark.parametrize("B,T,A", [(2, 2, 2), (256, 2, 6), (64, 4, 3), (1000, 3, 18)])
@pytest.mark.parametrize("kind", ["huber", "squared"])
@pytest.mark.parametrize("double_q,use_mask,use_w", [(False, False, False), (True, True, True)])
def test_dqn_td_loss_vs_oracle(dev, B, T, A, kind, double_q, use_mask, use_w):
    rng = np.random.RandomState(B + T + A)
    q = rng.randn(B, A).astype(np.float32) * 3
    qt = rng.randn(B, A).astype(np.float32) * 3
    qs = rng.randn(B, A).astype(np.float32) if double_q else None
    mask = None
    if use_mask:
        mask = (rng.rand(B, A) > 0.3).astype(np.int32)
        mask[np.arange(B), rng.randint(0, A, B)] = 1
    act = rng.randint(0, A, size=(B, T)).astype(np.int64)
    rew = rng.randn(B, T).astype(np.float32)
    disc = (rng.rand(B, T) > 0.15).astype(np.float32) * rng.choice([1.0, 0.9], (B, T)).astype(
        np.float32)
    st = rng.randint(0, 3, size=(B, T)).astype(np.int32)
    w = rng.rand(B).astype(np.float32) if use_w else None
    if use_w:
        w[::7] = 0.0
    want = odqn.td_loss_from_q(q, qt, act, rew, disc, st, gamma=0.99, reward_scale=0.5,
                               weights=w, loss=kind, q_next_select=qs, next_mask=mask,
                               global_batch=2 * B)
    loss = torch.zeros(1, device=dev)
    td_loss = torch.zeros(B, device=dev)
    td_err = torch.zeros(B, device=dev)
    dq = torch.full((B, A), 7.0, device=dev)
    ops.dqn_td_loss(dv(q, dev), dv(qt, dev), None if qs is None else dv(qs, dev),
                    None if mask is None else dv(mask, dev), dv(act, dev), dv(rew, dev),
                    dv(disc, dev), dv(st, dev), None if w is None else dv(w, dev), 0.99, 0.5,
                    _lib.AA_LOSS_HUBER if kind == "huber" else _lib.AA_LOSS_SQUARED,
                    float(2 * B), loss, td_loss, td_err, dq)
    np.testing.assert_array_equal(td_loss.cpu().numpy(), want["td_loss"])
    np.testing.assert_array_equal(td_err.cpu().numpy(), want["td_error"])
    np.testing.assert_array_equal(dq.cpu().numpy(), want["dq"])
    np.testing.assert_allclose(loss.cpu().numpy()[0], want["loss"], rtol=2e-6)


def test_dqn_td_loss_int32_actions_and_known_answer(dev):
    """dqn_agent_test.py:178-218 through the kernel: Q tables of DummyNet -> loss 26.0."""
    q = np.array([[5, 4], [11, 8]], np.float32)
    qt = np.array([[17, 12], [23, 16]], np.float32)
    loss = torch.zeros(1, device=dev)
    o = [torch.zeros(2, device=dev), torch.zeros(2, device=dev), torch.zeros(2, 2, device=dev)]
    ops.dqn_td_loss(dv(q, dev), dv(qt, dev), None, None,
                    dv(np.array([[0, 0], [1, 1]], np.int32), dev),
                    dv(np.array([[10, 10], [20, 20]], np.float32), dev),
                    dv(np.full((2, 2), 0.9, np.float32), dev),
                    dv(np.array([[0, 1], [0, 1]], np.int32), dev), None, 1.0, 1.0,
                    _lib.AA_LOSS_HUBER, 2.0, loss, *o)
    np.testing.assert_allclose(loss.item(), 26.0, rtol=1e-6)
    np.testing.assert_allclose(o[0].cpu().numpy(), [19.8, 32.2], rtol=1e-6)


# ---- scans ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,T", [(1, 1), (7, 9), (2048, 128), (70, 130), (3, 200), (17, 128),
                                 (33, 257), (16, 513), (2051, 127)])
@pytest.mark.parametrize("batch_major", [True, False])
def test_discounted_return_and_gae_bit_exact(dev, B, T, batch_major):
    lib = _lib.load()
    rng = np.random.RandomState(B * 3 + T)
    r = rng.randn(T, B).astype(np.float32)
    d = (rng.rand(T, B) * (rng.rand(T, B) > 0.1)).astype(np.float32)
    v = rng.randn(T, B).astype(np.float32)
    fv = rng.randn(B).astype(np.float32)
    want_ret = ovo.discounted_return(r, d, fv)
    want_gae = ovo.generalized_advantage_estimation(v, fv, d, r, 0.95)
    if batch_major:
        arr = lambda a: dv(a.T, dev)
        sb, st = T, 1
        shape = (B, T)
    else:
        arr = lambda a: dv(a, dev)
        sb, st = 1, B
        shape = (T, B)
    out = torch.empty(shape, device=dev)
    R, D, V, FV = arr(r), arr(d), arr(v), dv(fv, dev)  # keep the device buffers alive
    _lib.check(lib.aa_discounted_return(R.data_ptr(), D.data_ptr(), FV.data_ptr(), B, T, sb, st,
                                        out.data_ptr(), _lib.stream_ptr()), "ret")
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got.T if batch_major else got, want_ret)
    _lib.check(lib.aa_gae(V.data_ptr(), FV.data_ptr(), D.data_ptr(), R.data_ptr(), 0.95, B, T, sb,
                          st, out.data_ptr(), _lib.stream_ptr()), "gae")
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got.T if batch_major else got, want_gae)


def test_gae_precomputed_vector(dev):  # value_ops_test.py:239-278
    lib = _lib.load()
    d = np.array([[1, 1, 1, 1, 0, .9, .9, .9, 0]] * 2, np.float32)
    out = torch.empty(2, 9, device=dev)
    args = [dv(np.full((2, 9), 3.0, np.float32), dev), dv(np.full(2, 3.0, np.float32), dev),
            dv(d, dev), dv(np.ones((2, 9), np.float32), dev)]
    _lib.check(lib.aa_gae(*[a.data_ptr() for a in args], 0.95, 2, 9, 9, 1, out.data_ptr(),
                          _lib.stream_ptr()), "gae")
    truth = [2.0808625, 1.13775, 0.145, -0.9, -2.0, 0.56016475, -0.16355, -1.01, -2.0]
    np.testing.assert_allclose(out.cpu().numpy(), [truth, truth], rtol=1e-5)


@pytest.mark.parametrize("n", [6, 4096, 262144])
def test_normalize_moments(dev, n):
    lib = _lib.load()
    rng = np.random.RandomState(n)
    x = (rng.randn(n) * 3 + 1.5).astype(np.float32)
    out = torch.empty(n, device=dev)
    stats = torch.zeros(2 + 256, device=dev)
    _lib.check(lib.aa_normalize_moments(dv(x, dev).data_ptr(), n, 1e-8, out.data_ptr(),
                                        stats.data_ptr(), _lib.stream_ptr()), "norm")
    want, mean, var = ovo.normalize_advantages(x, 1e-8)
    np.testing.assert_allclose(stats[:2].cpu().numpy(), [mean, var], rtol=2e-6)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=2e-6)


# ---- optimizers / target update -----------------------------------------------------------------
def test_adam_matches_oracle(dev):
    from agents_amd import optimizers
    rng = np.random.RandomState(0)
    n = 1003 * 4
    p0 = rng.randn(n).astype(np.float32)
    opt = optimizers.Adam(1e-3, 0.9, 0.999, 1e-7)
    oo = ooptim.Adam(1e-3, 0.9, 0.999, 1e-7)
    p = dv(p0, dev).clone()
    pc = [torch.tensor(p0.copy())]
    for step in range(5):
        g = rng.randn(n).astype(np.float32)
        opt.apply_flat(p, dv(g, dev))
        oo.step(pc, [torch.tensor(g)])
    np.testing.assert_allclose(p.cpu().numpy(), pc[0].numpy(), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("centered,momentum", [(True, 0.95), (False, 0.0), (True, 0.0),
                                               (False, 0.9)])
def test_rmsprop_matches_oracle(dev, centered, momentum):
    from agents_amd import optimizers
    rng = np.random.RandomState(1)
    n = 4099
    p0 = rng.randn(n).astype(np.float32)
    opt = optimizers.RMSprop(2.5e-4, 0.95, momentum, 0.01, centered)
    oo = ooptim.RMSprop(2.5e-4, 0.95, momentum, 0.01, centered)
    p = torch.zeros(4100, device=dev)[:n]  # odd length exercises the scalar tail
    p.copy_(dv(p0, dev))
    pc = [torch.tensor(p0.copy())]
    for step in range(5):
        g = (rng.randn(n) * 0.1).astype(np.float32)
        opt.apply_flat(p, dv(g, dev))
        oo.step(pc, [torch.tensor(g)])
    np.testing.assert_allclose(p.cpu().numpy(), pc[0].numpy(), rtol=2e-6, atol=1e-7)


def test_soft_update_and_clip(dev):
    from agents_amd.utils import common
    lib = _lib.load()
    rng = np.random.RandomState(2)
    s = rng.randn(1001).astype(np.float32)
    t0 = rng.randn(1001).astype(np.float32)
    t = dv(t0, dev).clone()
    common.soft_variables_update(dv(s, dev), t, tau=0.005)
    want = (np.float32(1 - 0.005) * t0 + np.float32(0.005) * s).astype(np.float32)
    np.testing.assert_array_equal(t.cpu().numpy(), want)
    common.soft_variables_update(dv(s, dev), t, tau=1.0)
    np.testing.assert_array_equal(t.cpu().numpy(), s)
    # per-tensor and global clipping
    g0 = rng.randn(100).astype(np.float32) * 5
    offs = dv(np.array([0, 30, 64, 100], np.int64), dev)
    for per_tensor in (1, 0):
        g = dv(g0, dev).clone()
        ss = torch.zeros(3, device=dev)
        _lib.check(lib.aa_segment_sumsq(g.data_ptr(), offs.data_ptr(), 3, ss.data_ptr(),
                                        _lib.stream_ptr()), "sumsq")
        _lib.check(lib.aa_clip_by_norm(g.data_ptr(), offs.data_ptr(), 3, ss.data_ptr(), 2.0,
                                       per_tensor, _lib.stream_ptr()), "clip")
        want = g0.copy()
        segs = [(0, 30), (30, 64), (64, 100)]
        if per_tensor:
            for a, b in segs:
                nrm = np.sqrt((g0[a:b].astype(np.float64) ** 2).sum())
                want[a:b] = g0[a:b] * 2.0 / max(nrm, 2.0)
        else:
            gn = np.sqrt((g0.astype(np.float64) ** 2).sum())
            want = g0 * 2.0 * min(1.0 / gn, 0.5)
        np.testing.assert_allclose(g.cpu().numpy(), want, rtol=1e-5)


# ---- rollout -------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,elems", [("u8", 84 * 84 * 4), ("u8", 15), ("f32", 4), ("f32", 17)])
def test_vecenv_bit_exact_vs_oracle(dev, kind, elems):
    lib = _lib.load()
    B = 37
    cur = np.random.RandomState(0).randint(0, 3, B).astype(np.int32)
    counter = torch.full((1,), 5, dtype=torch.int64, device=dev)
    st = torch.empty(B, dtype=torch.int32, device=dev)
    rew = torch.empty(B, device=dev)
    disc = torch.empty(B, device=dev)
    obs = torch.empty((B, elems), dtype=torch.uint8 if kind == "u8" else torch.float32, device=dev)
    for force in (0, 1):
        _lib.check(lib.aa_vecenv_random_step(
            dv(cur, dev).data_ptr(), B, elems, 0 if kind == "u8" else 1, -4.0, 4.0, 0.3, 77,
            counter.data_ptr(), None, force, st.data_ptr(), rew.data_ptr(), disc.data_ptr(),
            obs.data_ptr(), _lib.stream_ptr()), "env")
        w_st, w_rew, w_disc, w_obs = oenv.step(cur, B, elems, kind, -4.0, 4.0, 0.3, 77, 5,
                                               bool(force))
        np.testing.assert_array_equal(st.cpu().numpy(), w_st)
        np.testing.assert_array_equal(rew.cpu().numpy(), w_rew)
        np.testing.assert_array_equal(disc.cpu().numpy(), w_disc)
        np.testing.assert_array_equal(obs.cpu().numpy(), w_obs)


def test_vecenv_advances_its_counter_in_kernel(dev):
    """With arrival words (144: nine counters on nine cache lines) the step kernel's last workgroup
    advances the step counter itself, whatever the grid size (no separate bump launch): same
    outputs, counter + 1, arrival words back to zero."""
    lib = _lib.load()
    B, elems = 300, 84 * 84 * 4       # many workgroups
    cur = torch.ones(B, dtype=torch.int32, device=dev)
    outs = []
    for arrival in (False, True):
        counter = torch.tensor([5, 0], dtype=torch.int64).to(dev)
        words = torch.zeros(144, dtype=torch.int64, device=dev)
        st = torch.empty(B, dtype=torch.int32, device=dev)
        rew, disc = torch.empty(B, device=dev), torch.empty(B, device=dev)
        obs = torch.empty((B, elems), dtype=torch.uint8, device=dev)
        for _ in range(3):
            _lib.check(lib.aa_vecenv_random_step(
                cur.data_ptr(), B, elems, 0, 0.0, 255.0, 0.3, 77, counter.data_ptr(),
                words.data_ptr() if arrival else None, 0, st.data_ptr(), rew.data_ptr(),
                disc.data_ptr(), obs.data_ptr(), _lib.stream_ptr()), "env")
            if not arrival:
                _lib.check(lib.aa_counter_add(counter.data_ptr(), 1, _lib.stream_ptr()), "add")
        assert counter.cpu().tolist() == [8, 0]
        assert int(words.abs().sum().item()) == 0
        outs.append((st.cpu(), rew.cpu(), obs.cpu()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_eps_greedy(dev):
    from agents_amd.policies import q_policy
    from agents_amd.specs import tensor_spec
    from agents_amd.trajectories import time_step as ts
    aspec = tensor_spec.BoundedTensorSpec((), torch.int64, 0, 5)
    tspec = ts.time_step_spec(tensor_spec.TensorSpec((3,), torch.float32))
    pol = q_policy.RandomTFPolicy(tspec, aspec, seed=5)
    q = torch.randn(4096, 6, device=dev)
    greedy = pol.select(q, None, 0.0)
    assert torch.equal(greedy.cpu(), q.cpu().argmax(1))
    a = pol.select(q, None, 1.0).cpu().numpy()
    counts = np.bincount(a, minlength=6)
    assert a.min() >= 0 and a.max() <= 5 and counts.min() > 4096 / 6 * 0.8
    a2 = pol.select(q, None, 1.0).cpu().numpy()
    assert not np.array_equal(a, a2)  # call counter advances the stream
    mixed = pol.select(q, None, 0.1).cpu().numpy()
    frac = (mixed != q.cpu().argmax(1).numpy()).mean()
    assert 0.04 < frac < 0.13  # ~ eps * (1 - 1/A)
    mask = torch.zeros(4096, 6, dtype=torch.int32, device=dev)
    mask[:, 2] = 1
    mask[:, 4] = 1
    am = pol.select(q, mask, 1.0).cpu().numpy()
    assert set(np.unique(am)) == {2, 4}
    gm = pol.select(q, mask, 0.0).cpu()
    qm = q.cpu().clone()
    qm[:, [0, 1, 3, 5]] = -3.4e38
    assert torch.equal(gm, qm.argmax(1))
    # ties -> first arg-max (Categorical(logits).mode())
    tie = torch.zeros(8, 6, device=dev)
    assert torch.equal(pol.select(tie, None, 0.0).cpu(), torch.zeros(8, dtype=torch.int64))


# ---- PPO loss ------------------------------------------------------------------------------------
def _ppo_ref(z, sb, am, ag, act, olp, adv, ret, vp, ovp, w, clip, vclip, cv, ce, denom):
    z = torch.tensor(z, dtype=torch.float64, requires_grad=True)
    sb_t = torch.tensor(sb, dtype=torch.float64, requires_grad=True)
    vp_t = torch.tensor(vp, dtype=torch.float64, requires_grad=True)
    loc = torch.tensor(am) + torch.tensor(ag) * torch.tanh(z)
    scale = torch.nn.functional.softplus(sb_t)
    dist = torch.distributions.Normal(loc, scale)
    lp = dist.log_prob(torch.tensor(act, dtype=torch.float64)).sum(-1)
    ent = dist.entropy().sum(-1)
    ratio = torch.exp(lp - torch.tensor(olp, dtype=torch.float64))
    a = torch.tensor(adv, dtype=torch.float64)
    wt = torch.tensor(w, dtype=torch.float64)
    pg = (-torch.minimum(ratio * a, torch.clamp(ratio, 1 - clip, 1 + clip) * a) * wt).sum() / denom
    R = torch.tensor(ret, dtype=torch.float64)
    verr = (R - vp_t) ** 2
    if vclip > 0:
        ov = torch.tensor(ovp, dtype=torch.float64)
        vc = ov + torch.clamp(vp_t - ov, -vclip, vclip)
        verr = torch.maximum(verr, (R - vc) ** 2)
    vl = (verr * wt).sum() / denom * cv
    el = (-ent * wt).sum() / denom * ce
    total = pg + vl + el
    gz, gsb, gv = torch.autograd.grad(total, [z, sb_t, vp_t])
    cf = ((ratio - 1).abs() > clip).double().mean()
    return pg.item(), vl.item(), el.item(), cf.item(), gz.numpy(), gsb.numpy(), gv.numpy()


@pytest.mark.parametrize("N,D,vclip", [(4096, 6, 0.0), (300, 1, 0.2), (70000, 17, 0.0)])
def test_ppo_loss_vs_autograd(dev, N, D, vclip):
    lib = _lib.load()
    rng = np.random.RandomState(N + D)
    z = rng.randn(N, D).astype(np.float32) * 0.5
    sb = (rng.randn(D) * 0.3).astype(np.float32)
    am = (rng.randn(D) * 0.1).astype(np.float32)
    ag = (1 + rng.rand(D)).astype(np.float32)
    act = rng.randn(N, D).astype(np.float32)
    olp = (-0.5 * D - rng.rand(N) * D).astype(np.float32)
    adv = rng.randn(N).astype(np.float32)
    ret = rng.randn(N).astype(np.float32)
    vp = rng.randn(N).astype(np.float32)
    ovp = (vp + rng.randn(N) * 0.3).astype(np.float32)
    w = (rng.rand(N) > 0.2).astype(np.float32)
    # make log-probs comparable so ratios straddle the clip range
    loc = am + ag * np.tanh(z)
    sc = np.log1p(np.exp(sb))
    lp = (-0.5 * ((act - loc) / sc) ** 2 - np.log(sc) - 0.9189385).sum(-1)
    olp = (lp + rng.randn(N) * 0.3).astype(np.float32)
    denom = float(2 * N)
    dz = torch.empty(N, D, device=dev)
    db = torch.empty(N, D, device=dev)
    dvv = torch.empty(N, device=dev)
    stats = torch.zeros(8 + 5 * 256, device=dev)
    t = [dv(x, dev) for x in (z, sb, am, ag, act, olp, adv, ret, vp, ovp, w)]
    _lib.check(lib.aa_ppo_loss(*[x.data_ptr() for x in t], N, D, 0.2, vclip, 0.5, 0.01, denom,
                               0.0, 0, dz.data_ptr(), db.data_ptr(), dvv.data_ptr(),
                               stats.data_ptr(), _lib.stream_ptr()), "ppo")
    pg, vl, el, cf, gz, gsb, gv = _ppo_ref(z, sb, am, ag, act, olp, adv, ret, vp, ovp, w, 0.2,
                                           vclip, 0.5, 0.01, denom)
    s = stats[:6].cpu().numpy()
    np.testing.assert_allclose(s[:3], [pg, vl, el], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(s[3], cf, atol=2.0 / N)
    np.testing.assert_allclose(s[5], pg + vl + el, rtol=2e-5, atol=1e-7)
    scale = max(np.abs(gz).max(), 1e-12)
    assert np.abs(dz.cpu().numpy() - gz).max() <= 3e-5 * scale
    np.testing.assert_allclose(dvv.cpu().numpy(), gv, rtol=1e-5, atol=1e-9)
    got_sb = db.cpu().double().sum(0).numpy()
    np.testing.assert_allclose(got_sb, gsb, rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("n", [1, 2, 5, 4096, 26419, 264192])
def test_random_permutation_matches_oracle(dev, n):
    """aa_random_permutation (PPOLearner's per-epoch shuffle) == oracle/perm.py bit for bit, and is
    a permutation."""
    from agents_amd import _lib
    from oracle import perm as operm
    lib = _lib.load()
    out = torch.empty((n,), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        for call in (0, 3):
            _lib.check(lib.aa_random_permutation(n, 0xDEADBEEF12345, call, out.data_ptr(),
                                                 _lib.stream_ptr()), "aa_random_permutation")
            got = out.cpu().numpy()
            assert np.array_equal(got, operm.random_permutation(n, 0xDEADBEEF12345, call))
            assert np.array_equal(np.sort(got), np.arange(n))
