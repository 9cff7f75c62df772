"""GPU: the C-ABI entry points added in round 4, each against what it replaces or restates.

  aa_uniform_sample                    bit-exact vs oracle/philox.py::uniform_bounded
                                       (RandomTFPolicy on a bounded continuous spec,
                                       tf_agents/policies/random_tf_policy.py:60-150)
  aa_conv_dw_frame_x6_slabs / _reduce  two layers through ONE reduce launch == aa_conv_dw_frame_x6
                                       per layer, bit for bit (dqn_agent.py:412-426)
  aa_adam_step_counted_target          == aa_adam_step_counted + aa_soft_update, bit for bit
                                       (sac_agent.py:286-330, 385-410 with target_update_period 1)
  critic_network.forward_two_pairs     == two forward_pair launches, bit for bit
                                       (sac_agent.py:559-640)
"""

import numpy as np
import pytest
import torch

from agents_amd import _lib, ops, optimizers
from agents_amd.networks import critic_network
from agents_amd.networks import layers as L
from agents_amd.specs import tensor_spec
from agents_amd.trajectories import time_step as ts
from oracle import philox

pytestmark = pytest.mark.gpu


def test_random_policy_on_a_continuous_spec_matches_the_oracle(dev):
    from agents_amd.policies import random_tf_policy
    lo, hi = np.array([-1.0, 0.0, -0.4], np.float32), np.array([1.0, 0.5, 0.4], np.float32)
    act = tensor_spec.BoundedTensorSpec((3,), torch.float32, lo, hi)
    tss = ts.time_step_spec(tensor_spec.TensorSpec((5,), torch.float32))
    pol = random_tf_policy.RandomTFPolicy(tss, act, seed=99)
    step = ts.restart(torch.zeros((37, 5), device=dev), batch_size=37)
    with torch.cuda.device(dev):
        for call in range(3):
            a = pol.action(step).action
            want = philox.uniform_bounded(lo, hi, 37, 99, call)
            got = a.cpu().numpy()
            assert got.shape == (37, 3)
            assert np.array_equal(got, want), f"call {call}"
            assert np.all(got >= lo) and np.all(got < hi)
    # discrete specs still go to the masked-uniform Q-policy draw
    dact = tensor_spec.BoundedTensorSpec((), torch.int64, 0, 4)
    assert type(random_tf_policy.RandomTFPolicy(tss, dact)).__name__ == "RandomTFPolicy"


def test_merged_conv_dw_reduce_equals_one_reduce_per_layer(dev):
    g = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    S = 64
    layers = [((S, 20, 20, 32), (4, 4, 32, 64), 2, 81), ((S, 9, 9, 64), (3, 3, 64, 64), 1, 49)]
    data = [(torch.relu(r(*xs)), r(S * npix, 64)) for xs, _, _, npix in layers]
    with torch.cuda.device(dev):
        ref = []
        for (xs, wsh, st, _), (x, dz) in zip(layers, data):
            gk, gb = torch.empty(wsh, device=dev), torch.empty(64, device=dev)
            ops.conv_dw(x, dz, wsh, st, gk, a_div=1.0, bias_grad=gb)
            ref.append((gk, gb))
        pend = ops.PendingDwReduce()
        out = []
        for (xs, wsh, st, _), (x, dz) in zip(layers, data):
            gk, gb = torch.full(wsh, 7.0, device=dev), torch.full((64,), 7.0, device=dev)
            ops.conv_dw(x, dz, wsh, st, gk, a_div=1.0, bias_grad=gb, defer=pend)
            out.append((gk, gb))
        assert len(pend.items) == 2            # both layers took the per-frame kernel
        ops.conv_dw_flush(pend)
        torch.cuda.synchronize()
    assert not pend.items
    for (gk, gb), (rk, rb) in zip(out, ref):
        assert torch.equal(gk, rk) and torch.equal(gb, rb)


def test_adam_with_the_target_update_in_its_launch(dev):
    g = torch.Generator().manual_seed(5)
    n = 66_307                                  # not a multiple of 4: the scalar tail runs too
    n4 = (n + 3) // 4 * 4
    p0 = torch.randn(n4, generator=g).to(dev)[:n].clone()
    t0 = torch.randn(n4, generator=g).to(dev)[:n].clone()
    grads = [torch.randn(n4, generator=g).to(dev)[:n].clone() for _ in range(3)]
    with torch.cuda.device(dev):
        pa, ta = p0.clone(), t0.clone()
        oa = optimizers.Adam(3e-4)
        pb, tb = p0.clone(), t0.clone()
        ob = optimizers.Adam(3e-4)
        lib = _lib.load()
        for gr in grads:
            oa.apply_flat(pa, gr.clone(), soft_target=(ta, 0.005))
            ob.apply_flat(pb, gr.clone())
            _lib.check(lib.aa_soft_update(tb.data_ptr(), pb.data_ptr(), n, 0.005,
                                          _lib.stream_ptr()), "aa_soft_update")
        torch.cuda.synchronize()
    assert torch.equal(pa, pb) and torch.equal(ta, tb)
    assert not torch.equal(ta, t0)
    with pytest.raises(ValueError):
        oa.apply_flat(pa, grads[0], soft_target=(ta[:-1], 0.005))


def test_two_critic_pairs_in_one_launch_equal_two_launches(dev):
    obs = tensor_spec.BoundedTensorSpec((376,), torch.float32, -1.0, 1.0)
    act = tensor_spec.BoundedTensorSpec((17,), torch.float32, -0.4, 0.4)
    mk = lambda seed: critic_network.CriticNetwork(
        (obs, act), joint_fc_layer_params=(256, 256), kernel_initializer=L.GlorotUniform(),
        last_kernel_initializer=L.GlorotUniform(), seed=seed)
    nets = [mk(s) for s in (1, 2, 3, 4)]
    for n_ in nets:
        n_.create_variables(device=dev)
    B = 256
    g = torch.Generator().manual_seed(9)
    o1, a1 = torch.randn(B, 376, generator=g).to(dev), torch.randn(B, 17, generator=g).to(dev)
    o2, a2 = torch.randn(B, 376, generator=g).to(dev), torch.randn(B, 17, generator=g).to(dev)
    with torch.cuda.device(dev):
        assert critic_network.pair_ok(nets[0], nets[1], o1, a1)
        assert critic_network.two_pairs_ok((nets[0], nets[1]), (nets[2], nets[3]))
        ta, tb = critic_network.forward_pair(nets[0], nets[1], o1, a1, slot="p")
        ta, tb = ta.clone(), tb.clone()
        qa, qb = critic_network.forward_pair(nets[2], nets[3], o2, a2, slot="q", need_grad=True)
        qa, qb = qa.clone(), qb.clone()
        (t1, t2), (q1, q2) = critic_network.forward_two_pairs(
            (nets[0], nets[1]), o1, a1, "p2", (nets[2], nets[3]), o2, a2, "q2", need_grad_b=True)
        torch.cuda.synchronize()
    assert torch.equal(t1, ta) and torch.equal(t2, tb)
    assert torch.equal(q1, qa) and torch.equal(q2, qb)
    assert not torch.equal(t1, t2) and not torch.equal(q1, t1)
