"""GPU: argument checking of the entry points added in round 6 (they must refuse bad descriptors
with an error code, not launch): aa_mlp_wide_forward_sample, aa_ppo_head_forward_sample,
aa_ppo_policy_step.  Their results are pinned elsewhere, bit for bit against the launches they
replace: tests/test_gpu_sac.py::test_fused_forward_sample_is_bit_identical,
tests/test_gpu_ppo_agent.py::test_fused_policy_step_is_bit_identical."""
import ctypes

import pytest
import torch

from agents_amd import _lib
from agents_amd.agents.ppo import ppo_actor_network as pan
from agents_amd.networks import actor_distribution_network as adn
from agents_amd.networks import sequential
from agents_amd.specs import tensor_spec

pytestmark = pytest.mark.gpu
INVALID, RANGE = -22, -34


def test_ppo_policy_step_refuses_bad_descriptors(dev):
    lib = _lib.load()
    obs_spec = tensor_spec.TensorSpec((7,), torch.float32)
    act_spec = tensor_spec.BoundedTensorSpec((2,), torch.float32, -1.0, 1.0)
    actor = pan.PPOActorNetwork().create_sequential_actor_net((16, 16), act_spec, seed=1)
    value = pan.value_network((8,), "tanh", seed=2)
    actor.create_variables(obs_spec, device=dev)
    value.create_variables(obs_spec, device=dev)
    B = 9
    f = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
    obs, loc, scale, action, v = f(B, 7), f(B, 2), f(B, 2), f(B, 2), f(B)
    counter = torch.zeros((1,), dtype=torch.int64, device=dev)
    arrival = torch.zeros((1,), dtype=torch.int64, device=dev)

    def desc():
        d = _lib.PpoPolicyStepDesc()
        d.x, d.ldx, d.B = obs.data_ptr(), 7, B
        d.nrm_eps, d.nrm_clip = 1e-3, 5.0
        (d.params_a, d.n_layers_a, d.dims_a, d.acts_a, d.k_off_a, d.b_off_a) = \
            sequential.small_mlp_layout(actor.body)
        (d.params_b, d.n_layers_b, d.dims_b, d.acts_b, d.k_off_b, d.b_off_b) = \
            sequential.small_mlp_layout(value.body)
        d.value_out = v.data_ptr()
        d.std_bias = actor._head_params.data_ptr()
        d.act_mean, d.act_mag = actor._mean.data_ptr(), actor._mag.data_ptr()
        d.D = 2
        d.loc, d.scale, d.action = loc.data_ptr(), scale.data_ptr(), action.data_ptr()
        d.seed = 5
        d.call_counter_dev, d.arrival_dev = counter.data_ptr(), arrival.data_ptr()
        return d

    run = lambda d: lib.aa_ppo_policy_step(ctypes.byref(d), _lib.stream_ptr())
    assert run(desc()) == 0
    torch.cuda.synchronize()
    assert int(counter.item()) == 1 and int(arrival.item()) == 0
    for field, bad in (("x", None), ("B", 0), ("value_out", None), ("std_bias", None),
                       ("loc", None), ("action", None), ("call_counter_dev", None),
                       ("arrival_dev", None), ("D", 3), ("ldx", 3), ("act_mag", None),
                       ("nrm_mean", obs.data_ptr())):       # a mean without a variance
        d = desc()
        setattr(d, field, bad)
        assert run(d) == INVALID, field
    d = desc()
    d.clip_lo = loc.data_ptr()                                # lower bound without an upper one
    assert run(d) == INVALID
    d = desc()                                                # value body that emits two numbers
    (d.params_b, d.n_layers_b, d.dims_b, d.acts_b, d.k_off_b, d.b_off_b) = \
        sequential.small_mlp_layout(actor.body)
    assert run(d) == INVALID
    torch.cuda.synchronize()
    assert int(counter.item()) == 1                           # none of the refused calls ran


def test_head_forward_sample_and_wide_sample_refuse_bad_arguments(dev):
    lib = _lib.load()
    z = torch.zeros((4, 3), device=dev)
    bias = torch.zeros((4,), device=dev)
    out = [torch.zeros((4, 3), device=dev) for _ in range(3)]
    counter = torch.zeros((1,), dtype=torch.int64, device=dev)
    arrival = torch.zeros((1,), dtype=torch.int64, device=dev)
    ok = [z.data_ptr(), bias.data_ptr(), None, None, 4, 3, out[0].data_ptr(), out[1].data_ptr(), 7,
          counter.data_ptr(), arrival.data_ptr(), None, None, out[2].data_ptr(), _lib.stream_ptr()]
    assert lib.aa_ppo_head_forward_sample(*ok) == 0
    for i in (0, 1, 6, 7, 9, 10, 13):
        bad = list(ok)
        bad[i] = None
        assert lib.aa_ppo_head_forward_sample(*bad) == INVALID, i
    bad = list(ok)
    bad[2] = bias.data_ptr()                                   # a spec mean without a magnitude
    assert lib.aa_ppo_head_forward_sample(*bad) == INVALID
    bad = list(ok)
    bad[11] = bias.data_ptr()                                  # a clip bound without the other
    assert lib.aa_ppo_head_forward_sample(*bad) == INVALID
    torch.cuda.synchronize()
    assert int(counter.item()) == 1

    # the wide-MLP forward with a sample tail: the tail must describe the launch's actor
    obs_spec = tensor_spec.BoundedTensorSpec((11,), torch.float32, -1.0, 1.0)
    act_spec = tensor_spec.BoundedTensorSpec((3,), torch.float32, -1.0, 1.0)
    actor = adn.ActorDistributionNetwork(
        obs_spec, act_spec, fc_layer_params=(128, 96),
        continuous_projection_net=lambda spec: adn.TanhNormalProjectionNetwork(spec), seed=4)
    actor.create_variables(device=dev)
    B = 8
    x = torch.zeros((B, 11), device=dev)
    body = actor.body
    lay = body.wide_layout()
    s = body._slot("t", B, False)
    d = _lib.MlpWideFwd()
    d.layout, d.n_nets, d.B, d.x_split = lay, 1, B, 11
    d.params[0], d.x[0], d.ldx[0] = body.flat_params.data_ptr(), x.data_ptr(), 11
    for i in range(lay.n_layers):
        d.y[0][i] = s.ys[i].data_ptr()
    mean, mag = torch.zeros(3, device=dev), torch.ones(3, device=dev)
    act, logp = torch.zeros((B, 3), device=dev), torch.zeros((B,), device=dev)

    def tail():
        t = _lib.SacSampleTail()
        t.net, t.A, t.std_kind = 0, 3, 0
        t.act_mean, t.act_mag = mean.data_ptr(), mag.data_ptr()
        t.seed = 3
        t.call_counter_dev, t.arrival_dev = counter.data_ptr(), arrival.data_ptr()
        t.action, t.logp = act.data_ptr(), logp.data_ptr()
        return t

    run = lambda t: lib.aa_mlp_wide_forward_sample(ctypes.byref(d), ctypes.byref(t),
                                                   _lib.stream_ptr())
    assert run(tail()) == 0
    for field, bad in (("net", 1), ("net", -1), ("A", 2), ("A", 0), ("act_mean", None),
                       ("action", None), ("logp", None), ("call_counter_dev", None)):
        t = tail()
        setattr(t, field, bad)
        assert run(t) == INVALID, field
    t = tail()
    t.save_tanh = act.data_ptr()                               # save buffers: all or none
    assert run(t) == INVALID
    assert lib.aa_mlp_wide_forward_sample(ctypes.byref(d), None, _lib.stream_ptr()) == INVALID
    torch.cuda.synchronize()
    assert int(counter.item()) == 2 and bool(torch.isfinite(logp).all())
