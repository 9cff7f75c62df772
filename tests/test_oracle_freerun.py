"""CPU: the float64 twins of oracle/freerun.py (the yardstick of tests/test_gpu_free_running.py)
restate the SAME training as the fp32 oracles: on small problems the two agree to fp32 rounding at
every step of a short free run, and the envelope rule accepts / rejects what it should."""
import numpy as np
import torch

from oracle import dqn as odqn
from oracle import freerun
from oracle import nets as onets
from oracle import optim as ooptim
from oracle import sac as osac


def test_f64_dqn_agent_tracks_the_fp32_oracle():
    rng = np.random.default_rng(0)
    layers = onets.mlp_q_layers((32, 16), 4, "relu")
    p0 = onets.init_params(layers, (8,), seed=3)
    mk = lambda: ooptim.RMSprop(1e-3, 0.95, 0.9, 0.01, True)
    o32 = odqn.OracleDqnAgent(layers, (8,), 4, p0, optimizer=mk(), gamma=0.97, loss="huber",
                              target_update_period=2)
    o64 = freerun.F64DqnAgent(layers, p0, mk(), gamma=0.97, loss="huber", target_update_period=2)
    for _ in range(5):
        obs = torch.from_numpy(rng.standard_normal((16, 2, 8)).astype(np.float32))
        act = rng.integers(0, 4, (16, 2))
        rew = rng.standard_normal((16, 2)).astype(np.float32) * 3
        disc = (rng.random((16, 2)) > 0.2).astype(np.float32)
        st = rng.integers(0, 3, (16, 2)).astype(np.int32)
        l32, _, _ = o32.train(obs, act, rew, disc, st)
        l64 = o64.train(obs, act, rew, disc, st)
        assert abs(float(l32) - l64) <= 2e-6 * max(abs(l64), 1.0)
    for p, q in zip(o32.params, o64.params):
        assert float((p.detach().double() - q.detach()).abs().max()) <= 1e-5
    for p, q in zip(o32.target, o64.target):
        assert float((p.double() - q).abs().max()) <= 1e-5


def test_f64_sac_oracle_tracks_the_fp32_oracle():
    g = torch.Generator().manual_seed(1)
    O, A, B = 5, 2, 16
    al, cl = onets.mlp_q_layers((16, 16), 2 * A, "relu"), onets.mlp_q_layers((16, 16), 1, "relu")
    pa, p1, p2 = (onets.init_params(al, (O,), seed=1), onets.init_params(cl, (O + A,), seed=2),
                  onets.init_params(cl, (O + A,), seed=3))
    mk = lambda dt: osac.OracleSacAgent(O, A, (16, 16), (16, 16), [0.0] * A, [1.0] * A, pa, p1, p2,
                                        reward_scale_factor=0.1, std_kind="clip_exp", dtype=dt)
    o32, o64 = mk(torch.float32), mk(torch.float64)
    r = lambda *s: torch.randn(*s, generator=g)
    for _ in range(4):
        args = (r(B, O), r(B, A).clamp(-1, 1), r(B, O), r(B), torch.ones(B), r(B, A), r(B, A),
                r(B, A))
        a, b = o32.train(*args), o64.train(*args)
        for n in ("loss", "critic_loss", "actor_loss", "alpha_loss"):
            assert abs(a[n] - b[n]) <= 5e-6 * max(abs(b[n]), 1.0), n
    assert o64.c1[0].dtype == torch.float64 and o64.log_alpha.dtype == torch.float64


def test_envelope_rule():
    f64 = [1.0, 2.0, 3.0]
    ok, rows = freerun.envelope([1.0, 2.0 + 2e-5, 3.0 + 1e-5], [1.0, 2.0 + 1e-5, 3.0], f64)
    assert ok and len(rows) == 3 and rows[2][3] == 0.0           # running maximum carries over
    ok, _ = freerun.envelope([1.0, 2.0 + 5e-5, 3.0], [1.0, 2.0 + 1e-5, 3.0], f64)
    assert not ok
    ok, _ = freerun.envelope([1.0 + 5e-7], [1.0], [1.0])        # inside the rounding floor
    assert ok
