"""The committed golden fixtures (tests/golden/) against the oracle (CPU) and, with a GPU, against
the HIP kernels.  See tests/golden/README.md for what each file pins."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import dqn as odqn
from oracle import philox
from oracle import ppo as oppo
from oracle import replay as oreplay
from oracle import value_ops as ovo

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
G = np.load(os.path.join(HERE, "oracle_vectors.npz"))


def test_philox_known_answers_file():
    kat = json.load(open(os.path.join(HERE, "random123_philox4x32_10.json")))["vectors"]
    for v in kat:
        got = philox.philox4x32_10(*v["counter"], *v["key"])
        assert [int(g) for g in got] == v["out"]


def test_reference_known_answers_file_is_reproduced_by_the_oracle():
    ka = json.load(open(os.path.join(HERE, "reference_known_answers.json")))
    g = ka["gae"]
    d = np.array([g["discounts"]] * 2, np.float32)
    adv = ovo.generalized_advantage_estimation(
        np.full((2, 9), g["values"], np.float32), np.full(2, g["values"], np.float32), d,
        np.full((2, 9), g["rewards"], np.float32), g["lambda"], time_major=False)
    np.testing.assert_allclose(adv[0], g["advantages"], rtol=1e-5)
    ppo = {e["what"].split()[0].rstrip(":"): e["value"] for e in ka["ppo"]}
    assert abs(ppo["kl_cutoff_loss"] - oppo.kl_cutoff_loss([[1.5, -0.5, 6.5, -1.5, -2.3]], 5.0,
                                                           0.1, 30.0)) < 1e-5


@pytest.mark.parametrize("name", ["notfull", "full", "atari"])
def test_replay_rows_golden(name):
    last_id, B, L, T, seed, call, n = [int(x) for x in G[f"replay_{name}_args"]]
    a, c = oreplay.raw_draws(seed=seed, call=call, n=n)
    rows, prob = oreplay.rows_from_draws(a, c, last_id, B, L, T)
    np.testing.assert_array_equal(rows, G[f"replay_{name}_rows"])
    np.testing.assert_array_equal(prob, G[f"replay_{name}_prob"])


@pytest.mark.parametrize("kind", ["huber", "squared"])
def test_dqn_loss_golden(kind):
    r = odqn.td_loss_from_q(G["dqn_q"], G["dqn_qt"], G["dqn_act"], G["dqn_rew"], G["dqn_disc"],
                            G["dqn_st"], gamma=0.99, loss=kind)
    assert np.float32(r["loss"]) == G[f"dqn_{kind}_loss"]
    np.testing.assert_array_equal(r["td_error"], G[f"dqn_{kind}_td_error"])
    np.testing.assert_array_equal(r["dq"], G[f"dqn_{kind}_dq"])


def test_scans_golden():
    np.testing.assert_array_equal(ovo.discounted_return(G["scan_r"], G["scan_d"], G["scan_fv"]),
                                  G["scan_return"])
    np.testing.assert_array_equal(
        ovo.generalized_advantage_estimation(G["scan_v"], G["scan_fv"], G["scan_d"], G["scan_r"],
                                             0.95), G["scan_gae"])


# ---- the HIP kernels against the stored vectors ------------------------------------------------
@pytest.mark.gpu
def test_gpu_scans_match_golden_bit_exact(dev):
    from agents_amd import _lib
    lib = _lib.load()
    T, B = G["scan_r"].shape
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    r, d, v, fv = t(G["scan_r"]), t(G["scan_d"]), t(G["scan_v"]), t(G["scan_fv"])
    out = torch.empty((T, B), device=dev)
    _lib.check(lib.aa_discounted_return(r.data_ptr(), d.data_ptr(), fv.data_ptr(), B, T, 1, B,
                                        out.data_ptr(), _lib.stream_ptr()), "ret")
    np.testing.assert_array_equal(out.cpu().numpy(), G["scan_return"])
    _lib.check(lib.aa_gae(v.data_ptr(), fv.data_ptr(), d.data_ptr(), r.data_ptr(), 0.95, B, T, 1,
                          B, out.data_ptr(), _lib.stream_ptr()), "gae")
    np.testing.assert_array_equal(out.cpu().numpy(), G["scan_gae"])


@pytest.mark.gpu
def test_gpu_replay_rows_match_golden_bit_exact(dev):
    from agents_amd import _lib
    lib = _lib.load()
    for name in ("notfull", "full", "atari"):
        last_id, B, L, T, seed, call, n = [int(x) for x in G[f"replay_{name}_args"]]
        lid = torch.tensor([last_id], dtype=torch.int64, device=dev)
        rows = torch.empty((n, T), dtype=torch.int64, device=dev)
        prob = torch.empty((n,), dtype=torch.float32, device=dev)
        err = torch.zeros((1,), dtype=torch.int32, device=dev)
        _lib.check(lib.aa_rb_sample_rows(lid.data_ptr(), B, L, n, T, seed, call, None, rows.data_ptr(),
                                         prob.data_ptr(), err.data_ptr(), _lib.stream_ptr()),
                   "sample")
        np.testing.assert_array_equal(rows.cpu().numpy(), G[f"replay_{name}_rows"])
        np.testing.assert_array_equal(prob.cpu().numpy(), G[f"replay_{name}_prob"])
        assert int(err.item()) == 0


@pytest.mark.gpu
def test_gpu_ppo_loss_matches_golden(dev):
    from agents_amd import _lib
    lib = _lib.load()
    t = lambda k: torch.as_tensor(np.ascontiguousarray(G[k]), device=dev)
    N, D = G["ppo_loc"].shape
    stats = torch.zeros((_lib.AA_PPO_DIST_STATS,), device=dev)
    beta = torch.ones((1,), device=dev)
    keep = [t(k) for k in ("ppo_loc", "ppo_scale", "ppo_old_loc", "ppo_old_scale", "ppo_actions",
                           "ppo_old_logp", "ppo_adv", "ppo_ret", "ppo_vpred")]
    w = t("ppo_w")
    _lib.check(lib.aa_ppo_loss_dist(*[k.data_ptr() for k in keep], None, w.data_ptr(), N, D, 0.2,
                                    0.0, 0.5, 0.01, float(N), 0.0, beta.data_ptr(), 1000.0, 0.02,
                                    None, None, None, stats.data_ptr(), _lib.stream_ptr()), "ppo")
    s = stats.cpu().numpy()
    for idx, k in ((0, "policy_gradient_loss"), (1, "value_estimation_loss"),
                   (2, "entropy_regularization_loss"), (5, "kl_penalty_loss"),
                   (3, "clip_fraction"), (6, "total"), (7, "mean_kl")):
        np.testing.assert_allclose(s[idx], G[f"ppo_{k}"], rtol=2e-5, atol=1e-8)
