#!/usr/bin/env python
"""Regenerates tests/golden/oracle_vectors.npz and random123_philox4x32_10.json from oracle/.
(The reference cannot be imported here -- see README.md in this directory -- so these vectors
freeze the oracle, which is itself pinned on reference_known_answers.json.)

    python tests/golden/make_oracle_vectors.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import dqn as odqn  # noqa: E402
from oracle import ppo as oppo  # noqa: E402
from oracle import replay as oreplay  # noqa: E402
from oracle import value_ops as ovo  # noqa: E402


def main():
    out = {}
    rng = np.random.default_rng(20260922)
    # ---- replay: raw draws -> rows / probabilities (tf_uniform_replay_buffer.py:242-292) ------
    for name, (last_id, B, L, T) in {"notfull": (5, 4, 16, 2), "full": (37, 4, 16, 2),
                                     "atari": (100000, 256, 3906, 2)}.items():
        a, c = oreplay.raw_draws(seed=7, call=3, n=64)
        rows, prob = oreplay.rows_from_draws(a, c, last_id, B, L, T)
        out[f"replay_{name}_args"] = np.array([last_id, B, L, T, 7, 3, 64], np.int64)
        out[f"replay_{name}_rows"] = rows.astype(np.int64)
        out[f"replay_{name}_prob"] = prob
    # ---- DQN TD loss (dqn_agent.py:462-579) ------------------------------------------------------
    B, T, A = 64, 3, 6
    q = rng.normal(size=(B, A)).astype(np.float32)
    qt = rng.normal(size=(B, A)).astype(np.float32)
    act = rng.integers(0, A, size=(B, T)).astype(np.int64)
    rew = rng.normal(size=(B, T)).astype(np.float32)
    disc = (rng.uniform(size=(B, T)) > 0.1).astype(np.float32)
    st = rng.integers(0, 3, size=(B, T)).astype(np.int32)
    for kind in ("huber", "squared"):
        r = odqn.td_loss_from_q(q, qt, act, rew, disc, st, gamma=0.99, loss=kind)
        out[f"dqn_{kind}_loss"] = np.float32(r["loss"])
        out[f"dqn_{kind}_td_error"] = r["td_error"]
        out[f"dqn_{kind}_dq"] = r["dq"]
    out.update(dqn_q=q, dqn_qt=qt, dqn_act=act, dqn_rew=rew, dqn_disc=disc, dqn_st=st)
    # ---- scans (value_ops.py:21-164) ----------------------------------------------------------
    Bt, Tt = 37, 70
    r2 = rng.normal(size=(Tt, Bt)).astype(np.float32)
    d2 = (rng.uniform(size=(Tt, Bt)) * (rng.uniform(size=(Tt, Bt)) > 0.1)).astype(np.float32)
    v2 = rng.normal(size=(Tt, Bt)).astype(np.float32)
    fv = rng.normal(size=Bt).astype(np.float32)
    out.update(scan_r=r2, scan_d=d2, scan_v=v2, scan_fv=fv,
               scan_return=ovo.discounted_return(r2, d2, fv),
               scan_gae=ovo.generalized_advantage_estimation(v2, fv, d2, r2, 0.95))
    # ---- PPO loss terms (ppo_agent.py:481-615) ---------------------------------------------------
    N, D = 128, 4
    loc = rng.normal(size=(N, D)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, size=(N, D)).astype(np.float32)
    oloc = (loc + 0.1 * rng.normal(size=(N, D))).astype(np.float32)
    oscale = (scale * rng.uniform(0.9, 1.1, size=(N, D))).astype(np.float32)
    acts = (oloc + oscale * rng.normal(size=(N, D))).astype(np.float32)
    t = torch.from_numpy
    old_logp = oppo.normal_log_prob(t(oloc), t(oscale), t(acts)).numpy()
    adv = rng.normal(size=N).astype(np.float32)
    ret = rng.normal(size=N).astype(np.float32)
    vp = rng.normal(size=N).astype(np.float32)
    w = (rng.uniform(size=N) > 0.2).astype(np.float32)
    res = oppo.losses(t(loc), t(scale), t(acts), t(old_logp), t(adv), t(ret), t(vp), t(w),
                      clip_eps=0.2, c_v=0.5, c_e=0.01, old_loc=t(oloc), old_scale=t(oscale),
                      kl_beta=1.0, kl_cutoff_coef=1000.0, kl_cutoff=0.02)
    out.update(ppo_loc=loc, ppo_scale=scale, ppo_old_loc=oloc, ppo_old_scale=oscale,
               ppo_actions=acts, ppo_old_logp=old_logp, ppo_adv=adv, ppo_ret=ret, ppo_vpred=vp,
               ppo_w=w)
    for k in ("policy_gradient_loss", "value_estimation_loss", "entropy_regularization_loss",
              "kl_penalty_loss", "clip_fraction", "total", "mean_kl"):
        out[f"ppo_{k}"] = np.float32(float(res[k]))
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), **out)
    kat = [
        {"counter": [0, 0, 0, 0], "key": [0, 0],
         "out": [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]},
        {"counter": [0xffffffff] * 4, "key": [0xffffffff] * 2,
         "out": [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]},
        {"counter": [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
         "key": [0xa4093822, 0x299f31d0],
         "out": [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]}]
    with open(os.path.join(HERE, "random123_philox4x32_10.json"), "w") as f:
        json.dump({"source": "Random123 kat_vectors, philox4x32 10", "vectors": kat}, f, indent=1)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
