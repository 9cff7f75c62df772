"""GPU: parity AT BASELINE.json configs[2] SHAPES (PPO HalfCheetah-shaped: 2,048 envs x 128 steps,
minibatch 4,096, actor / value MLPs (64,64) tanh, GAE lambda 0.95, clip 0.2, global-norm clip 0.5,
Adam(3e-4, eps 1e-5), reward + observation normalisers ON -- the reference's defaults,
tf_agents/examples/ppo/schulman17/train_eval_lib.py:85-112,197-226) on the stack
`bench.py --config ppo` times (tools/bench_ppo.py: build): one full epoch of `PPOLearner.run`
= 64 minibatch train steps replayed as ONE HIP graph bound to the learner's minibatch buffers
(tf_agents/train/ppo_learner.py:220-248,264-335; tf_agents/agents/ppo/ppo_agent.py:834-1076).

Oracle: oracle/ppo.py (losses, autograd) + oracle/tensor_normalizer.py + oracle/perm.py (the same
Feistel permutation) + oracle/optim.py.  Link by link, each from the same inputs:
  collection   stored value predictions / Normal parameters = networks on observations normalised
               with the (fresh) statistics of collection time
  normalisers  count exact, mean / M2 at fp32 rounding after PPOLearner._update_normalizers
  preprocess   returns / GAE advantages from the normalised rewards (updated statistics)
  minibatches  for each of the 64 steps: the rows the Feistel permutation selects, advantage
               normalisation over the minibatch, every loss term, the clipped gradient
               (relative L2 of the whole flat buffer), and the parameters after the Adam step.
               The oracle takes its Adam steps with the GPU's gradients, so both sides enter every
               step from the same parameters and the comparison stays at the rounding of ONE step
               (two free-running Adam trainings drift by whole +-lr updates wherever a gradient
               element changes sign in the last bit).
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from agents_amd.utils import graph                      # noqa: E402
from oracle import optim as ooptim                       # noqa: E402
from oracle import perm as operm                         # noqa: E402
from oracle import ppo as oppo                           # noqa: E402
from oracle import tensor_normalizer as otn              # noqa: E402

pytestmark = pytest.mark.gpu

B, T, MB, D, OBS = 2048, 128, 4096, 6, 17
TOL_LOSS, TOL_GRAD, TOL_PARAM = 1e-5, 2e-5, 2e-6


def _close(got, want, rtol, atol=0.0, what=""):
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else \
        np.asarray(got, np.float64)
    want = want.detach().cpu().double().numpy() if isinstance(want, torch.Tensor) else \
        np.asarray(want, np.float64)
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg=what)


def _unflatten(agent, flat, a, v):
    """agent.flat_* layout -> [actor kernels/biases..., std_bias, value kernels/biases...]."""
    flat = torch.as_tensor(flat)
    out = []
    na = agent.actor_net.body.flat_size
    for (s0, s1), p in zip(agent.actor_net.body.segment_offsets(), a):
        out.append(flat[s0:s1].reshape(p.shape).clone())
    out.append(flat[na:na + D].clone())
    nv0 = agent.actor_net.flat_size
    for (s0, s1), p in zip(agent._value_net.body.segment_offsets(), v):
        out.append(flat[nv0 + s0:nv0 + s1].reshape(p.shape).clone())
    return out


def test_ppo_bench_configuration_matches_oracle(dev):
    import bench_ppo
    import test_gpu_ppo_agent as tpa
    rec = []
    with torch.cuda.device(dev):
        holder = {}

        def hook(_exp, li):          # PPOLearner's after_train_strategy_step_fn: once per step
            ag = holder["agent"]
            rec.append(dict(loss=li.loss.clone(), pg=li.extra.policy_gradient_loss.clone(),
                            ve=li.extra.value_estimation_loss.clone(),
                            clip=li.extra.clip_fraction.clone(), grads=ag.flat_grads.clone(),
                            params=ag.flat_params.clone()))

        w = bench_ppo.build(dev, B, T, MB, epochs=1, after_train_step_fn=hook)
        agent = holder["agent"] = w["agent"]
        assert agent.actor_net.body._fused_small_ok()     # the one-launch (64,64) MLP kernels
        a, sb, v = tpa.oracle_params(agent)
        params = a + [sb] + v
        fwd = lambda obs_n: tpa.oracle_forward(a, sb, v, torch.from_numpy(obs_n), lo=-1.0, hi=1.0)
        w["collect_driver"].run()
        raw, _ = next(iter(w["raw_dataset_fn"]()))          # the [B, T+1] element the learner sees
        T1 = T + 1
        h = dict(obs=raw.observation.cpu().numpy(), rew=raw.reward.cpu().numpy(),
                 disc=raw.discount.cpu().numpy(), st=raw.step_type.cpu().numpy(),
                 nst=raw.next_step_type.cpu().numpy(), act=raw.action.cpu().numpy(),
                 loc=raw.policy_info["dist_params"]["loc"].cpu().numpy(),
                 scale=raw.policy_info["dist_params"]["scale"].cpu().numpy(),
                 vp=raw.policy_info["value_prediction"].cpu().numpy())
        assert h["obs"].shape == (B, T1, OBS) and h["vp"].shape == (B, T1)
        # ---- collection: networks on observations normalised with the FRESH statistics ---------
        o_obs, o_rew = otn.StreamingNormalizer((OBS,)), otn.StreamingNormalizer(())
        with torch.no_grad():
            loc0, scale0, val0 = fwd(o_obs.normalize(h["obs"].reshape(-1, OBS)))
        _close(h["vp"].reshape(-1), val0, rtol=2e-5, atol=2e-6, what="stored value predictions")
        _close(h["loc"].reshape(-1, D), loc0, rtol=2e-5, atol=2e-6, what="stored Normal loc")
        _close(h["scale"].reshape(-1, D), scale0, rtol=1e-6, what="stored Normal scale")
        # ---- the run under test: normaliser update + preprocess + 64 graphed minibatch steps ---
        w["learner"].run()
        torch.cuda.synchronize()
        n_steps = (B * T1) // MB
        assert len(rec) == n_steps == 64
        assert graph.graphed_train(agent).replays == n_steps - 2, "train step not graphed"
        assert int(agent.train_step_counter.numpy()) == n_steps
        # ---- normalisers -------------------------------------------------------------------------
        o_obs.update(h["obs"])
        o_rew.update(h["rew"])
        # At 264,192 frames numpy's fp32 reduction over the LEADING axes is a sequential sum (its
        # pairwise scheme only covers the contiguous axis): measured 6e-5 off on M2 against the
        # GPU's two-pass-per-workgroup + Chan merge.  So the moments are pinned on float64 and
        # the oracle continues from those, rounded to fp32 (count: 1e-8 + N rounds to N).
        for o, x in ((o_obs, h["obs"].reshape(-1, OBS)), (o_rew, h["rew"].reshape(-1))):
            x64 = x.astype(np.float64)
            mean64 = x64.mean(axis=0)
            o.avg = np.asarray(mean64, np.float32)
            o.m2 = np.asarray(((x64 - mean64) ** 2).sum(axis=0), np.float32)
            assert np.all(np.asarray(o.count) == np.float32(x.shape[0]))
        for nrm, o in ((agent._observation_normalizer, o_obs), (agent._reward_normalizer, o_rew)):
            count, avg, m2, _ = nrm.variables
            assert np.array_equal(count.cpu().numpy().reshape(-1), np.asarray(o.count).reshape(-1))
            _close(avg, o.avg, rtol=1e-5, atol=2e-7, what="normaliser mean vs float64")
            _close(m2, o.m2, rtol=5e-6, what="normaliser M2 vs float64")
        # ---- preprocess: returns / advantages from normalised rewards ----------------------------
        rew_n = o_rew.normalize(h["rew"], clip_value=10.0, center_mean=False)
        ret, adv = oppo.compute_return_and_advantage(rew_n, h["disc"], h["nst"], h["vp"], 0.99,
                                                     0.95, True, False)
        pre = agent.preprocess_sequence(raw)       # same statistics as the learner's train pass
        ret_g = pre.policy_info["return"].cpu().numpy()
        adv_g = pre.policy_info["advantage"].cpu().numpy()
        scale_r = float(np.abs(ret).max())
        _close(ret_g, oppo.pad_last(ret), rtol=5e-5, atol=5e-6 * scale_r, what="returns")
        _close(adv_g, oppo.pad_last(adv), rtol=5e-5, atol=5e-6 * scale_r, what="advantages")
        # ---- the 64 minibatch steps ----------------------------------------------------------------
        F = B * T1
        perm = operm.random_permutation(F, 0, 0)
        flat = dict(obs=h["obs"].reshape(F, OBS), act=h["act"].reshape(F, D),
                    loc=h["loc"].reshape(F, D), scale=h["scale"].reshape(F, D),
                    st=h["st"].reshape(F), ret=ret_g.reshape(F), adv=adv_g.reshape(F))
        opt = ooptim.Adam(3e-4, eps=1e-5)
        worst = dict(loss=0.0, grad=0.0, param=0.0)
        p0 = None
        for i in range(n_steps):
            idx = perm[i * MB:(i + 1) * MB]
            mask = oppo.trajectory_mask(flat["st"][idx], flat["ret"][idx], flat["adv"][idx])
            adv_n = oppo.normalize_advantages(flat["adv"][idx])
            acts = torch.from_numpy(flat["act"][idx])
            old_logp = oppo.normal_log_prob(torch.from_numpy(flat["loc"][idx]),
                                            torch.from_numpy(flat["scale"][idx]), acts)
            loc, scale, val = fwd(o_obs.normalize(flat["obs"][idx]))
            out = oppo.losses(loc, scale, acts, old_logp, torch.from_numpy(adv_n),
                              torch.from_numpy(flat["ret"][idx]), val, torch.from_numpy(mask),
                              clip_eps=0.2, c_v=0.5)
            grads = torch.autograd.grad(out["total"], params)
            gn = torch.sqrt(sum((g ** 2).sum() for g in grads))
            sc = 0.5 * min(1.0 / float(gn), 1.0 / 0.5)          # tf.clip_by_global_norm
            clipped = [g * sc for g in grads]
            want = tpa.flat_oracle_grads(agent, clipped[:len(a)], clipped[len(a)],
                                         clipped[len(a) + 1:])
            r = rec[i]
            # losses: the value term dominates the total; the surrogate is a mean of +-O(1) terms
            # around zero (normalised advantages), so it gets the rounding of that SUM as atol
            for name, got, wanted, atol in (
                    ("total", r["loss"], out["total"], 2e-6),
                    ("value_estimation_loss", r["ve"], out["value_estimation_loss"], 0.0),
                    ("policy_gradient_loss", r["pg"], out["policy_gradient_loss"], 2e-6)):
                g_, w_ = float(got), float(wanted)
                err = max(abs(g_ - w_) - atol, 0.0) / max(abs(w_), 1e-6)
                worst["loss"] = max(worst["loss"], err)
                assert err <= TOL_LOSS, f"step {i} {name}: {g_!r} vs {w_!r}"
            assert abs(float(r["clip"]) - float(out["clip_fraction"])) <= 3.0 / MB, f"step {i}"
            got = r["grads"].cpu().numpy().astype(np.float64)
            gerr = float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))
            worst["grad"] = max(worst["grad"], gerr)
            assert gerr <= TOL_GRAD, f"step {i}: clipped gradient relative L2 error {gerr:.2e}"
            # Adam with the GPU's gradient, then the parameters
            opt.step(params, _unflatten(agent, r["grads"].cpu(), a, v))
            for got_p, want_p in zip(_unflatten(agent, r["params"].cpu(), a, v), params):
                sc_p = max(float(want_p.detach().abs().max()), 1e-6)
                err = float((got_p - want_p.detach()).abs().max()) / sc_p
                worst["param"] = max(worst["param"], err)
                assert err <= TOL_PARAM, f"step {i}: parameters {err:.2e} of max|p|"
        print(f"PPO configs[2] parity over {n_steps} graphed minibatch steps: worst loss rel err "
              f"{worst['loss']:.2e}, clipped-gradient relative L2 {worst['grad']:.2e}, parameters "
              f"after one Adam step {worst['param']:.2e} of max|p|")
