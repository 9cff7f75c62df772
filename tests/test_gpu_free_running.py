"""GPU: FREE-RUNNING parity envelopes, one per BASELINE.json configuration bench.py times.

The bench-configuration tests (test_gpu_bench_config*.py) are link by link: the oracle is put back
on the GPU's parameters (DQN) or takes its optimizer steps with the GPU's gradients (PPO, SAC)
every step, because two fp32 trainings drift apart chaotically.  Here nothing is re-synchronised:
K = 8 consecutive train steps of (a) the HIP path, (b) the torch-CPU fp32 oracle and (c) the same
training in float64 (oracle/freerun.py) start from identical state and consume identical batches
(and identical N(0,1) draws for SAC), and at EVERY step

    |loss_HIP - loss_f64|  <=  3 * max_{j<=k} |loss_fp32 - loss_f64|  +  1e-6 * |loss_f64|

-- the HIP path stays as close to the exact trajectory as an independent fp32 implementation does
(contract precedent: tf_agents/train/learner_test.py:446-562 holds two runs to 1e-2; the north
star asks 1e-5 on fp32 losses, asserted here as well on the steps where the fp32 oracle itself
achieves it)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bench                                             # noqa: E402
from agents_amd.agents.sac import sac_agent              # noqa: E402
from agents_amd.utils import nest_utils                  # noqa: E402
from oracle import dqn as odqn                           # noqa: E402
from oracle import freerun                               # noqa: E402
from oracle import nets as onets                         # noqa: E402
from oracle import optim as ooptim                       # noqa: E402
from oracle import perm as operm                         # noqa: E402
from oracle import ppo as oppo                           # noqa: E402
from oracle import sac as osac                           # noqa: E402
from oracle import tensor_normalizer as otn              # noqa: E402

pytestmark = pytest.mark.gpu
K = 8


def _judge(name, hip, fp32, f64):
    ok, rows = freerun.envelope(hip, fp32, f64)
    print(f"{name}: free-running loss trajectories over {len(rows)} steps\n" +
          freerun.format_rows(rows))
    assert ok, f"{name}: HIP left the fp32 envelope around the float64 trajectory\n" + \
        freerun.format_rows(rows)
    for k, d, e_h, e_s, _ in rows:
        # north star (1e-5 relative on fp32 losses) wherever the fp32 oracle itself meets it
        if e_s <= 1e-5 * abs(d):
            assert e_h <= 3e-5 * abs(d) + 1e-9, f"{name} step {k}: {e_h:.2e} of {d:.6g}"


@pytest.mark.parametrize("period", [2500, 3])
def test_dqn_configs1_free_running_envelope(dev, period):
    """configs[1] shapes as bench.py builds them (batch 256, Mnih-15 net on uint8 84x84x4, Huber,
    centred RMSProp), replay ring shortened to 8 frames per env; eager `agent.train`.
    period = 3: the hard target update (dqn_agent.py:385-409) fires after steps 3 and 6, so the
    free-running trajectory crosses two target updates inside the K = 8 steps (the benchmark's
    2,500-step period never fires in a test-sized run); the graphed loop is held bit-identical
    to these eager steps across such updates by tests/test_gpu_early_target.py."""
    S = 256
    with torch.cuda.device(dev):
        w = bench.build_workload(dev, 0, 1, 256, 8, S, seed=1)
        w["rb"]._dataset_ring = 0
        agent, net, rb = w["agent"], w["net"], w["rb"]
        if period != 2500:
            agent._update_target = agent._get_target_updater(1.0, period)
        w["init_driver"]._num_steps = 256 * 8
        w["init_driver"].run()
        batches = [rb.get_next(S, 2)[0] for _ in range(K)]
        torch.cuda.synchronize()
        layers = onets.atari_q_layers(bench.NUM_ACTIONS)
        p0 = [torch.tensor(a) for a in net.get_weights()]
        rms = lambda: ooptim.RMSprop(2.5e-4, 0.95, 0.95, 0.01, True)
        o32 = odqn.OracleDqnAgent(layers, bench.OBS_SHAPE, bench.NUM_ACTIONS, p0, optimizer=rms(),
                                  gamma=0.99, loss="huber", target_update_period=period)
        o64 = freerun.F64DqnAgent(layers, p0, rms(), gamma=0.99, loss="huber",
                                  target_update_period=period)
        hip, fp32, f64 = [], [], []
        for exp in batches:
            hip.append(float(agent.train(exp).loss))
            st, obs, act, nst, rew, disc = [t.cpu().numpy() for t in nest_utils.flatten(exp)]
            total, _, _ = o32.train(torch.from_numpy(obs), act, rew, disc, st)
            fp32.append(float(total))
            f64.append(o64.train(torch.from_numpy(obs), act, rew, disc, st))
        _judge(f"DQN configs[1] (target update every {period})", hip, fp32, f64)
        if period != 2500:
            assert agent._target_writes >= 1 + K // period   # initialize() + the periodic updates
        # the parameters after K free steps, against float64: same yardstick
        d = lambda ps: float(torch.sqrt(sum(((p.detach().double() - q.detach()) ** 2).sum()
                                            for p, q in zip(ps, o64.params))))
        e_hip = d([v.cpu() for v in net.variables])
        e_32 = d(o32.params)
        print(f"DQN configs[1]: |theta - theta_f64| after {K} free steps: HIP {e_hip:.3e}, "
              f"torch-fp32 {e_32:.3e}")
        assert e_hip <= 3 * e_32 + 1e-6


def test_sac_configs4_free_running_envelope(dev):
    """configs[4] shapes (obs 376, action 17, (256,256) actor and twin critics, batch 256, three
    Adam(3e-4), tau 0.005); eager `agent.train` with the N(0,1) draws recorded."""
    import bench_sac
    O, A, H, S = 376, 17, 256, 256
    with torch.cuda.device(dev):
        w = bench_sac.build(dev, envs=1024, max_length=8, batch=S, record_noise=True)
        agent, rb = w["agent"], w["rb"]
        c1n, c2n = agent.critic_networks
        cpu = lambda arrs: [torch.from_numpy(np.array(a, copy=True)) for a in arrs]
        mean, mag = sac_agent._spec_means_and_magnitudes(w["action_spec"])
        mk = lambda dt: osac.OracleSacAgent(
            O, A, (H, H), (H, H), mean, mag, cpu(w["actor"].get_weights()),
            cpu(c1n.get_weights()), cpu(c2n.get_weights()), actor_lr=3e-4, critic_lr=3e-4,
            alpha_lr=3e-4, gamma=0.99, reward_scale_factor=0.1, tau=0.005, std_kind="clip_exp",
            dtype=dt)
        o32, o64 = mk(torch.float32), mk(torch.float64)
        batches = [rb.get_next(S, 2)[0] for _ in range(K)]
        hip = {n: [] for n in ("loss", "critic_loss", "actor_loss", "alpha_loss")}
        fp32 = {n: [] for n in hip}
        f64 = {n: [] for n in hip}
        for exp in batches:
            li = agent.train(exp)
            torch.cuda.synchronize()
            hip["loss"].append(float(li.loss))
            for n in ("critic_loss", "actor_loss", "alpha_loss"):
                hip[n].append(float(getattr(li.extra, n)))
            wk = agent._work[S]
            eps = [wk[n]["eps"].cpu().clone() for n in ("save_next", "save", "save_alpha")]
            e = nest_utils.map_structure(lambda t: t.cpu(), exp)
            for o, rec in ((o32, fp32), (o64, f64)):
                out = o.train(e.observation[:, 0], e.action[:, 0], e.observation[:, 1],
                              e.reward[:, 0], e.discount[:, 0], *eps)
                for n in rec:
                    rec[n].append(out[n])
        for n in ("critic_loss", "actor_loss", "alpha_loss", "loss"):
            _judge(f"SAC configs[4] {n}", hip[n], fp32[n], f64[n])


def test_ppo_configs2_free_running_envelope(dev):
    """configs[2] shapes (2,048 envs x 128 steps, minibatch 4,096, (64,64) tanh MLPs, clip 0.2,
    global-norm clip 0.5, Adam(3e-4, eps 1e-5), normalisers on) through the path bench.py times:
    one epoch of PPOLearner.run = 64 fused minibatch steps, free-running on the GPU by
    construction.  The two oracles replay the first K of them free-running as well."""
    import bench_ppo
    import test_gpu_ppo_agent as tpa
    B, T, MB, D, OBS = 2048, 128, 4096, 6, 17
    rec = []
    with torch.cuda.device(dev):
        holder = {}

        def hook(_exp, li):
            rec.append(float(li.loss))

        w = bench_ppo.build(dev, B, T, MB, epochs=1, after_train_step_fn=hook)
        agent = holder["agent"] = w["agent"]
        a, sb, v = tpa.oracle_params(agent)
        w["collect_driver"].run()
        raw, _ = next(iter(w["raw_dataset_fn"]()))
        w["learner"].run()
        torch.cuda.synchronize()
        assert len(rec) == (B * (T + 1)) // MB
        # the learner's inputs, as the GPU prepared them (statistics after the normaliser update)
        pre = agent.preprocess_sequence(raw)
        F = B * (T + 1)
        cnt, avg, m2, _ = agent._observation_normalizer.variables
        o_obs = otn.StreamingNormalizer((OBS,))
        o_obs.count = np.asarray(cnt.cpu().numpy(), np.float32).reshape(np.shape(o_obs.count))
        o_obs.avg = avg.cpu().numpy().astype(np.float32)
        o_obs.m2 = m2.cpu().numpy().astype(np.float32)
        flat = dict(obs=raw.observation.cpu().numpy().reshape(F, OBS),
                    act=raw.action.cpu().numpy().reshape(F, D),
                    loc=raw.policy_info["dist_params"]["loc"].cpu().numpy().reshape(F, D),
                    scale=raw.policy_info["dist_params"]["scale"].cpu().numpy().reshape(F, D),
                    st=raw.step_type.cpu().numpy().reshape(F),
                    ret=pre.policy_info["return"].cpu().numpy().reshape(F),
                    adv=pre.policy_info["advantage"].cpu().numpy().reshape(F))
        perm = operm.random_permutation(F, 0, 0)

        def free_run(dtype):
            cast = lambda ps: [p.detach().to(dtype).clone().requires_grad_(True) for p in ps]
            pa, psb, pv = cast(a), cast([sb])[0], cast(v)
            params = pa + [psb] + pv
            opt = ooptim.Adam(3e-4, eps=1e-5)
            t = lambda x: torch.from_numpy(np.asarray(x)).to(dtype)
            out_losses = []
            for i in range(K):
                idx = perm[i * MB:(i + 1) * MB]
                mask = oppo.trajectory_mask(flat["st"][idx], flat["ret"][idx], flat["adv"][idx])
                adv_n = oppo.normalize_advantages(flat["adv"][idx])
                acts = t(flat["act"][idx])
                old_logp = oppo.normal_log_prob(t(flat["loc"][idx]), t(flat["scale"][idx]), acts)
                loc, scale, val = tpa.oracle_forward(pa, psb, pv,
                                                     t(o_obs.normalize(flat["obs"][idx])),
                                                     lo=-1.0, hi=1.0)
                out = oppo.losses(loc, scale, acts, old_logp, t(adv_n), t(flat["ret"][idx]), val,
                                  t(mask), clip_eps=0.2, c_v=0.5)
                grads = torch.autograd.grad(out["total"], params)
                gn = torch.sqrt(sum((g ** 2).sum() for g in grads))
                sc = 0.5 * min(1.0 / float(gn), 1.0 / 0.5)
                opt.step(params, [g * sc for g in grads])
                out_losses.append(float(out["total"]))
            return out_losses

        fp32, f64 = free_run(torch.float32), free_run(torch.float64)
        _judge("PPO configs[2]", rec[:K], fp32, f64)
