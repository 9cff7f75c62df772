"""GPU: parity AT BASELINE.json configs[4] SHAPES (SAC Humanoid-shaped: obs f32[376], action
f32[17], actor and twin critics (256,256), batch 256, tau 0.005, three Adam(3e-4), 4,096 envs), on
the stack `bench.py --config sac` times (tools/bench_sac.py: build) with the replay ring shortened
to 8 frames per env -- i.e. through the GRAPHED path: the collect / sample / train HIP graphs,
train graphs bound to the sampler's ring slots.  At these widths the actor and the twin critics
take the wide-MLP kernels (csrc/mlp_wide.hip: one forward launch, two backward launches, both
critics of a pair per launch with [observation | action] read in place); the (32,32) networks of
tests/test_gpu_sac.py take the small-MLP kernels instead, so this is the agent-level check of that
route
(tf_agents/agents/sac/sac_agent.py:314-410,559-740; examples/sac/haarnoja18/sac_train_eval.py:182-199).

Oracle: oracle/sac.py (torch-CPU autograd), fed the batch the train graph consumed and the very
N(0,1) draws it used (`SacAgent.record_noise`).  Two free-running Adam trainings drift apart
chaotically (an Adam step is lr * sign(g) for the small g: a last-bit difference flips whole
elements by 2 lr), which says nothing about either side, so the comparison is link by link:
  losses      critic / actor / alpha / total             1e-5 relative (north star)
  gradients   of each phase, per tensor, relative L2     2e-5
  optimizer   the oracle takes its Adam steps with the GPU's gradients (`grads_override`), so
              parameters, targets and log_alpha are compared after every step at the rounding of
              ONE Adam / soft-update evaluation, and both sides enter the next phase (the actor
              loss differentiates through the UPDATED critics, the alpha loss samples the UPDATED
              actor) from the same point.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from agents_amd.agents.sac import sac_agent              # noqa: E402
from agents_amd.utils import graph, nest_utils          # noqa: E402
from oracle import sac as osac                           # noqa: E402

pytestmark = pytest.mark.gpu

O, A, H, ENVS, L_RING, S, STEPS = 376, 17, 256, 4096, 8, 256, 8
TOL_LOSS, TOL_GRAD, TOL_PARAM = 1e-5, 2e-5, 2e-6
# Gradients are compared ON THE KERNELS' OWN LINEAR BRANCH: the oracle evaluates its differentiated
# forwards with the kernels' ReLU activation patterns imposed (oracle/sac.py: _mlp), and every unit
# where that pattern differs from the oracle's own must have a pre-activation within FLIP_TOL of
# zero (relative to the layer's largest).  ~650 k ReLU units are differentiated per step, each
# within rounding of zero with probability ~2e-7: about one boundary flip per 8 steps, and one flip
# removes one of the 256 terms of a weight-gradient column (4e-3 of the tensor: oracle/arbiter.py).
FLIP_TOL = 1e-5


class _Recorder:
    """Iterator wrapper: remembers the element the Learner just pulled."""

    def __init__(self, it):
        self._it, self.last = it, None

    def __iter__(self):
        return self

    def __next__(self):
        self.last = next(self._it)
        return self.last


def _rel_l2(got, want):
    got = got.detach().cpu().double().reshape(-1)
    want = want.detach().double().reshape(-1)
    return float((got - want).norm() / max(float(want.norm()), 1e-30))


def _max_rel(got, want):
    got = got.detach().cpu().double().reshape(-1)
    want = want.detach().double().reshape(-1)
    return float((got - want).abs().max() / max(float(want.abs().max()), 1e-30))


def test_sac_bench_configuration_matches_oracle_through_the_graphs(dev):
    import bench_sac
    with torch.cuda.device(dev):
        w = bench_sac.build(dev, envs=ENVS, max_length=L_RING, batch=S, record_noise=True)
        agent = w["agent"]
        c1n, c2n = agent.critic_networks
        # the route under test: 256-wide layers are NOT the fused small-MLP path
        assert not w["actor"].body._fused_small_ok() and not c1n.body._fused_small_ok()
        cpu = lambda arrs: [torch.from_numpy(np.array(a, copy=True)) for a in arrs]
        mean, mag = sac_agent._spec_means_and_magnitudes(w["action_spec"])
        oracle = osac.OracleSacAgent(
            O, A, (H, H), (H, H), mean, mag, cpu(w["actor"].get_weights()),
            cpu(c1n.get_weights()), cpu(c2n.get_weights()), actor_lr=3e-4, critic_lr=3e-4,
            alpha_lr=3e-4, gamma=0.99, reward_scale_factor=0.1, tau=0.005, std_kind="clip_exp")
        it = _Recorder(iter(w["dataset"]))
        collect, lrn = w["collect"], w["learner"]
        graph.enable_overlap(dev)
        worst = dict(loss=0.0, grad=0.0, param=0.0)
        flips = 0
        try:
            ts_ = None
            for i in range(STEPS):
                ts_, _ = collect(ts_)
                li = lrn.run(iterations=1, iterator=it)
                graph.join_lanes(dev)
                torch.cuda.synchronize()
                exp = nest_utils.map_structure(lambda t: t.cpu(), it.last[0])
                wk = agent._work[S]
                eps = {k: wk[n]["eps"].cpu() for k, n in
                       (("next", "save_next"), ("actor", "save"), ("alpha", "save_alpha"))}
                n1 = c1n.flat_size
                g_c1 = [g.cpu().clone() for g in c1n.body.gradients]
                g_c2 = [g.cpu().clone() for g in c2n.body.gradients]
                g_a = [g.cpu().clone() for g in w["actor"].body.gradients]
                g_l = float(agent._log_alpha_grad[0])
                assert agent._critic_grads[:n1].data_ptr() == c1n.flat_grads.data_ptr()
                # ReLU activation patterns of the five differentiated forwards (hidden layers)
                pat = lambda net, slot: [(y > 0).cpu() for y in
                                         net.body._slots[(slot, S)].ys[:-1]] + [None]
                masks = {"c1.critic": pat(c1n, "critic"), "c2.critic": pat(c2n, "critic"),
                         "c1.actor_q": pat(c1n, "actor_q"), "c2.actor_q": pat(c2n, "actor_q"),
                         "actor": pat(w["actor"], "actor")}
                out = oracle.train(exp.observation[:, 0], exp.action[:, 0], exp.observation[:, 1],
                                   exp.reward[:, 0], exp.discount[:, 0], eps["next"], eps["actor"],
                                   eps["alpha"],
                                   grads_override=dict(critic=g_c1 + g_c2, actor=g_a, alpha=g_l),
                                   masks=masks)
                # the kernels' branch differs from the oracle's own only at numerically-zero units
                for tag, n, z_rel in oracle.flips:
                    flips += n
                    assert z_rel <= FLIP_TOL and n <= 16, \
                        f"step {i} {tag}: {n} activations differ, one at {z_rel:.2e} of the scale"
                # ---- losses ---------------------------------------------------------------------
                for name, got in (("critic_loss", li.extra.critic_loss),
                                  ("actor_loss", li.extra.actor_loss),
                                  ("alpha_loss", li.extra.alpha_loss), ("loss", li.loss)):
                    want = out[name]
                    err = abs(float(got) - want) / max(abs(want), 1e-3)
                    worst["loss"] = max(worst["loss"], err)
                    assert err <= TOL_LOSS, f"step {i} {name}: {float(got)!r} vs {want!r}"
                # ---- gradients of the three phases ------------------------------------------------
                pairs = list(zip(g_c1 + g_c2, out["critic_grads"])) + \
                    list(zip(g_a, out["actor_grads"]))
                for k, (got, want) in enumerate(pairs):
                    err = _rel_l2(got, want)
                    worst["grad"] = max(worst["grad"], err)
                    assert err <= TOL_GRAD, f"step {i} gradient {k} {tuple(want.shape)}: {err:.2e}"
                assert abs(g_l - out["alpha_grad"]) <= 1e-5 * max(abs(out["alpha_grad"]), 1e-3)
                # ---- parameters after the three Adam steps and the soft target update ------------
                groups = ((w["actor"].variables, oracle.actor), (c1n.variables, oracle.c1),
                          (c2n.variables, oracle.c2),
                          (agent.target_critic_networks[0].variables, oracle.t1),
                          (agent.target_critic_networks[1].variables, oracle.t2))
                for gi, (got_vars, want_vars) in enumerate(groups):
                    for k, (got, want) in enumerate(zip(got_vars, want_vars)):
                        err = _max_rel(got, want)
                        worst["param"] = max(worst["param"], err)
                        assert err <= TOL_PARAM, f"step {i} params {gi}/{k}: {err:.2e}"
                assert abs(float(agent.log_alpha) - float(oracle.log_alpha)) <= 1e-6
        finally:
            graph.disable_overlap()
        gt = graph.graphed_train(agent)
        assert gt.replays == STEPS - 2, "the train step did not go through the HIP graphs"
        assert collect.replays == STEPS - 2
        assert int(agent.train_step_counter.numpy()) == STEPS
        print(f"SAC configs[4] parity through the graphs over {STEPS} steps: worst loss rel err "
              f"{worst['loss']:.2e}, gradient relative L2 {worst['grad']:.2e}, parameter error "
              f"after one optimizer step {worst['param']:.2e} of max|p|; {flips} boundary flips")


def test_sac_stream_overlap_matches_single_stream(dev):
    """With overlap on, the train step replays as two graphs -- the critic update beside the
    collect step (both only read the actor), the actor / alpha / target updates after it -- and
    the sampler draws on its own lane.  Bit-identical to the same loop on one stream: parameters,
    targets, log_alpha, optimizer state and losses after every iteration block."""
    import bench_sac
    runs = []
    for overlap in (False, True):
        w = bench_sac.build(dev, envs=256, max_length=8, batch=64)
        agent, collect, lrn = w["agent"], w["collect"], w["learner"]
        it = iter(w["dataset"])
        if overlap:
            graph.enable_overlap(dev)
        try:
            tsx, snaps = None, []
            for i in range(30):
                tsx, _ = collect(tsx)
                li = lrn.run(iterations=1, iterator=it)
                if i % 6 == 5:
                    graph.join_lanes(dev)
                    torch.cuda.synchronize()
                    snaps.append([t.clone() for t in agent.replicated_state()] +
                                 [li.loss.clone(), tsx.observation.clone()])
            gt = graph.graphed_train(agent)
            assert gt.replays > 20
            runs.append(snaps)
        finally:
            graph.disable_overlap()
    for a, b in zip(*runs):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert torch.equal(x, y)

