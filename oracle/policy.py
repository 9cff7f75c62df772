"""TEST INFRASTRUCTURE ONLY -- never imported by agents_amd/ or by bench.py's timed region.

CPU restatement (numpy) of the discrete collect policies' action selection:

  BoltzmannPolicy._apply_temperature / _distribution   tf_agents/policies/boltzmann_policy.py:83-101
      logits = wrapped logits / temperature
  QPolicy._distribution                                 tf_agents/policies/q_policy.py:150-194
      masked actions get logits = float32 min (:175-180)
  tfp Categorical.sample is a third-party, unseeded draw: the random STREAM is ours
  (agents_amd/csrc/rollout.hip: aa_boltzmann_kernel -- Philox4x32-10, counter (row, call), key =
  seed, word 0 -> u in [0, 1)); what is pinned here is the mapping u -> action: inverse CDF of
  softmax(logits), accumulated in float64 in action order.  Parity of the stream: unpinned
  (nothing in the reference fixes it); the distribution itself is pinned by
  boltzmann_policy_test.py:87-123 (logits 4.0 / 5.5 -> 8.0 / 11.0 at temperature 0.5, mode 1)."""
import numpy as np

from oracle import philox

FLT_MAX = np.finfo(np.float32).max


def boltzmann_logits(q, temperature, mask=None):
    q = np.asarray(q, np.float32)
    logits = (q / np.float32(temperature)).astype(np.float32)
    if mask is not None:
        logits = np.where(np.asarray(mask) != 0, logits, -FLT_MAX).astype(np.float32)
    return logits


def boltzmann_actions(q, temperature, seed, call, mask=None, action_min=0):
    """Actions [B] of aa_boltzmann_action for call number `call` of a policy seeded `seed`."""
    logits = boltzmann_logits(q, temperature, mask)
    B, A = logits.shape
    ok = np.ones((B, A), bool) if mask is None else (np.asarray(mask) != 0)
    b = np.arange(B, dtype=np.uint64)
    r0, _, _, _ = philox.philox4x32_10(b & philox.MASK, b >> np.uint64(32),
                                       int(call) & 0xFFFFFFFF, (int(call) >> 32) & 0xFFFFFFFF,
                                       int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    u = philox.u01(r0).astype(np.float64)
    out = np.zeros(B, np.int64)
    for i in range(B):
        lmax = np.float32(-FLT_MAX)
        for a in range(A):
            if logits[i, a] > lmax:
                lmax = logits[i, a]
        p = [np.exp(np.float64(logits[i, a]) - np.float64(lmax)) if ok[i, a] else None
             for a in range(A)]
        total = 0.0
        for a in range(A):
            if p[a] is not None:
                total += p[a]
        thr = u[i] * total
        allowed = [a for a in range(A) if ok[i, a]]
        act = allowed[-1] if allowed else 0
        cum = 0.0
        for a in allowed:
            cum += p[a]
            if cum > thr:
                act = a
                break
        out[i] = action_min + act
    return out
