"""numpy restatement of the synthetic vector env's Philox stream (TEST INFRASTRUCTURE).

The env is this package's own definition (the reference's RandomTFEnvironment draws from an
unseeded TF stream: tf_agents/environments/random_tf_environment.py); the contract it must honour
comes from the reference: StepType FIRST/MID/LAST = 0/1/2, restart -> reward 0 discount 1,
termination -> discount 0 (tf_agents/trajectories/time_step.py:135-348), and a LAST step is
followed by a reset regardless of the action (tf_agents/environments/py_environment.py:233-239).
Counter layout as in agents_amd/csrc/rollout.hip::aa_vecenv_step_kernel.
"""
import numpy as np

from oracle import philox

f32 = np.float32


def step(cur_step_type, B, obs_elems, obs_kind, lo, hi, p_end, seed, step_index, force_first):
    s_lo, s_hi = step_index & 0xFFFFFFFF, (step_index >> 32) & 0xFFFFFFFF
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    b = np.arange(B, dtype=np.uint64)
    per = 16 if obs_kind == "u8" else 4
    chunks = (obs_elems + per - 1) // per
    j = np.arange(chunks, dtype=np.uint64)
    x = philox.philox4x32_10(j[None, :], b[:, None], s_lo, s_hi, k0, k1)
    words = np.stack(x, axis=-1)  # [B, chunks, 4]
    if obs_kind == "u8":
        obs = words.astype("<u4").view(np.uint8).reshape(B, chunks * 16)[:, :obs_elems]
    else:
        u = philox.u01(words.reshape(B, chunks * 4)[:, :obs_elems])
        obs = (f32(lo) + (u * f32(f32(hi) - f32(lo))).astype(f32)).astype(f32)
    h = philox.philox4x32_10(np.uint64(0xFFFFFFFF), b, s_lo, s_hi, k0, k1)
    end = philox.u01(h[0]) < f32(p_end)
    u = philox.u01(h[1])
    rew = np.where(u < f32(0.05), f32(-1), np.where(u < f32(0.95), f32(0), f32(1))).astype(f32)
    if force_first:
        reset = np.ones(B, bool)
    else:
        reset = np.asarray(cur_step_type) == 2
    st = np.where(reset, 0, np.where(end, 2, 1)).astype(np.int32)
    reward = np.where(reset, f32(0), rew).astype(f32)
    discount = np.where(reset, f32(1), np.where(end, f32(0), f32(1))).astype(f32)
    return st, reward, discount, obs
