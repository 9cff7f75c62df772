"""GPU: `bench.py --gpus 2` end to end, exactly as the driver launches it (`python -m
torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ...`), with the two
ranks sharing the one GPU of the test box over gloo (AA_BENCH_BACKEND / AA_BENCH_SHARE_GPU are
development aids of bench.py for exactly this).  Everything but the transport is the path the
8-GPU run takes -- per-rank replay shard, Learner hooks, bucketed all-reduce between the train
graphs, barriers, max-over-ranks timing, rank-0 JSON -- so that run cannot die on a bench-side bug.
Asserts the contract of the JSON line."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(400)
def test_bench_two_ranks_prints_a_well_formed_line(dev):
    env = dict(os.environ, AA_BENCH_BACKEND="gloo", AA_BENCH_SHARE_GPU="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3",
           "--max-length", "16", "--no-breakdown"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=380)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line, from rank 0"
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in out, key
    assert out["n_gpus"] == 2 and out["steps"] == 12 and out["warmup"] == 3
    assert out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["config"]["global_batch"] == 512 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["envs_per_gpu"] == 256
    assert out["captures_in_timed_region"] == 0
    # whole-job aggregate: transitions trained per second over BOTH ranks
    steps_per_sec = 1e3 / out["ms_per_step"]
    assert abs(out["value"] - steps_per_sec * 256 * 2) <= 1e-6 * out["value"]
    assert "cpu_baseline" not in out and "other_configs" not in out      # N = 1 only
    assert "process group up: backend gloo, 2 ranks" in r.stderr


@pytest.mark.timeout(400)
def test_plain_python_bench_gpus_2_launches_its_own_ranks(dev):
    """The driver's command line is `python bench.py --gpus N ...` with no WORLD_SIZE: bench.py must
    become the launcher (torch.distributed.run, one rank per GPU) instead of exiting."""
    env = dict(os.environ, AA_BENCH_BACKEND="gloo", AA_BENCH_SHARE_GPU="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8",
           "--warmup", "2", "--max-length", "16", "--no-breakdown", "--steady-steps", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=380)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 8
    assert "re-launching under torch.distributed.run (2 ranks" in r.stderr
    assert len(lines[0]) < 4096


def test_dqn_line_reports_the_collectives(dev):
    """The 2-rank DQN line says what the data path exchanged: two all-reduces per step (the dense
    tail's bucket and the conv head's) + one for the LossInfo scalars, 6.75 MB of gradients, and the
    stream time left exposed -- the fields that make a first real scaling run interpretable."""
    env = dict(os.environ, AA_BENCH_BACKEND="gloo", AA_BENCH_SHARE_GPU="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8",
           "--warmup", "2", "--max-length", "16", "--no-breakdown", "--steady-steps", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=380)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    c = out["collectives"]
    assert c["ranks"] == 2 and c["backend"] == "gloo"
    assert c["allreduce_calls_per_step"] == 3
    n_params = 1687206        # Mnih-15 Q-network with 6 actions (fp32)
    assert 4 * n_params <= c["allreduce_bytes_per_step"] <= 4 * n_params + 4096
    assert c["allreduce_exposed_ms_per_step"] >= 0.0


@pytest.mark.timeout(600)
@pytest.mark.parametrize("config", ["sac", "ppo"])
def test_other_configs_run_on_two_ranks(dev, config):
    """`bench.py --config sac|ppo --gpus 2` (BASELINE.json configs[4] is an 8-GPU configuration):
    launches its own two ranks, trains data-parallel through the Learner's strategy, rank 0 prints
    one line with the whole-job aggregate and the collectives of a step."""
    env = dict(os.environ, AA_BENCH_BACKEND="gloo", AA_BENCH_SHARE_GPU="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    steps = "30" if config == "sac" else "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--gpus", "2",
           "--steps", steps, "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=580)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line, from rank 0"
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["scaling"] == "weak"
    assert out["config"]["parallelism"] == "dp2" and out["value"] > 0
    c = out["collectives"]
    assert c["ranks"] == 2 and c["allreduce_calls_per_step"] >= 1
    assert c["allreduce_bytes_per_step"] > 0 and c["allreduce_exposed_ms_per_step"] >= 0.0
    assert "cpu_baseline" not in out
    assert "re-launching under torch.distributed.run (2 ranks" in r.stderr
