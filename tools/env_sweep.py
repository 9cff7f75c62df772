#!/usr/bin/env python
"""Round-robin A/B of HIP-runtime (or AA_*) environment settings on ONE box: every round runs
`bench.py` once per setting, in the given order; prints ms/iteration per run and the means.
A setting is `VAR=value[,VAR2=value2...]`; the literal `base` is the unmodified environment.

  python tools/env_sweep.py base HIP_FORCE_DEV_KERNARG=0 HIP_FORCE_DEV_KERNARG=1 --rounds 2
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(setting, steps, extra):
    env = dict(os.environ)
    if setting != "base":
        for kv in setting.split(","):
            k, v = kv.split("=", 1)
            env[k] = v
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps",
                              str(steps), "--no-cpu-baseline", "--no-breakdown",
                              "--no-other-configs", "--no-inloop"] + extra, env=env, capture_output=True, text=True,
                             timeout=150)
    except subprocess.TimeoutExpired:
        return None, "timed out after 150 s (a hang under this setting)"
    for line in reversed(out.stdout.strip().splitlines()):
        if line.startswith("{"):
            d = json.loads(line)
            return d["ms_per_step"], d.get("host_work_ms_per_step", 0.0)
    return None, out.stderr[-600:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("settings", nargs="+")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--bench-args", default="")
    args = ap.parse_args()
    res = {s: [] for s in args.settings}
    for _ in range(args.rounds):
        for s in args.settings:
            ms, hw = run(s, args.steps, args.bench_args.split())
            if ms is None:
                print(f"{s}: FAILED {hw}", flush=True)
                continue
            res[s].append(ms)
            print(f"{s}: {ms:.4f} ms (host work {hw:.4f})", flush=True)
    print(json.dumps({s: {"runs": v, "mean": sum(v) / len(v) if v else None}
                      for s, v in res.items()}))


if __name__ == "__main__":
    main()
