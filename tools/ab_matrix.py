#!/usr/bin/env python
"""Round-robin A/B/C/... of environment-variable combinations on ONE box (boxes of the pool differ
by +-8 % on the same tree, so only runs of one gpurun call are comparable): every combination is
run once per round, `--rounds` rounds, `bench.py --no-breakdown --no-cpu-baseline`.

  python tools/ab_matrix.py --rounds 2 --steps 300 "base:" "pw:AA_PREPARED_WEIGHTS=1" \
      "pw_pair:AA_PREPARED_WEIGHTS=1,AA_PW_KINDS=pair"
Each argument is  name:VAR=value,VAR=value  (empty after the colon = the defaults); the pseudo
variable ARGS=--flag+--other adds bench.py arguments to that combination."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(envs, steps, extra):
    env = dict(os.environ)
    env.update(envs)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps),
                          "--no-cpu-baseline", "--no-breakdown", "--no-other-configs"] + extra,
                         env=env, capture_output=True, text=True)
    for line in reversed(out.stdout.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)["ms_per_step"]
    raise RuntimeError(out.stderr[-2000:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("combos", nargs="+")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--bench-args", default="")
    args = ap.parse_args()
    combos = []
    for c in args.combos:
        name, _, rest = c.partition(":")
        envs = dict(kv.split("=", 1) for kv in rest.split(",") if kv)
        combos.append((name, envs))
    res = {name: [] for name, _ in combos}
    for _ in range(args.rounds):
        for name, envs in combos:
            envs = dict(envs)
            own = envs.pop("ARGS", "").replace("+", " ").split()   # e.g. ARGS=--no-overlap
            try:
                ms = run(envs, args.steps, args.bench_args.split() + own)
            except RuntimeError as e:
                print(f"{name}: FAILED {str(e)[-300:]}", flush=True)
                continue
            res[name].append(ms)
            print(f"{name}: {ms:.4f} ms", flush=True)
    base = None
    for name, _ in combos:
        v = res[name]
        if not v:
            continue
        m = sum(v) / len(v)
        base = base or m
        print(f"{name:24s} mean {m:.4f} ms  ({m / base:.3f} x first)  runs {['%.4f' % x for x in v]}")


if __name__ == "__main__":
    main()
