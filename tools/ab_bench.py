#!/usr/bin/env python
"""A/B of one environment toggle on ONE box: alternates `bench.py` runs with VAR=a and VAR=b and
prints the per-run ms/iteration plus the means.  Boxes of the pool differ by +-6 % on the same
tree, so only alternating runs inside one gpurun call are comparable.

  python tools/ab_bench.py AA_FUSE_CONV_PAIRS 1 0 [--pairs 3] [--steps 400]

Toggles that exist: AA_FUSE_CONV_PAIRS (conv2->conv3 forward in one launch), AA_CONV_DX_FRAME
(gather-form conv input gradient), AA_FIELD_SUMS (LossInfo sums from the loss launch),
AA_LAST_DW_ON_MAIN (first layer's weight gradient on the main stream)."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(var, val, steps):
    env = dict(os.environ)
    env[var] = val
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps),
                          "--no-cpu-baseline", "--no-breakdown", "--no-other-configs"], env=env, capture_output=True,
                         text=True)
    for line in reversed(out.stdout.strip().splitlines()):
        if line.startswith("{"):
            d = json.loads(line)
            print(f"    (host enqueue {d.get('host_enqueue_ms_per_step', 0):.4f} ms / step)",
                  flush=True)
            return d["ms_per_step"]
    raise RuntimeError(out.stderr[-2000:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("var")
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("more", nargs="*", help="further values of the toggle (round robin with a, b)")
    ap.add_argument("--pairs", type=int, default=3)
    ap.add_argument("--steps", type=int, default=400)
    args = ap.parse_args()
    vals = [args.a, args.b] + list(args.more)
    res = {v: [] for v in vals}
    for _ in range(args.pairs):
        for v in vals:
            ms = run(args.var, v, args.steps)
            res[v].append(ms)
            print(f"{args.var}={v}: {ms:.4f} ms", flush=True)
    ma, mb = (sum(res[v]) / len(res[v]) for v in (args.a, args.b))
    out = {"var": args.var, "mean_" + args.a: ma, "mean_" + args.b: mb, "b_over_a": mb / ma}
    for v in vals:
        out[v] = res[v]
        out["mean_" + v] = sum(res[v]) / len(res[v])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
