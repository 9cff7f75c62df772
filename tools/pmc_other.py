#!/usr/bin/env python
"""HBM traffic (and matrix-pipe busy share) per launch of the dominant kernels of the PPO and SAC
loops, from rocprofv3 --pmc passes collected as MI355X_MICROARCH.md prescribes: one counter set per
pass, --kernel-trace only alongside; FETCH_SIZE doubled (gfx950 tallies the 128-B requests of wide
coalesced reads at 64 B), KiB -> bytes.

    python tools/pmc_other.py [out.json]      (on the GPU box; default gpurun_out/r06_pmc_other.json)

`bench.py --config ppo|sac` reads the committed copy (profiles/r06_pmc_other.json) for the
`traffic` field of its roofline."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {
    # name -> (command, kernel substring)
    "mlp_wide_fwd": ([sys.executable, os.path.join(ROOT, "tools", "bench_sac.py"), "--iters", "30",
                      "--max-length", "64"], "aa_mlp_wide_fwd_kernel"),
    "ppo_fused_step": ([sys.executable, os.path.join(ROOT, "tools", "bench_ppo.py"), "--iters", "1",
                        "--epochs", "2"], "aa_ppo_fused_step_kernel"),
}
PASSES = (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES"])


def run_pass(cmd, counters):
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    tmp = tempfile.mkdtemp(prefix="aa_pmc_", dir="/tmp")
    full = [prof, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", tmp,
                                         "-o", "r", "--"] + cmd
    r = subprocess.run(full, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                       capture_output=True, text=True, timeout=900)
    rows = []
    for path in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            rows += list(csv.DictReader(fh))
    shutil.rmtree(tmp, ignore_errors=True)
    if r.returncode != 0 and not rows:
        raise RuntimeError(f"{' '.join(full)} failed: {r.stderr[-600:]}")
    return rows


def mean_of(rows, sub, ctr):
    v = [float(r["Counter_Value"]) for r in rows
         if sub in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    v = v[2:] if len(v) > 4 else v          # the first launches warm the caches
    return (sum(v) / len(v), len(v)) if v else (None, 0)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out",
                                                             "r06_pmc_other.json")
    res = {}
    for name, (cmd, sub) in CASES.items():
        rec = {"kernel": sub, "command": " ".join(os.path.relpath(c, ROOT) if os.path.isabs(c)
                                                  and c.startswith(ROOT) else c for c in cmd)}
        vals = {}
        for counters in PASSES:
            rows = run_pass(cmd, counters)
            for c in counters:
                vals[c], rec.setdefault("launches_counted", {})[c] = mean_of(rows, sub, c)
        if vals.get("FETCH_SIZE") is not None and vals.get("WRITE_SIZE") is not None:
            rec["FETCH_SIZE_KiB_raw"], rec["WRITE_SIZE_KiB_raw"] = vals["FETCH_SIZE"], \
                vals["WRITE_SIZE"]
            rec["bytes_per_launch"] = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
        if vals.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and vals.get("SQ_BUSY_CYCLES"):
            # SQ_BUSY_CYCLES sums over 32 shader engines' SQs; 1,024 SIMDs can issue MFMAs
            rec["mfma_busy"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / \
                (vals["SQ_BUSY_CYCLES"] / 32.0 * 1024.0)
        res[name] = rec
        print(name, json.dumps(rec), flush=True)
    res["method"] = ("rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES "
                     "SQ_BUSY_CYCLES in separate passes with --kernel-trace only; mean over the "
                     "kernel's launches of the command (first two dropped); bytes = (2 x "
                     "FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 FETCH_SIZE correction, "
                     "MI355X_MICROARCH.md); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                     "(SQ_BUSY_CYCLES / 32 x 1024)")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
