#!/usr/bin/env python
"""Folds the rocprofv3 --pmc csv files of tools/pmc_r02.sh into one JSON per kernel:
  hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950 FETCH_SIZE correction,
                                                                MI355X_MICROARCH.md section HBM)
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles * 1024 SIMDs), kernel cycles taken from
              SQ_BUSY_CYCLES / 32 shader engines; mfma_busy_at_2p4GHz uses the traced duration at
              the peak clock instead (a lower bound).  SQ_VALU_MFMA_BUSY_CYCLES is exact: it equals
              16 (bf16 16x16x32) or 64 (fp32 32x32x2) cycles x the MFMA count of the launch.
  valu_insts, wait_inst_any / active_inst_any / wave_cycles (quad-cycles), LDS conflict share.
Per case the DOMINANT kernel of that case (largest summed GRBM_GUI_ACTIVE) is reported together
with every other kernel of the package the case launched (pre-passes, reduces)."""
import csv
import glob
import json
import os
import sys

N_SIMD = 256 * 4


def load(root, case, tag):
    """{kernel name: {counter: [values per dispatch]}}"""
    out = {}
    for path in glob.glob(os.path.join(root, f"{case}_{tag}", "**", "*counter_collection.csv"),
                          recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"]
                if not k.startswith(("aa_", "void aa_")):
                    continue
                out.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(
                    float(row["Counter_Value"]))
    return out


def durations(root, case, tag):
    """{kernel name: [duration ns per dispatch]} from the pass's kernel trace."""
    out = {}
    for path in glob.glob(os.path.join(root, f"{case}_{tag}", "**", "*kernel_trace.csv"),
                          recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "")
                if k.startswith(("aa_", "void aa_")):
                    out.setdefault(k, []).append(float(row["End_Timestamp"]) -
                                                 float(row["Start_Timestamp"]))
    return out


def mean_tail(v):
    v = v[1:] if len(v) > 2 else v        # first launch warms caches
    return sum(v) / len(v) if v else None


def main():
    root, out_path = sys.argv[1], sys.argv[2]
    cases = sorted({os.path.basename(p).rsplit("_", 1)[0] for p in glob.glob(os.path.join(root, "*_SQ"))})
    res = {}
    for case in cases:
        sq, fe, wr = load(root, case, "SQ"), load(root, case, "FETCH_SIZE"), load(root, case, "WRITE_SIZE")
        kernels = {}
        dur = durations(root, case, "FETCH_SIZE")     # (the lightest pass: one TCC counter)
        for k, c in sq.items():
            g = lambda n: mean_tail(c.get(n, []))
            act = g("GRBM_GUI_ACTIVE")
            e = {"launches_seen": len(c.get("GRBM_GUI_ACTIVE", [])), "GRBM_GUI_ACTIVE": act}
            for n in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU",
                      "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT",
                      "SQ_LDS_IDX_ACTIVE"):
                e[n] = g(n)
            # matrix-pipe busy cycles summed over all SIMDs / (kernel cycles x 1024 SIMDs).  The
            # kernel's own cycle count comes (a) from SQ_BUSY_CYCLES / 32 shader engines and (b)
            # from its traced duration at the 2.4 GHz peak clock (a LOWER bound on busy: the
            # clock under load is lower).  GRBM_GUI_ACTIVE spans the profiler's whole dispatch
            # window (>= 200 k cycles even for a 2 us kernel) and is not usable as denominator.
            d_ns = mean_tail(dur.get(k, []))
            e["duration_ns_traced"] = d_ns
            mb = e["SQ_VALU_MFMA_BUSY_CYCLES"]
            if mb is not None and e["SQ_BUSY_CYCLES"]:
                e["mfma_busy"] = mb / (e["SQ_BUSY_CYCLES"] / 32.0 * N_SIMD)
            if mb is not None and d_ns:
                e["mfma_busy_at_2p4GHz"] = mb / (d_ns * 2.4 * N_SIMD)
            if e["SQ_WAVE_CYCLES"]:
                if e["SQ_WAIT_INST_ANY"] is not None:
                    e["wait_inst_share"] = e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"]
                if e["SQ_ACTIVE_INST_ANY"] is not None:
                    e["active_inst_share"] = e["SQ_ACTIVE_INST_ANY"] / e["SQ_WAVE_CYCLES"]
            if e["SQ_LDS_IDX_ACTIVE"]:
                e["lds_conflict_share"] = (e["SQ_LDS_BANK_CONFLICT"] or 0.0) / e["SQ_LDS_IDX_ACTIVE"]
            f_ = mean_tail(fe.get(k, {}).get("FETCH_SIZE", []))
            w_ = mean_tail(wr.get(k, {}).get("WRITE_SIZE", []))
            if f_ is not None and w_ is not None:
                e["hbm_bytes_per_launch"] = (2.0 * f_ + w_) * 1024.0
                e["FETCH_SIZE_KiB_raw"], e["WRITE_SIZE_KiB_raw"] = f_, w_
            kernels[k] = e
        if kernels:
            dom = max(kernels, key=lambda k: (kernels[k]["GRBM_GUI_ACTIVE"] or 0) *
                      kernels[k]["launches_seen"])
            res[case] = {"dominant_kernel": dom, "kernels": kernels}
    json.dump({"cases": res,
               "method": "rocprofv3 --pmc <8 SQ counters> GRBM_GUI_ACTIVE / --pmc FETCH_SIZE / --pmc "
                         "WRITE_SIZE, three separate passes with --kernel-trace only, means over "
                         "launches 2..8 of tools/gemm_one.py <case>; mfma_busy = "
                         "SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 * 1024 SIMDs); bytes = "
                         "(2*FETCH_SIZE + WRITE_SIZE) * 1024"}, open(out_path, "w"), indent=1)
    for case, r in res.items():
        d = r["kernels"][r["dominant_kernel"]]
        print(f"{case:16s} {r['dominant_kernel'][:48]:48s} mfma_busy "
              f"{d.get('mfma_busy', float('nan')):.3f}  hbm {d.get('hbm_bytes_per_launch', 0) / 1e6:.1f} MB"
              f"  wait {d.get('wait_inst_share', float('nan')):.2f} lds_conf "
              f"{d.get('lds_conflict_share', float('nan')):.2f}")


if __name__ == "__main__":
    main()
