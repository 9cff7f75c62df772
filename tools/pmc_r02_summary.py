#!/usr/bin/env python
"""Folds the rocprofv3 --pmc csv files of tools/pmc_r02.sh into one JSON per kernel:
  hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024   (gfx950 FETCH_SIZE correction,
                                                                MI355X_MICROARCH.md section HBM)
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs)   (the gfx94x MfmaUtil
              formula: matrix-pipe busy cycles over all SIMDs / available SIMD cycles)
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
        for k, c in sq.items():
            g = lambda n: mean_tail(c.get(n, []))
            act = g("GRBM_GUI_ACTIVE")
            e = {"launches_seen": len(c.get("GRBM_GUI_ACTIVE", [])), "GRBM_GUI_ACTIVE": act}
            for n in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU",
                      "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT",
                      "SQ_LDS_IDX_ACTIVE"):
                e[n] = g(n)
            if act and e["SQ_VALU_MFMA_BUSY_CYCLES"] is not None:
                e["mfma_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (act * N_SIMD)
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
                         "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs); bytes = "
                         "(2*FETCH_SIZE + WRITE_SIZE) * 1024"}, open(out_path, "w"), indent=1)
    for case, r in res.items():
        d = r["kernels"][r["dominant_kernel"]]
        print(f"{case:16s} {r['dominant_kernel'][:48]:48s} mfma_busy "
              f"{d.get('mfma_busy', float('nan')):.3f}  hbm {d.get('hbm_bytes_per_launch', 0) / 1e6:.1f} MB"
              f"  wait {d.get('wait_inst_share', float('nan')):.2f} lds_conf "
              f"{d.get('lds_conflict_share', float('nan')):.2f}")


if __name__ == "__main__":
    main()
