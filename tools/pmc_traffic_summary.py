#!/usr/bin/env python
"""Folds the rocprofv3 --pmc csv files written by tools/pmc_traffic.sh into
{"bytes_per_launch": {bench kernel name: HBM bytes}} (FETCH_SIZE x2 + WRITE_SIZE, KiB -> bytes;
MI355X_MICROARCH.md §HBM: gfx950 FETCH_SIZE reports half of a wide coalesced read)."""
import csv
import glob
import json
import os
import sys

CASES = {"conv1.fwd": ("conv1.fwd(u8)", "aa_conv_u8_bf16x3_kernel"),
         "conv1.dW": ("conv1.dW(u8,+bias grad)", "aa_conv_u8_dw_bf16x3_kernel"),  # main kernel only
         "conv23.fwd": ("conv2+conv3.fwd(fused)", "aa_conv_pair_kernel"),
         "conv2.fwd": ("conv2.fwd", "aa_gemm"),
         "replay.gather": ("replay.get_next(sample+gather 512 rows)", "aa_rb_gather")}


def counter_mean(root, case, ctr, kernel_substr):
    vals = []
    for path in glob.glob(os.path.join(root, f"{case}_{ctr}", "**", "*counter_collection.csv"),
                          recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                    vals.append(float(row["Counter_Value"]))
    if not vals:
        return None
    vals = vals[1:] if len(vals) > 2 else vals     # first launch warms the caches
    return sum(vals) / len(vals)


def main():
    root, out = sys.argv[1], sys.argv[2]
    res, detail = {}, {}
    for case, (name, sub) in CASES.items():
        fe = counter_mean(root, case, "FETCH_SIZE", sub)
        wr = counter_mean(root, case, "WRITE_SIZE", sub)
        if fe is None or wr is None:
            continue
        res[name] = (2.0 * fe + wr) * 1024.0
        detail[name] = {"FETCH_SIZE_KiB_raw": fe, "WRITE_SIZE_KiB_raw": wr,
                        "fetch_correction": 2.0}
    json.dump({"bytes_per_launch": res, "detail": detail,
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, mean "
                         "over launches 2..10 of tools/gemm_one.py; bytes = (2*FETCH_SIZE + "
                         "WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)"},
              open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
