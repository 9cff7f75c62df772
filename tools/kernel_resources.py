#!/usr/bin/env python
"""Static register / LDS / occupancy table of every kernel of the library (no GPU needed):
compiles each csrc/*.hip with -Rpass-analysis=kernel-resource-usage and prints one row per kernel;
rows with scratch or spills are flagged.  Run before any GPU time is spent on a kernel change
(DESIGN.md "Measurement hygiene": a hoisted load once cost the second wave per SIMD).
  python tools/kernel_resources.py [file.hip ...] [--all]      (default: occupancy < 8 only)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from agents_amd import _build  # noqa: E402

KEYS = ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "VGPRs Spill",
        "LDS Size [bytes/block]")


def rows(src, flags):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "agents_amd", "csrc"),
           "-Wno-unused-function", *flags, "-c", src, "-o", os.devnull,
           "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    for line in err.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
            if m.group(1).strip() == KEYS[-1]:
                yield cur
                cur = None


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names),
                         capture_output=True, text=True).stdout.splitlines()
    return out if len(out) == len(names) else names


def main():
    show_all = "--all" in sys.argv
    want = [a for a in sys.argv[1:] if not a.startswith("--")]
    for name, flags in _build.SOURCES:
        if want and name not in want:
            continue
        rs = list(rows(os.path.join(ROOT, "agents_amd", "csrc", name), flags))
        for r, dn in zip(rs, demangle([r["name"] for r in rs])):
            occ = int(r.get(KEYS[3], "0"))
            bad = r.get(KEYS[2], "0") != "0" or r.get(KEYS[4], "0") != "0"
            if not (show_all or bad or occ < 8):
                continue
            dn = re.sub(r"\(.*", "", dn)[:64]
            print(f"{'!!' if bad else '  '} {name:22s} {dn:64s} V {r.get('VGPRs', '?'):>3s} "
                  f"A {r.get('AGPRs', '?'):>3s} occ {occ} scratch {r.get(KEYS[2], '?')} "
                  f"spill {r.get(KEYS[4], '?')} lds {r.get(KEYS[5], '?')}")


if __name__ == "__main__":
    main()
