#!/bin/bash
# HBM traffic per launch of the dominant kernels, from rocprofv3 PMC counters, collected as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slot limits),
# kernel-trace only alongside; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B).
# usage (on the GPU box): tools/pmc_traffic.sh  -> gpurun_out/pmc_traffic.json (+ raw csv)
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/pmc
mkdir -p $OUT
for case in conv1.fwd conv1.dW conv23.fwd replay.gather; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/${case}_$ctr -o r -- \
      python tools/gemm_one.py $case --reps 10 --no-time > $OUT/${case}_$ctr.log 2>&1
  done
done
python tools/pmc_traffic_summary.py $OUT gpurun_out/pmc_traffic.json
