#!/bin/bash
# Round-2 counters for the roofline of the step's kernels, collected the way
# MI355X_MICROARCH.md prescribes: --pmc passes carry --kernel-trace only; FETCH_SIZE and WRITE_SIZE
# in SEPARATE passes (TCC slot limits); FETCH_SIZE doubled on gfx950.  One SQ pass for the issue /
# matrix-pipe counters.
# usage (on the GPU box): tools/pmc_r02.sh  -> gpurun_out/pmc_r02.json (+ raw csv under gpurun_out/pmc2)
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${AA_PMC_OUT:-gpurun_out/pmc2}
EXTRA=${AA_PMC_ARGS:-}      # e.g. "--batch 512"
mkdir -p $OUT
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for case in ${AA_PMC_CASES:-conv23.fwd conv2.dX conv3.dX conv2.dW conv3.dW fc1.fwd fc1.dW fc1.dX replay.get_next conv1.fwd}; do
  rocprofv3 --pmc $SQ GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${case}_SQ -o r -- \
    python tools/gemm_one.py $case --reps 8 --no-time $EXTRA > $OUT/${case}_SQ.log 2>&1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/${case}_$ctr -o r -- \
      python tools/gemm_one.py $case --reps 8 --no-time $EXTRA > $OUT/${case}_$ctr.log 2>&1
  done
done
python tools/pmc_r02_summary.py $OUT ${AA_PMC_JSON:-gpurun_out/pmc_r02.json}
