#!/bin/bash
# Everything profiles/r06_* is made of, in one gpurun call (about 5 GPU-minutes):
#   1. bench.py at the driver's command line (default flags) -> the JSON line the judge sees, with the
#      in-loop kernel table it measured (--trace-out)
#   2. rocprofv3 --kernel-trace --stats of the DQN loop, of the PPO and of the SAC configuration
#      -> per-kernel tables (+ the DQN timeline)
#   3. rocprofv3 --pmc passes (traffic + matrix-pipe counters) of the step's kernels in isolation
# usage (GPU box): tools/profile_r06.sh <tag>    -> gpurun_out/<tag>_*
set -u
TAG=${1:-r06}
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp && cd "$R"
mkdir -p gpurun_out
python bench.py --trace-out gpurun_out/${TAG}_bench_inloop_kernels.csv \
  > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo '{"cmd": "python bench.py --trace-out gpurun_out/'${TAG}'_bench_inloop_kernels.csv"}' \
  > gpurun_out/${TAG}_bench_driver_cmdline.json
rm -rf /tmp/prof_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- python bench.py --gpus 1 --steps 300 \
  --warmup 40 --no-cpu-baseline --no-breakdown --no-other-configs \
  > gpurun_out/${TAG}_bench_prof.json 2> gpurun_out/${TAG}_bench_prof.err
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
if [ -n "$DB" ]; then
  python tools/rocpd_stats.py "$DB" gpurun_out/${TAG}_bench_kernel_stats.csv > /dev/null
  python tools/timeline.py "$DB" 160 gpurun_out/${TAG}_bench_timeline.txt > /dev/null
fi
# GPU-side event timeline of the SAC loop (who waits for whom: tools/bench_sac.py --timeline)
python tools/bench_sac.py --iters 200 --timeline 40 > gpurun_out/${TAG}_sac_timeline.json \
  2> gpurun_out/${TAG}_sac_timeline.txt
# in-kernel timelines of the fp32 MFMA GEMM on the fc1 shapes + the memory-pipeline micro-benchmark
[ -x tools/_bin/fc1_probe ] && tools/_bin/fc1_probe > gpurun_out/${TAG}_fc1_probe.txt 2>&1
[ -x tools/_bin/fc1_mem_probe ] && tools/_bin/fc1_mem_probe > gpurun_out/${TAG}_fc1_mem_probe.txt 2>&1
for cfg in ppo sac; do
  rm -rf /tmp/prof_${TAG}_$cfg
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_$cfg -o r -- python tools/bench_$cfg.py \
    > gpurun_out/${TAG}_${cfg}_prof.json 2> gpurun_out/${TAG}_${cfg}_prof.err
  DB=$(find /tmp/prof_${TAG}_$cfg -name "*.db" | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py "$DB" gpurun_out/${TAG}_${cfg}_kernel_stats.csv > /dev/null
done
# the DQN loop on the prioritized buffer (td_error -> priorities through the Learner hook)
python bench.py --replay prioritized --steps 200 > gpurun_out/${TAG}_bench_prioritized.json \
  2> gpurun_out/${TAG}_bench_prioritized.err
python tools/pmc_other.py > gpurun_out/${TAG}_pmc_other.log 2>&1
tools/pmc_r02.sh > gpurun_out/${TAG}_pmc.log 2>&1
cp gpurun_out/pmc_r02.json gpurun_out/${TAG}_pmc.json
tail -5 gpurun_out/${TAG}_pmc.log
tail -1 gpurun_out/${TAG}_bench.json | cut -c1-600
