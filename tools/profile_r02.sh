#!/bin/bash
# Everything profiles/r02_* is made of, in one gpurun call (about 3 GPU-minutes):
#   1. bench.py at the driver's command line and at the builder's long one (JSON lines)
#   2. rocprofv3 --kernel-trace --stats of the same bench command -> per-kernel table + timeline
#   3. rocprofv3 --pmc passes (traffic + matrix-pipe counters) of the step's kernels in isolation
# usage (GPU box): tools/profile_r02.sh <tag>    -> gpurun_out/<tag>_*
set -u
TAG=${1:-r02}
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp && cd "$R"
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_20.json 2> gpurun_out/${TAG}_bench_20.err
python bench.py --gpus 1 --steps 300 --warmup 40 --no-cpu-baseline > gpurun_out/${TAG}_bench_300.json 2> gpurun_out/${TAG}_bench_300.err
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- python bench.py --gpus 1 --steps 300 --warmup 40 \
  --no-cpu-baseline --no-breakdown > gpurun_out/${TAG}_bench_prof.json 2> gpurun_out/${TAG}_bench_prof.err
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
if [ -n "$DB" ]; then
  python tools/rocpd_stats.py "$DB" gpurun_out/${TAG}_bench_kernel_stats.csv > /dev/null
  python tools/timeline.py "$DB" 160 gpurun_out/${TAG}_bench_timeline.txt > /dev/null
fi
tools/pmc_r02.sh > gpurun_out/${TAG}_pmc.log 2>&1
cp gpurun_out/pmc_r02.json gpurun_out/${TAG}_pmc.json
tail -12 gpurun_out/${TAG}_pmc.log
tail -3 gpurun_out/${TAG}_bench_20.json | cut -c1-400
