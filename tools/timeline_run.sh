#!/bin/bash
# rocprofv3 kernel timeline of the DQN loop with the current defaults -> gpurun_out/<tag>_timeline.txt
# (+ the per-kernel table); usage (GPU box): tools/timeline_run.sh <tag> [ENV=val ...]
TAG=${1:-tl}; shift
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp && cd "$R"
mkdir -p gpurun_out; rm -rf /tmp/prof_$TAG
env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o r -- python bench.py --gpus 1 \
  --steps 300 --warmup 40 --no-cpu-baseline --no-breakdown --no-other-configs \
  > gpurun_out/${TAG}_prof.json 2> gpurun_out/${TAG}_prof.err
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
if [ -n "$DB" ]; then
  python tools/rocpd_stats.py "$DB" gpurun_out/${TAG}_kernel_stats.csv > /dev/null
  python tools/timeline.py "$DB" 200 gpurun_out/${TAG}_timeline.txt > /dev/null
fi
tail -1 gpurun_out/${TAG}_prof.json | cut -c1-300
