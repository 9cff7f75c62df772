#!/bin/bash
# rocprofv3 kernel timeline of the SAC loop (tools/bench_sac.py): the last dispatches with queue ids.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp && cd "$R"
mkdir -p gpurun_out
rm -rf /tmp/prof_sac_tl
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_sac_tl -o r -- python tools/bench_sac.py --iters 120 \
  > gpurun_out/sac_tl_run.json 2> gpurun_out/sac_tl_run.err
DB=$(find /tmp/prof_sac_tl -name "*.db" | head -1)
[ -n "$DB" ] && python tools/timeline.py "$DB" 150 gpurun_out/sac_timeline.txt > /dev/null
tail -1 gpurun_out/sac_tl_run.json | cut -c1-300
