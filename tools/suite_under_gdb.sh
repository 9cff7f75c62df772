#!/bin/bash
# The full GPU suite under rocgdb: a native backtrace of every thread if the process faults
# (round 6: hipGraphLaunch segfault in tests/test_gpu_opt_slabs.py after ~670 tests).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
/opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "handle SIGUSR1 nostop noprint pass" \
  -ex "handle SIGPIPE nostop noprint pass" -ex run -ex "bt 40" -ex "info sharedlibrary amdhip" \
  -ex "thread apply all bt 12" \
  --args python -X faulthandler -m pytest tests -x -q -m gpu "$@" > gpurun_out/suite_gdb.log 2>&1
tail -c 6000 gpurun_out/suite_gdb.log
