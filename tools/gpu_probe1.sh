#!/bin/bash
# launch-floor probe: host enqueue cost vs device-only cost of a dependent launch chain
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
nproc; lscpu | grep -E "Model name|MHz" | head -3
timeout 120 tools/build/launch_probe 2000 > gpurun_out/r2_probe2.log 2>&1
grep -E "^E |^F |^G |^D |nonblocking" gpurun_out/r2_probe2.log
