#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 200 tools/build/launch_probe 500 > gpurun_out/r2_probe4.log 2>&1
grep -E "^I" gpurun_out/r2_probe4.log
