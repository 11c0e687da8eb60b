#!/bin/bash
# one PMC pass: L2 hits / misses of the decoder kernels inside real training steps -> gpurun_out/<tag>_l2.json
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $R/gpurun_out/${TAG}_pmc_l2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-secondary > $R/gpurun_out/${TAG}_pmc_l2.log 2>&1; echo "pmc l2 rc=$?")
cd $R
DB=$(ls gpurun_out/${TAG}_pmc_l2/*/*_results.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/pmc_l2.py $DB gpurun_out/${TAG}_l2.json; rm -rf gpurun_out/${TAG}_pmc_l2; else tail -5 gpurun_out/${TAG}_pmc_l2.log; fi
