#!/bin/bash
# End-of-round evidence: bench line (default flags), rocprofv3 kernel summary of the same command, PMC traffic of the
# dominant conv kernel (separate passes).  usage (via gpurun): bash tools/gpu_final.sh <tag>
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export PYTHONUNBUFFERED=1
cd $R
(timeout 600 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?"); tail -1 gpurun_out/${TAG}_bench.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_prof.log 2>&1; echo "prof rc=$?")
cd $R
DB=$(ls gpurun_out/${TAG}_prof/*/*_results.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/prof_summary.py $DB gpurun_out/${TAG}_kernels.csv "$TAG"; python tools/prof_by_grid.py $DB > gpurun_out/${TAG}_bygrid.txt 2>&1; rm -rf gpurun_out/${TAG}_prof; fi
tail -1 gpurun_out/${TAG}_prof.log > gpurun_out/${TAG}_bench_line_profiled.json
cd /tmp
for CNT in FETCH_SIZE WRITE_SIZE; do
  (timeout 300 rocprofv3 --pmc $CNT --kernel-trace -d $R/gpurun_out/${TAG}_pmc_$CNT -- python $R/tools/pmc_conv.py > $R/gpurun_out/${TAG}_pmc_$CNT.log 2>&1; echo "pmc $CNT rc=$?")
done
cd $R
F=$(ls gpurun_out/${TAG}_pmc_FETCH_SIZE/*/*_results.db | head -1); Wd=$(ls gpurun_out/${TAG}_pmc_WRITE_SIZE/*/*_results.db | head -1)
python tools/pmc_traffic.py $F $Wd gpurun_out/${TAG}_conv_traffic.json conv_halo2wg_kernel
rm -rf gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE
