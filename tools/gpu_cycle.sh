#!/bin/bash
# One measurement cycle on the GPU box: parity tests, bench line, rocprof kernel summary.
# usage (via gpurun): bash tools/gpu_cycle.sh <tag> [steps]
TAG=${1:-cycle}; STEPS=${2:-4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export PYTHONUNBUFFERED=1
cd $R
[ -n "$SKIPTESTS" ] || (timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?")
tail -4 gpurun_out/${TAG}_tests.log
(timeout 600 python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?")
tail -2 gpurun_out/${TAG}_bench.log
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_prof.log 2>&1; echo "prof rc=$?")
cd $R
DB=$(ls gpurun_out/${TAG}_prof/*/*_results.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/prof_summary.py $DB gpurun_out/${TAG}_kernels.csv "$TAG"; head -14 gpurun_out/${TAG}_kernels.csv | cut -c1-150; python tools/prof_by_grid.py $DB > gpurun_out/${TAG}_bygrid.txt 2>&1; python tools/prof_seq.py $DB conv_ 120 > gpurun_out/${TAG}_convseq.txt 2>&1; rm -rf gpurun_out/${TAG}_prof; fi
