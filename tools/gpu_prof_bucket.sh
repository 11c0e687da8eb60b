#!/bin/bash
# bash tools/gpu_prof_bucket.sh <tag> <HxW> <B>: rocprofv3 kernel trace of the training step of one real bucket -> gpurun_out/<tag>_kernels.csv / _timeline.txt
TAG=$1; SH=$2; B=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
RB_SHAPES=$SH RB_BATCHES=$B timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -- python $R/tools/real_buckets.py ${TAG}x 20 > $R/gpurun_out/${TAG}_prof.log 2>&1; echo "prof rc=$?"
cd $R
DB=$(ls gpurun_out/${TAG}_prof/*/*_results.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/prof_summary.py $DB gpurun_out/${TAG}_kernels.csv "$TAG"; head -45 gpurun_out/${TAG}_kernels.csv | cut -c1-170; python tools/prof_timeline.py $DB > gpurun_out/${TAG}_timeline.txt 2>&1; tail -90 gpurun_out/${TAG}_timeline.txt | cut -c1-170; rm -rf gpurun_out/${TAG}_prof; fi
