#!/bin/bash
# round-6 GPU-box driver: bash tools/gpu_r6.sh <tag> <what...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for what in "$@"; do
case $what in
  alltests) timeout 2400 python -m pytest tests -m gpu -q -rP > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" gpurun_out/${TAG}_tests.log | tail -3; grep -E "^(FAILED|ERROR)" gpurun_out/${TAG}_tests.log | head -20;;
  tests)    timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/${TAG}_tests.log;;
  t:*)      f=${what#t:}; timeout 1200 python -m pytest tests/$f -m gpu -q -s > gpurun_out/${TAG}_${f%.py}.log 2>&1; echo "$f rc=$?"; grep -E "passed|failed|^B=|FAILED|Error|assert" gpurun_out/${TAG}_${f%.py}.log | tail -40;;
  smoke)    timeout 300 python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.log;;
  benchfull) timeout 1500 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-400; python tools/bench_brief.py gpurun_out/${TAG}_bench.log 2>/dev/null | head -40;;
  benchq)   timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-pmc > gpurun_out/${TAG}_benchq.log 2>&1; echo "benchq rc=$?"; python tools/bench_brief.py gpurun_out/${TAG}_benchq.log;;
  buckets)  timeout 1500 python tools/real_buckets.py ${TAG} ${RB_STEPS:-8} > gpurun_out/${TAG}_buckets.txt 2>&1; echo "buckets rc=$?"; cat gpurun_out/${TAG}_buckets.txt | cut -c1-200;;
  prof)     cd /tmp && export TMPDIR=/tmp
            LXO_ENC_OVERLAP=${PROF_OVERLAP:-0} timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_prof.log 2>&1; echo "prof rc=$?"
            cd $R
            DB=$(ls gpurun_out/${TAG}_prof/*/*_results.db 2>/dev/null | head -1)
            if [ -n "$DB" ]; then python tools/prof_summary.py $DB gpurun_out/${TAG}_kernels.csv "$TAG"; head -30 gpurun_out/${TAG}_kernels.csv | cut -c1-160; python tools/prof_by_grid.py $DB > gpurun_out/${TAG}_bygrid.txt 2>&1; python tools/prof_timeline.py $DB > gpurun_out/${TAG}_timeline.txt 2>&1; rm -rf gpurun_out/${TAG}_prof; fi;;
  sqconv)   bash tools/gpu_pmc_sq.sh ${TAG} > gpurun_out/${TAG}_sqconv.log 2>&1; echo "sqconv rc=$?"; tail -30 gpurun_out/${TAG}_sqconv.log;;
  stamps)   timeout 300 python tools/xdec_stamps.py > gpurun_out/${TAG}_xdec_stamps.txt 2>&1; echo "stamps rc=$?"; tail -15 gpurun_out/${TAG}_xdec_stamps.txt
            timeout 300 python tools/xdec_stamps_bwd.py > gpurun_out/${TAG}_xdec_bwd_stamps.txt 2>&1; echo "bwd stamps rc=$?"; tail -15 gpurun_out/${TAG}_xdec_bwd_stamps.txt;;
  *) if [ -f "tools/$what" ]; then timeout 900 python tools/$what > gpurun_out/${TAG}_${what%.py}.log 2>&1; echo "$what rc=$?"; tail -25 gpurun_out/${TAG}_${what%.py}.log; else echo "unknown $what"; fi;;
esac
done
