#!/bin/bash
# generic GPU-box driver: bash tools/gpu_call.sh <tag> <what...>   what in: tests newtests loop bench prof
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for what in "$@"; do
case $what in
  tests)    timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/${TAG}_tests.log;;
  newtests) timeout 1200 python -m pytest tests/test_gpu_benchcfg.py tests/test_gpu_parity.py -m gpu -q -s --durations=30 > gpurun_out/${TAG}_newtests.log 2>&1; echo "newtests rc=$?"; tail -60 gpurun_out/${TAG}_newtests.log;;
  loop)     timeout 300 python tools/loop_probe.py > gpurun_out/${TAG}_loop.log 2>&1; echo "loop rc=$?"; tail -3 gpurun_out/${TAG}_loop.log;;
  bench)    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-600;;
  benchfull) timeout 900 python bench.py > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?"; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-1500;;
  prof)     cd /tmp && export TMPDIR=/tmp
            timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_prof.log 2>&1; echo "prof rc=$?"
            cd $R
            DB=$(ls gpurun_out/${TAG}_prof/*/*_results.db 2>/dev/null | head -1)
            if [ -n "$DB" ]; then python tools/prof_summary.py $DB gpurun_out/${TAG}_kernels.csv "$TAG"; head -24 gpurun_out/${TAG}_kernels.csv | cut -c1-160; python tools/prof_by_grid.py $DB > gpurun_out/${TAG}_bygrid.txt 2>&1; rm -rf gpurun_out/${TAG}_prof; fi;;
  ktests)   timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/${TAG}_ktests.log 2>&1; echo "ktests rc=$?"; tail -4 gpurun_out/${TAG}_ktests.log;;
  benchq)   timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-pmc > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?"; python tools/bench_brief.py gpurun_out/${TAG}_bench.log;;
  benchold) LXO_STEP_KERNELS=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_benchold.log 2>&1; echo "benchold rc=$?"; tail -1 gpurun_out/${TAG}_benchold.log | cut -c1-300;;
  wstamps)  timeout 300 python tools/wgrad_stamps.py > gpurun_out/${TAG}_wgrad_stamps.log 2>&1; echo "wstamps rc=$?"; tail -12 gpurun_out/${TAG}_wgrad_stamps.log;;
  traffic)  cd /tmp && export TMPDIR=/tmp
            for CNT in FETCH_SIZE WRITE_SIZE; do
              (timeout 600 rocprofv3 --pmc $CNT --kernel-trace -d $R/gpurun_out/${TAG}_pmc_$CNT -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-secondary > $R/gpurun_out/${TAG}_pmc_$CNT.log 2>&1; echo "pmc $CNT rc=$?")
            done
            cd $R
            F=$(ls gpurun_out/${TAG}_pmc_FETCH_SIZE/*/*_results.db | head -1); Wd=$(ls gpurun_out/${TAG}_pmc_WRITE_SIZE/*/*_results.db | head -1)
            python tools/pmc_traffic.py $F $Wd gpurun_out/${TAG}_conv_traffic.json conv_halo2wg_kernel; cat gpurun_out/${TAG}_conv_traffic.json | cut -c1-600
            rm -rf gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE;;
  *) echo "unknown $what";;
esac
done
