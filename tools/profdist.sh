#!/bin/bash
# Measurement aid (via gpurun): rocprofv3 kernel summary of bench.py on the data-parallel code path with ONE rank
# (LXO_FORCE_DIST=1: RCCL process group of size 1, bucketed gradient all-reduce hooks, device-side token count), to compare
# with the plain single-GPU profile.  Round 2: +0.34 ms per step at world = 1 (44 tiny fills + 24 tiny copies per step from
# RCCL's single-rank path run beside the latency-bound step kernels, which slow by ~4 %).
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 LXO_FORCE_DIST=1
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pd_prof -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-secondary > $R/gpurun_out/pd_prof.log 2>&1; echo rc=$?
cd $R; DB=$(ls gpurun_out/pd_prof/*/*_results.db | head -1)
python tools/prof_summary.py $DB gpurun_out/pd_kernels.csv pd; python tools/prof_by_grid.py $DB > gpurun_out/pd_bygrid.txt 2>&1; rm -rf gpurun_out/pd_prof
tail -1 gpurun_out/pd_prof.log | grep -o "ms_per_step\": [0-9.]*"
