#!/bin/bash
# counter passes over the two decoder chains inside real training steps: SQ activity (one pass) and L2 hits / misses (one pass)
TAG=${1:-r4c}
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
(timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace -d $R/gpurun_out/${TAG}_pmc_sq -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-secondary --no-pmc > $R/gpurun_out/${TAG}_pmc_sq.log 2>&1; echo "pmc sq rc=$?")
(timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $R/gpurun_out/${TAG}_pmc_l2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-secondary --no-pmc > $R/gpurun_out/${TAG}_pmc_l2.log 2>&1; echo "pmc l2 rc=$?")
cd $R
DB=$(ls gpurun_out/${TAG}_pmc_sq/*/*_results.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then for k in xdec_fwd_kernel xdec_bwd_kernel datt_img_kernel; do python tools/pmc_dump.py $DB $k > gpurun_out/${TAG}_sq_$k.txt 2>&1; cat gpurun_out/${TAG}_sq_$k.txt; done; rm -rf gpurun_out/${TAG}_pmc_sq; else tail -5 gpurun_out/${TAG}_pmc_sq.log; fi
DB=$(ls gpurun_out/${TAG}_pmc_l2/*/*_results.db 2>/dev/null | head -1)
if [ -n "$DB" ]; then python tools/pmc_l2.py $DB gpurun_out/${TAG}_l2.json xdec_fwd_kernel xdec_bwd_kernel conv_halo2wg_kernel conv_wgrad_kernel datt_img_kernel gemm_tn_tr_kernel; rm -rf gpurun_out/${TAG}_pmc_l2; else tail -5 gpurun_out/${TAG}_pmc_l2.log; fi
