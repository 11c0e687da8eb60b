#!/bin/bash
# SQ counter pass: where do the conv kernels' cycles go?  usage: bash tools/gpu_pmc_sq.sh <tag> <workload.py> <kernel substring>
TAG=${1:-r02}; WL=${2:-tools/pmc_conv.py}; PAT=${3:-conv_halo2wg}
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/gpurun_out/${TAG}_pmc_sq -- python $R/$WL > $R/gpurun_out/${TAG}_pmc_sq.log 2>&1; echo "pmc rc=$?"
cd $R
DB=$(ls gpurun_out/${TAG}_pmc_sq/*/*_results.db | head -1)
python tools/pmc_dump.py $DB $PAT > gpurun_out/${TAG}_sq_${PAT}.txt 2>&1
rm -rf gpurun_out/${TAG}_pmc_sq
cat gpurun_out/${TAG}_sq_${PAT}.txt
