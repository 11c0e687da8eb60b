#!/bin/bash
# SQ / LDS counter pass over the conv launches.  usage: bash tools/gpu_pmc_lds.sh <tag> [kernel substring]
TAG=${1:-r03}; PAT=${2:-conv_halo2wg}
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gpurun_out/${TAG}_pmc_lds -- python $R/tools/pmc_conv.py > $R/gpurun_out/${TAG}_pmc_lds.log 2>&1; echo "pmc rc=$?"
cd $R
DB=$(ls gpurun_out/${TAG}_pmc_lds/*/*_results.db | head -1)
python tools/pmc_dump.py $DB $PAT > gpurun_out/${TAG}_lds_${PAT}.txt 2>&1
rm -rf gpurun_out/${TAG}_pmc_lds
head -50 gpurun_out/${TAG}_lds_${PAT}.txt
