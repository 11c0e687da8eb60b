#!/bin/bash
# round-3 GPU-box driver: bash tools/gpu_r3.sh <tag> <what...>
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for what in "$@"; do
case $what in
  refgold)  timeout 600 python -m pytest tests/test_gpu_refgold.py -m gpu -q -s > gpurun_out/${TAG}_refgold.log 2>&1; echo "refgold rc=$?"; grep -E "passed|failed|cosine|bf16|Error|assert" gpurun_out/${TAG}_refgold.log | tail -30;;
  trained)  timeout 1500 python -m pytest tests/test_gpu_trained.py tests/test_gpu_benchcfg.py tests/test_gpu_refgold.py -m gpu -q -s --durations=12 > gpurun_out/${TAG}_trained.log 2>&1; echo "trained rc=$?"; grep -E "passed|failed|loss|agreement|beam 5|margin|Error|assert|cosine|s call" gpurun_out/${TAG}_trained.log | tail -60;;
  cstamps)  timeout 300 python tools/conv_stamps.py > gpurun_out/${TAG}_conv_stamps.log 2>&1; echo "cstamps rc=$?";;
  cdiag)    for d in 1 2 4 7; do LXO_CONV_DIAG=$d timeout 300 python tools/conv_stamps.py 2>&1 | grep -v amdgpu | head -17 > gpurun_out/${TAG}_conv_diag$d.log; echo "diag $d rc=$?"; done;;
  ctimeline) timeout 300 python tools/conv_timeline.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_conv_timeline.log; echo "timeline rc=$?";;
  rstamps)  timeout 300 python tools/rstep_stamps.py > gpurun_out/${TAG}_rstep_stamps.log 2>&1; echo "rstamps rc=$?";;
  wstamps)  timeout 300 python tools/wgrad_stamps.py > gpurun_out/${TAG}_wgrad_stamps.log 2>&1; echo "wstamps rc=$?";;
  wtime)    for v in "" _wgold _wg2 _wg5 _wg7; do f=latex_ocr_amd/liblxo$v.so; [ -f $f ] || continue; echo "== $f"; LXO_LIB_PATH=$R/$f timeout 300 python tools/wgrad_time.py 2>&1 | grep -v amdgpu; done > gpurun_out/${TAG}_wgrad_time.log 2>&1; echo "wtime rc=$?"; cat gpurun_out/${TAG}_wgrad_time.log;;
  *) bash tools/gpu_call.sh $TAG $what;;
esac
done
