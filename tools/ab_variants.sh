#!/bin/bash
# A/B of library variants on one box, alternating processes: bash tools/ab_variants.sh <tool.py> <reps> <variant suffixes...>   ("" = the shipped library)
TOOL=$1; REPS=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    if [ "$v" = "base" ]; then lib=$R/latex_ocr_amd/liblxo.so; else lib=$R/latex_ocr_amd/liblxo_$v.so; fi
    [ -f $lib ] || { echo "no $lib"; continue; }
    echo -n "[$v] "; LXO_LIB_PATH=$lib timeout 300 python tools/$TOOL 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-1}
  done
done
