#!/bin/bash
# sweep LXO_WG_MINBLK over a few real buckets (tools/real_buckets.py rows)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for mb in 1 4 8 16; do
  echo "== LXO_WG_MINBLK=$mb"
  LXO_WG_MINBLK=$mb RB_SHAPES="50x120,50x240,50x400,100x500" timeout 600 python tools/real_buckets.py wg$mb 8 2>&1 | grep -v amdgpu.ids | cut -c1-130
done
