#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for U in 4 8; do for WGS in 256 512 1024; do
  echo -n "U=$U WGS=$WGS: "
  LXO_ATT_U=$U LXO_ATT_WGS=$WGS python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], 'attfwd_us', d['roofline_attention']['avg_launch_us'])"
done; done
