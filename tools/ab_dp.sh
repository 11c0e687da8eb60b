# A/B aid: the data-parallel plumbing at world size 1 (nccl) against the plain step
for rep in 1 2; do
for cfg in "dp_host" "dp_stream" "plain"; do
  case $cfg in plain) E="";; dp_host) E="LXO_FORCE_DIST=1 LXO_DP_HOST_ORDERED=1";; dp_stream) E="LXO_FORCE_DIST=1 LXO_DP_HOST_ORDERED=0";; esac
  env $E RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1])
print('$cfg', d['ms_per_step'], d.get('data_parallel'))
"; done; done
