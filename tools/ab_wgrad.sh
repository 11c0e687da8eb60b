# A/B aid: in-step wgrad timing (bench.py's per-launch HIP events) for LXO_WG_STAGGER values / library builds
for rep in 1 2; do for sg in 10 14 20 30; do LXO_WG_STAGGER=$sg python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
w=d['roofline_wgrad']
print('stagger $sg', d['ms_per_step'], 'wgrad frac', w['frac'], {k:v['us'] for k,v in w['per_launch'].items()}, 'conv', d['roofline']['frac'])
"; done; done
