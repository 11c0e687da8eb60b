# A/B aid: bench.py headline + in-step conv / wgrad rooflines for the shipped library and latex_ocr_amd/liblxo_prev.so (if present)
for rep in 1 2; do for v in "" _prev; do f=latex_ocr_amd/liblxo$v.so; [ -f $f ] || continue; LXO_LIB_PATH=$GRAFT_REPO_ROOT/$f python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
w=d['roofline_wgrad']; c=d['roofline']
print('lib$v', d['ms_per_step'], 'conv', c['frac'], {k.split(':')[0][5:]+k.split(':')[1][4:]:v['us'] for k,v in c['per_launch'].items()}, 'wgrad', w['frac'])
"; done; done
