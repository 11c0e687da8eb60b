for rep in 1 2; do for v in "" _wgold; do f=latex_ocr_amd/liblxo$v.so; LXO_LIB_PATH=$GRAFT_REPO_ROOT/$f python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
w=d['roofline_wgrad']
print('$v', d['ms_per_step'], 'wgrad frac', w['frac'], {k:v['us'] for k,v in w['per_launch'].items()}, 'conv', d['roofline']['frac'])
"; done; done
