cd /tmp && export TMPDIR=/tmp
for cf in 512 1024 2048; do for cb in 512 1024; do
  LXO_C1_CAPF=$cf LXO_C1_CAP=$cb rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c1_${cf}_${cb} -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-secondary > /dev/null 2>&1
  DB=$(ls $GRAFT_REPO_ROOT/gpurun_out/c1_${cf}_${cb}/*/*_results.db | head -1)
  echo "capf=$cf capb=$cb"; python $GRAFT_REPO_ROOT/tools/prof_by_grid.py $DB 2>/dev/null | grep conv1 | cut -c1-120
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/c1_${cf}_${cb}
done; done
