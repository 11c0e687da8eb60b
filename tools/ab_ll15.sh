# same-box A/B of the forward chain with polled chunk partials (gpurun_diag_ll15.so: -DLXO_XDEC_LLMASK=15) against the shipped library
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
echo "== correctness of the variant (chain == launch-per-step == oracle)"
LXO_LIB_PATH=$R/gpurun_diag_ll15.so timeout 900 python -m pytest tests/test_gpu_xdec.py tests/test_gpu_benchcfg.py tests/test_gpu_realbatch.py tests/test_gpu_determinism.py -m gpu -q -x 2>&1 | tail -4
echo "== timing"
for rep in 1 2 3; do
  LXO_XDEC_LL=shipped python tools/xdec_ab.py 2>&1 | grep "^LL"
  LXO_XDEC_LL=ll15 LXO_LIB_PATH=$R/gpurun_diag_ll15.so python tools/xdec_ab.py 2>&1 | grep "^LL"
done
LXO_LIB_PATH=$R/gpurun_diag_ll15.so python tools/xdec_stamps.py 2>&1 | tail -16
