set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_wmsa_block_gpu.py -x -q > gpurun_out/r06_hg_tests.log 2>&1 || { tail -n 30 gpurun_out/r06_hg_tests.log; exit 1; }
tail -n 3 gpurun_out/r06_hg_tests.log
timeout -k 10 300 python tools/mb_wmsa.py 8 --hg-stamps > gpurun_out/r06_mb_deph.log 2>&1 || { tail -n 20 gpurun_out/r06_mb_deph.log; exit 1; }
cat gpurun_out/r06_mb_deph.log
for v in r5 noslp r5; do
  echo "== $v"
  SODT_LIB_PATH=$GRAFT_REPO_ROOT/small-object-detection-transformers_amd/libsodt_hip_$v.so timeout -k 10 300 python tools/mb_wmsa.py 8 > gpurun_out/r06_mb_$v.log 2>&1 || { tail -n 20 gpurun_out/r06_mb_$v.log; exit 1; }
  grep fused gpurun_out/r06_mb_$v.log
done
echo "== default again"
timeout -k 10 300 python tools/mb_wmsa.py 8 | grep fused
