set -o pipefail
cd $GRAFT_REPO_ROOT
P=$GRAFT_REPO_ROOT/small-object-detection-transformers_amd
for r in 1 2; do
  timeout -k 10 300 python tools/ab_gemm_lib.py base > gpurun_out/r06_slp_base$r.log 2>&1 || { tail -n 20 gpurun_out/r06_slp_base$r.log; exit 1; }
  SODT_LIB_PATH=$P/libsodt_hip_ns3.so timeout -k 10 300 python tools/ab_gemm_lib.py noslp > gpurun_out/r06_slp_ns$r.log 2>&1 || { tail -n 20 gpurun_out/r06_slp_ns$r.log; exit 1; }
done
paste <(grep -v "^\[base\] lib" gpurun_out/r06_slp_base1.log | cut -c1-80) <(awk '{print $(NF-3), $(NF-2)}' gpurun_out/r06_slp_ns1.log) <(awk '{print $(NF-3), $(NF-2)}' gpurun_out/r06_slp_base2.log) <(awk '{print $(NF-3), $(NF-2)}' gpurun_out/r06_slp_ns2.log)
echo "== mlp"
timeout -k 10 300 python tools/mb_mlp.py 8 --rounds 3 2>&1 | tail -n 6
SODT_LIB_PATH=$P/libsodt_hip_nsm.so timeout -k 10 300 python tools/mb_mlp.py 8 --rounds 3 2>&1 | tail -n 6
