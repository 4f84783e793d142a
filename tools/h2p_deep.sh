R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ub
cd $R
timeout 600 python -m pytest tests/test_gpu_f16x2.py -x -q > $R/gpurun_out/ub/pytest_f16x2.txt 2>&1; tail -3 $R/gpurun_out/ub/pytest_f16x2.txt
cd /tmp
for lib in _bin/h2p_prio0/liblvae_hip.so ""; do
  [ -n "$lib" ] && L=$R/$lib || L=""
  echo "== ring: ${lib:-product (deep ring for few tiles)}"
  for b in 1 4 8; do LVAE_LIB=$L timeout 200 python $R/tools/microbench.py gemmsk_b $b 2>&1 | grep "rows/img"; done
  for shape in "6144 768 384 1" "6144 384 768 2" "1536 768 384 1" "1536 384 768 2" "6144 448 256 1" "6144 256 448 2"; do
    echo -n "$shape: "; LVAE_LIB=$L LVAE_PREC=4 LVAE_H2P=1 LVAE_OUT_H2=1 timeout 120 python $R/tools/microbench.py gemm1 $shape 2>&1 | grep "us" | tail -1
  done
done | tee $R/gpurun_out/ub/h2p_deep.txt
