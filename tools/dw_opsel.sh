R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/dw
mkdir -p $O
cd /tmp
timeout 300 $R/tools/ubench/pk_opsel_probe 2>&1 | tee $O/pk_opsel_probe.txt
for v in product dw_PKHI; do
  if [ $v = product ]; then lib=""; else lib=$R/_bin/$v/liblvae_hip.so; fi
  for rep in 1 2 3 4 5; do
    echo -n "$v rep $rep: "; LVAE_LIB=$lib timeout 300 python $R/tools/stress_dw.py both 2>&1 | grep -v amdgpu | tail -2 | tr '\n' ' '; echo
  done
  echo -n "$v dwdw: "; LVAE_LIB=$lib timeout 300 python $R/tools/stress_dw.py dwdw 2>&1 | grep -v amdgpu | tail -1
done | tee $O/stress.txt
for v in product dw_PKHI; do
  if [ $v = product ]; then lib=""; else lib=$R/_bin/$v/liblvae_hip.so; fi
  echo "== $v"; LVAE_LIB=$lib timeout 300 python $R/tools/dw_bench.py 2>&1 | grep -v amdgpu
done | tee $O/dw_bench_ab.txt
