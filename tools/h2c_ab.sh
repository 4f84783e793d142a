R=$GRAFT_REPO_ROOT
cd /tmp
for rep in 1 2; do
for v in product NOPF NOGELU NOEPI NOMFMA; do
  if [ $v = product ]; then lib=""; else lib=$R/_bin/h2c_$v/liblvae_hip.so; fi
  echo -n "$v: "
  LVAE_LIB=$lib LVAE_MLP_SHAPE=192,384 timeout 200 python $R/tools/microbench.py mlpf 2>&1 | grep "M= 196608\|M=  98304" | sed 's/fc1 + fc2 launches//' | tr '\n' ' '; echo
done; done | tee $R/gpurun_out/h2c/ab.txt
