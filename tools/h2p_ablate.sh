R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ub
cd /tmp
for shape in "49152 768 384 1" "49152 384 768 2" "12288 768 384 1" "24576 768 384 1" "24576 448 256 1"; do
  for v in product NOMFMA NODSR NOEPI NODMA; do
    if [ $v = product ]; then lib=""; else lib=$R/_bin/h2p_$v/liblvae_hip.so; fi
    echo -n "$shape $v: "
    LVAE_LIB=$lib LVAE_PREC=4 LVAE_H2P=1 LVAE_OUT_H2=1 timeout 120 python $R/tools/microbench.py gemm1 $shape 2>&1 | grep "us" | tail -1
  done
done | tee $R/gpurun_out/ub/h2p_ablate.txt
