R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ub
cd /tmp
for shape in "49152 768 384 1" "49152 384 768 2" "12288 768 384 1" "24576 768 384 1" "24576 448 256 1" "12288 1024 512 1"; do
  for v in old new new_prio3 old new; do
    case $v in old) lib=$R/_bin/h2p_prio0/liblvae_hip.so;; new) lib="";; new_prio3) lib=$R/_bin/h2p_ord_prio3/liblvae_hip.so;; esac
    echo -n "$shape $v: "
    LVAE_LIB=$lib LVAE_PREC=4 LVAE_H2P=1 LVAE_OUT_H2=1 timeout 120 python $R/tools/microbench.py gemm1 $shape 2>&1 | grep "us" | tail -1
  done
done | tee $R/gpurun_out/ub/h2p_order.txt
