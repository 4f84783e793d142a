R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ub
cd /tmp
for shape in "49152 768 384 1" "12288 768 384 1" "49152 448 256 1" "49152 384 768 2" "12288 1024 512 1" "24576 768 384 1"; do
  for v in "h2p_prio0 0" "h2p_prio3 0" "h2p_prio0 1" "h2p_prio3 1" "h2p_prio3 2"; do
    set -- $v
    echo -n "$shape lib=$1 stagger=$2: "
    LVAE_LIB=$R/_bin/$1/liblvae_hip.so LVAE_H2P_STAGGER=$2 LVAE_PREC=4 LVAE_H2P=1 LVAE_OUT_H2=1 timeout 120 python $R/tools/microbench.py gemm1 $shape 2>&1 | grep "us" | tail -1
  done
done | tee $R/gpurun_out/ub/h2p_prio.txt
