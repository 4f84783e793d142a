# Wave-quantisation probe for gemm_h2p: time per 128-row panel of M when the tile count is / is not a multiple of the CU slots.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/h2pq
mkdir -p $O
cd /tmp
for nke in "384 768 2" "768 384 1" "256 448 2" "448 256 1"; do
  for M in 16384 32768 40960 49152 57344 65536 81920 98304; do
    for cfg in 1 42 41; do
      echo -n "M=$M $nke cfg=$cfg: "
      LVAE_PREC=4 LVAE_H2P=$cfg LVAE_OUT_H2=1 timeout 120 python $R/tools/microbench.py gemm1 $M $nke 2>&1 | grep "us" | tail -1
    done
  done
done | tee $O/quant.txt
