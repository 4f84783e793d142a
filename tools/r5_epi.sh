# A/B of gemm_h2p's straight-line epilogue (h2p_epilogue_fast): product (full-line pre-split stores) | half-line stores | gemm_epilogue
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_epi
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_f16x2.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
for rep in 1 2; do for B in 4 8; do for V in product epi_half epi_generic; do
  LIB=""; [ $V != product ] && LIB=$R/_bin/$V/liblvae_hip.so
  echo "== B=$B $V rep $rep" | tee -a $O/sweep.txt
  LVAE_LIB=$LIB LVAE_PREC=4 LVAE_H2P=1 LVAE_OUT_H2=1 python tools/microbench.py gemm $B 2>&1 | grep -E "^s|total" | tee -a $O/sweep.txt
done; done; done
