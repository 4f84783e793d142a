# A/B of gemm_h2p's straight-line epilogue (h2p_epilogue_fast) on the MLP GEMM shapes: product | variants under _bin/ (tools/build_exp.sh):
#   epi_generic = gemm_epilogue + select-form quad transpose (the tree before this work), epi_qtsel = product with the select-form transpose,
#   epi_half = product with half-line pre-split stores
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_epi
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_f16x2.py tests/test_gpu_kernels.py tests/test_gpu_fp8.py tests/test_gpu_bf16.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
for rep in 1 2; do for B in 4 8; do for V in product ${VARIANTS:-epi_qtsel epi_generic}; do
  LIB=""; [ $V != product ] && LIB=$R/_bin/$V/liblvae_hip.so
  echo "== B=$B $V rep $rep" | tee -a $O/sweep.txt
  LVAE_LIB=$LIB LVAE_PREC=4 LVAE_H2P=1 LVAE_OUT_H2=1 python tools/microbench.py gemm $B 2>&1 | grep -E "^s|total" | tee -a $O/sweep.txt
done; done; done
grep -E "^==|total" $O/sweep.txt
