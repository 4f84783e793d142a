# Round 6 study: gemm_h2p with its LDS ring filled through registers (buffer_load -> VGPR -> ds_write_b128, -DH2P_EXP_STAGED) against the product's
# LDS-DMA form, on the model's MLP shapes (both operands pre-split), batch 4 and 8; bit-equality first.
R=$GRAFT_REPO_ROOT
cd $R
LVAE_LIB=$R/_bin/staged/liblvae_hip.so timeout 600 python -m pytest tests/test_gpu_f16x2.py -m gpu -x -q -k "h2p" 2>&1 | tail -3
for B in 4 8; do
  for rep in 1 2; do
    echo "== product (LDS-DMA) B=$B"; LVAE_PREC=4 LVAE_H2P=1 LVAE_OUT_H2=1 python tools/microbench.py gemm $B 2>/dev/null | grep -E "^s(4|8|16) |total"
    echo "== staged B=$B";            LVAE_LIB=$R/_bin/staged/liblvae_hip.so LVAE_PREC=4 LVAE_H2P=1 LVAE_OUT_H2=1 python tools/microbench.py gemm $B 2>/dev/null | grep -E "^s(4|8|16) |total"
  done
done
