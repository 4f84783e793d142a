# Round 6: serial split-K GEMM (gemm_h2p FOLD) with eight loader waves (LVAE_FOLD_LOADERS=1) against the compute waves issuing their own LDS-DMA (=0):
# bit-equality tests, then the small-map MLP layers at batch 1, 4, 8 (tools/microbench.py gemmsk_b)
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_f16x2.py -m gpu -x -q -k "split_k or mlp_sk" 2>&1 | tail -2
for B in 1 4 8; do for v in 0 1; do echo "== LVAE_FOLD_LOADERS=$v B=$B"; LVAE_FOLD_LOADERS=$v python tools/microbench.py gemmsk_b $B 2>/dev/null | grep serial; done; done
