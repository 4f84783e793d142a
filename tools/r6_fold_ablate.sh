# Round 6: what a serial split-K (FOLD) launch of gemm_h2p costs without its MFMAs / fragment reads / DMA traffic / epilogue (timing ablations,
# wrong results by construction), with and without the loader waves:   bash tools/r6_fold_ablate.sh  (GPU box; builds made by the caller)
R=$GRAFT_REPO_ROOT
cd $R
for fl in 1 0; do
  for v in product NOMFMA NODSR NODMA NOEPI; do
    L=_bin/h2p_$v/liblvae_hip.so; [ $v = product ] && L=_bin/pipe/liblvae_hip.so
    echo "-- loaders=$fl $v"
    LVAE_LIB=$L LVAE_FOLD_LOADERS=$fl python tools/microbench.py gemmsk_b ${1:-4} 2>&1 | grep -v amdgpu | awk '{print $1, $3, $4, $5, $6, $7, $8, "serial", $(NF-1)}'
  done
done
