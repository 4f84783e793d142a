R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_h2p23
mkdir -p $O
cd $R
for B in 4 8; do for T in 21 23 22; do
  echo "== B=$B tile $T" | tee -a $O/sweep.txt
  LVAE_PREC=4 LVAE_H2P=$T LVAE_OUT_H2=1 python tools/microbench.py gemm $B 2>&1 | grep -v amdgpu | grep -E "^s|total" | tee -a $O/sweep.txt
done; done
