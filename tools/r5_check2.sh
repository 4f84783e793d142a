# second half of round 5: full GPU suite, smoke, tile sweep of gemm_h2p under the straight-line epilogue, determinism soak
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_check2
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | grep smoke | tee $O/smoke.txt
for B in 4 8; do for T in 21 23 22 42; do
  echo "== B=$B tile $T" | tee -a $O/tile_sweep.txt
  LVAE_PREC=4 LVAE_H2P=$T LVAE_OUT_H2=1 python tools/microbench.py gemm $B 2>&1 | grep -E "^s|total" | tee -a $O/tile_sweep.txt
done; done
python tools/soak.py --seconds ${SOAK_S:-300} 2>/dev/null | tail -1 > $O/soak_determinism.json; cat $O/soak_determinism.json
