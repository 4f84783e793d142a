# persistent loader-wave GEMM (tools/studies/gemm_h2q.hip, cfg 61) vs gemm_h2p 128 x 64 (cfg 21) and the product choice (cfg 1)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/h2q
mkdir -p $O
# build first (container): EXTRA_SRC=gemm_h2q.hip tools/build_exp.sh h2q gemm_h2p.hip -DLVAE_EXP_H2Q
cd /tmp
export LVAE_LIB=$R/_bin/h2q/liblvae_hip.so
timeout 600 python $R/tools/h2q_equal.py 2>&1 | tail -12 | tee $O/equal.txt
for shape in "49152 768 384 1" "49152 384 768 2" "49152 448 256 1" "49152 256 448 2" "49152 512 256 1" "24576 448 256 1" "24576 768 384 1" "12288 768 384 1" "196608 192 128 1" "49152 768 384 0"; do
  for v in 21 61 1 21 61 1; do
    echo -n "$shape cfg=$v: "
    LVAE_PREC=4 LVAE_H2P=$v LVAE_OUT_H2=1 timeout 120 python $R/tools/microbench.py gemm1 $shape 2>&1 | grep "us" | tail -1
  done
done | tee $O/bench.txt
