# Epilogue-interleaved persistent fc1 GEMM study (tools/studies/gemm_h2e.hip): bit-equality, then cfg 1 (product choice) against cfg 51, then the
# study's ablations.  Build first (container):
#   EXTRA_SRC=gemm_h2e.hip tools/build_exp.sh h2e gemm_h2p.hip -DLVAE_EXP_H2E
#   for v in NOSLICE NOSTORE NOMFMA; do EXTRA_SRC=gemm_h2e.hip tools/build_exp.sh h2e_$v gemm_h2p.hip -DLVAE_EXP_H2E -DH2E_EXP_$v; done
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/h2e
mkdir -p $O
cd /tmp
LVAE_LIB=$R/_bin/h2e/liblvae_hip.so timeout 600 python $R/tools/h2e_equal.py 2>&1 | tail -8 | tee $O/equal.txt
for shape in "49152 768 384 1" "24576 768 384 1" "12288 1024 512 1" "49152 448 256 1" "6144 768 384 1"; do
  for v in 1 51; do
    echo -n "$shape cfg=$v: "
    LVAE_LIB=$R/_bin/h2e/liblvae_hip.so LVAE_PREC=4 LVAE_H2P=$v LVAE_OUT_H2=1 timeout 120 python $R/tools/microbench.py gemm1 $shape 2>&1 | grep "us" | tail -1
  done
done | tee $O/h2e_bench.txt
