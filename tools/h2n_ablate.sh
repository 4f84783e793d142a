R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/h2n
mkdir -p $O
cd /tmp
{
for v in product CENTER NOMFMA NOLOAD; do
  if [ $v = product ]; then lib=""; else lib=$R/_bin/h2n_$v/liblvae_hip.so; fi
  echo -n "conv3 196608x48x432 $v: "; LVAE_LIB=$lib LVAE_PREC=4 LVAE_CFG=3 LVAE_CONV3=128,192 timeout 120 python $R/tools/microbench.py gemm1 196608 48 432 1 2>&1 | grep "us" | tail -1
  echo -n "conv3 49152x96x864 $v: "; LVAE_LIB=$lib LVAE_PREC=4 LVAE_CFG=3 LVAE_CONV3=64,96 timeout 120 python $R/tools/microbench.py gemm1 49152 96 864 1 2>&1 | grep "us" | tail -1
  echo -n "plain 196608x48x384 $v: "; LVAE_LIB=$lib LVAE_PREC=4 LVAE_CFG=3 timeout 120 python $R/tools/microbench.py gemm1 196608 48 384 1 2>&1 | grep "us" | tail -1
done
echo -n "plain 196608x48x432 (340 MB of A from HBM): "; LVAE_PREC=4 LVAE_CFG=3 timeout 120 python $R/tools/microbench.py gemm1 196608 48 432 1 2>&1 | grep "us" | tail -1
} | tee $O/ablate.txt
