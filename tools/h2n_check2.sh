R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/h2n
mkdir -p $O
cd /tmp
{
for sh in "128,192 196608 48 432" "64,96 49152 96 864" "64,96 12288 96 864" "128,192 24576 48 432"; do
  set -- $sh
  echo -n "conv3 $2 $3 $4 bf16x3 (what the model runs today): "; LVAE_PREC=2 LVAE_CONV3=$1 timeout 120 python $R/tools/microbench.py gemm1 $2 $3 $4 1 2>&1 | grep "us" | tail -1
  echo -n "conv3 $2 $3 $4 f16x2 gemm_h2n: "; LVAE_PREC=4 LVAE_CFG=3 LVAE_CONV3=$1 timeout 120 python $R/tools/microbench.py gemm1 $2 $3 $4 1 2>&1 | grep "us" | tail -1
done
} | tee $O/bench2.txt
