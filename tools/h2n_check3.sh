R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/h2n
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_f16x2.py -x -q -k "h2n" 2>&1 | tail -8 | tee $O/pytest.txt
bash tools/h2n_check2.sh
cd /tmp
b() { for cfg in 1 3; do echo -n "$1 cfg=$cfg: "; env LVAE_PREC=4 LVAE_CFG=$cfg $2 $3 timeout 120 python $R/tools/microbench.py gemm1 $4 $5 $6 $7 2>&1 | grep "us" | tail -1; done; }
{
b "qres s8 3x3 96->96" LVAE_CONV3=64,96 X=1 49152 96 864 1
b "qres s4 1x1 384->48" X=1 X=1 196608 48 384 1
b "qres s4 1x1 192->48" X=1 X=1 196608 48 192 1
b "qres s8 1x1 768->96" X=1 X=1 49152 96 768 1
b "qres s8 1x1 384->96" X=1 X=1 49152 96 384 1
b "qarv s8 head 3x3 256->8" LVAE_CONV3=64,96 X=1 49152 8 2304 0
b "qarv s8 head 3x3 256->8 S=3" LVAE_CONV3=64,96 LVAE_KSPLIT=3 49152 8 2304 0
b "qarv s16 head 3x3 384->96 S=6" LVAE_CONV3=32,48 LVAE_KSPLIT=6 12288 96 3456 0
} | tee $O/bench3.txt
