R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/h2c384
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_f16x2.py -x -q -k "mlp_h2" 2>&1 | tail -8 | tee $O/pytest.txt
cd /tmp
for rep in 1 2; do
LVAE_MLP_SHAPE=384,768 LVAE_MLP_MS=49152,24576,12288,6144 timeout 300 python $R/tools/microbench.py mlpf 2>&1 | grep "M=\|fused"
done | tee $O/mlpf.txt
