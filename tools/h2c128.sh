R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/h2c128
mkdir -p $O
cd /tmp
for rep in 1 2; do
echo "== product (mlp_h2f_kernel, weights in LDS)"; timeout 300 python $R/tools/microbench.py mlpf 2>&1 | grep "M=\|fused"
echo "== mlp_h2c<128,192,64>"; LVAE_LIB=$R/_bin/h2c_128/liblvae_hip.so LVAE_EXP_H2C_128=1 timeout 300 python $R/tools/microbench.py mlpf 2>&1 | grep "M=\|fused"
done | tee $O/mlpf.txt
