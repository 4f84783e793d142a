# Round 5: mlp_h2c<128, 192, 64> with the tile's A rows resident in LDS (ARES) against the ring form (-DH2C_EXP_NOARES build), alternating on one box;
# bit-identity tests of the fused kernels; in-kernel timeline of the resident form
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_h2c_ares
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_f16x2.py -q -x -k "mlp_h2" 2>&1 | tail -3 | tee $O/tests.txt
cd /tmp
export LVAE_MLP_SHAPE=128,192 LVAE_MLP_MS=196608,98304,49152,24576,6144,640
for rep in 1 2 3; do
echo "== resident A rows (product)"; timeout 300 python $R/tools/microbench.py mlpf 2>&1 | grep "M=\|fused"
echo "== ring form (NOARES)"; LVAE_LIB=$R/_bin/h2c_NOARES/liblvae_hip.so timeout 300 python $R/tools/microbench.py mlpf 2>&1 | grep "M=\|fused"
done > $O/ab.txt
cat $O/ab.txt
for M in 196608 98304; do
LVAE_TRACE_M=$M LVAE_LIB=$R/_bin/h2c_TRACE/liblvae_hip.so timeout 300 python $R/tools/microbench.py mlptrace 2>&1 | grep -v amdgpu
done > $O/timeline.txt
cat $O/timeline.txt
