# Round 5: where a tile of mlp_h2c<128, 192, 64> (the decode tail's MLP) spends its time: ablation builds (tools/build_exp.sh h2c_<X>
# mlp_h2c.hip -DH2C_EXP_<X>: WRONG RESULTS by construction, they remove work to time what is left) and the in-kernel timeline
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_h2c128
mkdir -p $O
cd /tmp
export LVAE_MLP_SHAPE=128,192 LVAE_MLP_MS=196608,98304
for rep in 1 2; do
echo "== product"; timeout 300 python $R/tools/microbench.py mlpf 2>&1 | grep "M="
for v in NOGELU NOADMA NOWDMA NOMFMA NOEPI NODSR NOBAR; do
echo "== $v"; LVAE_LIB=$R/_bin/h2c_$v/liblvae_hip.so timeout 300 python $R/tools/microbench.py mlpf 2>&1 | grep "M="
done
done > $O/ablations.txt
for M in 196608 98304; do
LVAE_TRACE_M=$M LVAE_LIB=$R/_bin/h2c_TRACE/liblvae_hip.so timeout 300 python $R/tools/microbench.py mlptrace 2>&1 | grep -v amdgpu
done > $O/timeline.txt
cat $O/ablations.txt $O/timeline.txt
