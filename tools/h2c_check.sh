R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/h2c
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_f16x2.py -x -q -k "mlp_h2" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
cd /tmp
LVAE_MLP_SHAPE=192,384 timeout 300 python $R/tools/microbench.py mlpf 2>&1 | grep -v amdgpu | tee $O/mlpf_192_384.txt
[ -n "$H2C_EXTRA" ] && bash $R/$H2C_EXTRA
