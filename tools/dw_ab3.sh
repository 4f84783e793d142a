R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp8.py -q -m gpu -k "dwconv" -x 2>&1 | tail -2
echo "== product"; python tools/dw_bench.py 2>&1 | grep -v amdgpu
for th in 8 18 28 38 48; do echo "== 10*tpw+TH=$th"; LVAE_DW_CL=$th python tools/dw_bench.py 2>&1 | grep -v amdgpu | sed -n 4,8p; done
