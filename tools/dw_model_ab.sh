R=$GRAFT_REPO_ROOT; cd $R
echo "== kernel tests"; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp8.py -q -m gpu -k "dwconv" -x 2>&1 | tail -3
echo "== model tests"; timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_qres.py tests/test_gpu_bf16.py -q -m gpu -x 2>&1 | tail -25
echo "== bench"; python tools/dw_bench.py 2>&1 | grep -v amdgpu
