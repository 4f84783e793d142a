# depthwise+LN kernel: unit tests (incl. beside-GEMMs), qarv / qres model parity, timings
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp8.py -q -m gpu -k "dwconv" -x 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_qres.py tests/test_gpu_bf16.py -q -m gpu -x 2>&1 | tail -4
python tools/dw_bench.py 2>&1 | grep -v amdgpu
