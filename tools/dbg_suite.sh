R=$GRAFT_REPO_ROOT; cd $R
echo "== alone x2"; for i in 1 2; do timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "beside_gemms" 2>&1 | tail -1; done
echo "== whole file"; timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu 2>&1 | tail -3
echo "== split_k + beside"; timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "split_k or beside" 2>&1 | tail -3
echo "== dwconv + beside"; timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "dwconv" 2>&1 | tail -3
echo "== gemm (no split_k) + beside"; timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "(gemm and not split_k) or beside" 2>&1 | tail -3
