R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_kernels.py -q -m gpu -k "not test_gpu_kernels or beside_gemms" -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|AssertionError|shape" | cut -c1-1500 | tail -8
