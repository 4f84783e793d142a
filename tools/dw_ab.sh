# depthwise+LN: correctness tests, then timings of the earlier forms (LVAE_DW_CL=0) vs the channel-per-lane kernel
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp8.py -q -m gpu -k "dwconv" -x 2>&1 | tail -8
echo "== earlier forms"; LVAE_DW_CL=0 python tools/dw_bench.py 2>&1 | grep -v amdgpu
echo "== channel-per-lane"; python tools/dw_bench.py 2>&1 | grep -v amdgpu
for rs in 1 2 4 8; do echo "== channel-per-lane TH=$rs"; LVAE_DW_CL=$rs python tools/dw_bench.py 2>&1 | grep -v amdgpu ; done
