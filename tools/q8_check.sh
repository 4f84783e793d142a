R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/q8
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_fp8.py -x -q -s 2>&1 | tail -25
python bench.py --precision fp8 --no-cpu-baseline > $O/bench_fp8_b8.json 2>/dev/null; cut -c1-300 $O/bench_fp8_b8.json
python bench.py --precision fp8 --no-cpu-baseline --batch 4 --height 1216 --width 1216 --steps 8 > $O/bench_fp8_b4_1216.json 2>/dev/null; cut -c1-300 $O/bench_fp8_b4_1216.json
python bench.py --no-cpu-baseline --no-kernel-timing --batch 4 --height 1216 --width 1216 --steps 8 --fp32-steps 0 2>/dev/null | cut -c1-200
OP_TIMES_PRECISION=fp8 python tools/op_times.py 8 2>&1 | grep -v amdgpu > $O/op_times_fp8.txt; head -30 $O/op_times_fp8.txt
