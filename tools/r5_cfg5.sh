R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_cfg5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
OP_TIMES_PRECISION=fp8 python $R/tools/op_times.py 4 1216 1216 2>&1 | grep -v amdgpu > $O/op_times_fp8_b4_1216.txt
python $R/tools/op_times.py 4 1216 1216 2>&1 | grep -v amdgpu > $O/op_times_f16x2_b4_1216.txt
python $R/bench.py --precision fp8 --no-cpu-baseline --batch 4 --height 1216 --width 1216 --steps 8 > $O/bench_fp8.json 2>$O/bench_fp8.err
head -c 1500 $O/bench_fp8.json
head -30 $O/op_times_fp8_b4_1216.txt
