# Round-4 baseline on a fresh box: GPU tests, default bench line, per-op table (before any change of the round)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_base
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/bench_default.json | cut -c1-400
python $R/tools/op_times.py 8 2>&1 | grep -v amdgpu > $O/op_times_b8.txt
python $R/tools/op_times.py 1 2>&1 | grep -v amdgpu > $O/op_times_b1.txt
