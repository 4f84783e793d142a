# GPU box: full -m gpu suite + default bench line + per-op table with the round-3 defaults
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
timeout 2700 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; echo "pytest rc=$?" >> $O/pytest_full.log
tail -75 $O/pytest_full.log
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err; cut -c1-330 $O/bench_default.json
python tools/op_times.py 8 2>&1 | grep -v amdgpu > $O/op_times_b8.txt; head -3 $O/op_times_b8.txt
