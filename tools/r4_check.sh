# round-4 check: selected tests first, then (R4_FULL=1) the whole GPU suite, then (R4_BENCH=1) the default bench line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_check
mkdir -p $O
cd $R
timeout 900 python -m pytest ${R4_TESTS:-tests/test_gpu_overflow.py} -x -q > $O/pytest_new.txt 2>&1; echo "rc=$?" >> $O/pytest_new.txt
tail -25 $O/pytest_new.txt
if [ "${R4_FULL:-1}" = "1" ]; then
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -6 $O/pytest_gpu.txt
fi
cd /tmp && export TMPDIR=/tmp
if [ "${R4_BENCH:-1}" = "1" ]; then
python $R/bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
cut -c1-330 $O/bench_default.json
fi
if [ -n "$R4_EXTRA" ]; then bash $R/$R4_EXTRA; fi
