# GPU box check: the driver's own sequence (pytest -m gpu, smoke, default bench) with outputs under gpurun_out/check/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/check
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -3 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
cat $O/bench_default.json
