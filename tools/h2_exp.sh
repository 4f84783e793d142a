R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/h2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_f16x2.py -x -q > $O/pytest_f16x2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_f16x2.log
tail -25 $O/pytest_f16x2.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "golden or batch_equals or round_trip" > $O/pytest_model.log 2>&1; echo "pytest rc=$?" >> $O/pytest_model.log
tail -40 $O/pytest_model.log
python bench.py --precision f16x2 --no-cpu-baseline --fp32-steps 0 > $O/bench_f16x2_h2p.json 2> $O/bench_f16x2_h2p.err; tail -3 $O/bench_f16x2_h2p.err; cut -c1-400 $O/bench_f16x2_h2p.json
OP_TIMES_PRECISION=f16x2 python tools/op_times.py 8 2>&1 | grep -v amdgpu > $O/op_times_b8_f16x2_h2p.txt
head -48 $O/op_times_b8_f16x2_h2p.txt
