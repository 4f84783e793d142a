# GPU box: f16x2 kernel tests, micro-benchmark against bf16x3 on the model's shapes, model-level bench lines of both arithmetics
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/h2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_f16x2.py -x -q -s > $O/pytest_f16x2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_f16x2.log
tail -45 $O/pytest_f16x2.log
for p in 2 4; do LVAE_PREC=$p python tools/microbench.py gemm 8 2>&1 | grep -v amdgpu > $O/microbench_prec$p.txt; done
paste -d'|' $O/microbench_prec2.txt $O/microbench_prec4.txt | cut -c1-200
python bench.py --precision f16x2 --no-cpu-baseline --fp32-steps 0 > $O/bench_f16x2.json 2> $O/bench_f16x2.err; tail -3 $O/bench_f16x2.err; cut -c1-900 $O/bench_f16x2.json
python bench.py --precision bf16x3 --no-cpu-baseline --fp32-steps 0 > $O/bench_bf16x3.json 2> $O/bench_bf16x3.err; cut -c1-500 $O/bench_bf16x3.json
