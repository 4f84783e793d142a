# final check of the round: GPU tests, smoke, the default bench line (what the driver runs)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_final
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | grep smoke | tee $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json, os
j = json.load(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r5_final/bench_default.json'))
print({k: j[k] for k in ('value', 'ms_per_step', 'enc_ms_per_step', 'dec_ms_per_step', 'fp32_mfma_mode_value', 'bf16x3_mode_value', 'qres34m_value', 'config5_value')})
print(j['b1']); print(j['config5']['speedup_vs_fp32_class'], j['config5']['enc_ms_per_step'], j['config5']['dec_ms_per_step'])
print({k: j['roofline'][k] for k in ('achieved', 'frac', 'launches_per_step', 'timed_plans_launches_per_step', 'avg_launch_us', 'traffic')})
PY
