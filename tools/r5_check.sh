# Round-5 check: GPU tests (fail fast), then the bench line (short extras), host timing of encode, decode timeline
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_check
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --fp32-steps 0 --config5-steps 0 --qres-steps 0 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json, os
j = json.load(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r5_check/bench.json'))
print({k: j[k] for k in ('value', 'ms_per_step', 'enc_ms_per_step', 'dec_ms_per_step')}, j['b1'])
r = j['roofline']; print({k: r[k] for k in ('achieved', 'frac', 'launches', 'launches_per_step', 'timed_plans_launches_per_step', 'avg_launch_us', 'measured_over')})
PY
python tools/enc_tail.py 8 2>&1 | grep -v amdgpu | tail -1 | tee $O/enc_tail_b8.txt
python tools/enc_tail.py 1 2>&1 | grep -v amdgpu | tail -1 | tee $O/enc_tail_b1.txt
python tools/dec_timeline.py 8 20 2>&1 | grep -v amdgpu | tee $O/dec_timeline_b8.txt | head -3
python tools/dec_timeline.py 1 20 2>&1 | grep -v amdgpu | tee $O/dec_timeline_b1.txt | head -3
