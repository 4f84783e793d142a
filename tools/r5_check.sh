# Round-5 check: GPU tests (fail fast), then the bench line (short extras) and the host timing breakdown
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_check
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --fp32-steps 0 --config5-steps 0 --qres-steps 0 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json, os
j = json.load(open(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/r5_check/bench.json'))
print({k: j[k] for k in ('value', 'ms_per_step', 'enc_ms_per_step', 'dec_ms_per_step')}, j['b1'], j['host_coder'])
PY
python tools/enc_tail.py 8 2>&1 | grep -v amdgpu | tail -1 | tee $O/enc_tail_b8.txt
python tools/enc_tail.py 1 2>&1 | grep -v amdgpu | tail -1 | tee $O/enc_tail_b1.txt
