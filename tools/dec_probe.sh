R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_qres.py -x -q 2>&1 | tail -3
for cfg in "2 2" "2 3" "2 4" "3 3" "4 4" "2 8"; do
  set -- $cfg
  echo "== enc_groups $1 dec_groups $2"
  LVAE_TIMING=1 LVAE_ENC_GROUPS=$1 LVAE_DEC_GROUPS=$2 python bench.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 2>&1 >/tmp/o.json | grep "host phase"
  python -c "import sys,json; j=json.loads(open('/tmp/o.json').read()); print(j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])"
done
echo "== python replay, 2 2"; LVAE_PY_REPLAY=1 python bench.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])"
echo "== B=1"; python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --fp32-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])"
echo "== B=1 python replay"; LVAE_PY_REPLAY=1 python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --fp32-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])"
