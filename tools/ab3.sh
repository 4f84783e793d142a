R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
  (cd $R/_old && python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('OLD', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])")
  (cd $R && python bench.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('NEW', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])")
done
