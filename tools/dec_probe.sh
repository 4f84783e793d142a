R=$GRAFT_REPO_ROOT
cd $R
for cfg in "2 2" "2 4" "2 8" "1 1"; do
  set -- $cfg
  echo "== enc_groups $1 dec_groups $2"
  LVAE_TIMING=1 LVAE_ENC_GROUPS=$1 LVAE_DEC_GROUPS=$2 python bench.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 2>&1 >/tmp/o.json | grep "host phase"
  python -c "import sys,json; j=json.loads(open('/tmp/o.json').read()); print(j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])"
done
