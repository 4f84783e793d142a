R=$GRAFT_REPO_ROOT
cd $R
for cfg in "2 2" "2 4" "4 4" "1 2" "2 3" "3 3"; do
  set -- $cfg
  LVAE_ENC_GROUPS=$1 LVAE_DEC_GROUPS=$2 python bench.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('enc_groups $1 dec_groups $2:', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])"
done
