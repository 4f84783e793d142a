# pipeline-group sweep of the default bench line on one box: tools/groups_sweep.sh [precision]
R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/r3
P=${1:-f16x2}
for cfg in "2 2" "2 3" "2 4" "3 2" "3 3" "2 2" "2 3"; do
  set -- $cfg
  LVAE_ENC_GROUPS=$1 LVAE_DEC_GROUPS=$2 python bench.py --precision $P --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('enc_groups $1 dec_groups $2:', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])"
done | tee gpurun_out/r3/groups_sweep_$P.txt
