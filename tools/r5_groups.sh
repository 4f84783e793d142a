R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_groups
mkdir -p $O
cd $R
for cfg in "2 2" "1 2" "3 2" "2 2" "1 2"; do
  set -- $cfg
  LVAE_ENC_GROUPS=$1 LVAE_DEC_GROUPS=$2 python bench.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 --b1-steps 0 --qres-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('enc_groups $1 dec_groups $2:', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])" | tee -a $O/sweep.txt
done
LVAE_GROUPS=1 python tools/op_times.py 8 2>&1 | grep -v amdgpu | head -3 | tee -a $O/sweep.txt
