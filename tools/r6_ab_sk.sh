# Round 6: the fused small-map MLP (csrc/mlp_sk.hip) on / off in the bench (same box, alternating): B=8 headline, B=1, calibrated rows
R=$GRAFT_REPO_ROOT
BX="--no-cpu-baseline --fp32-steps 0 --config5-steps 0 --qres-steps 0 --size-steps 0 --no-kernel-timing --steps 20"
for rep in 1 2 3; do
for v in 0 1024; do
  LVAE_MLP_SK_MAX_ROWS=$v python $R/bench.py $BX 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1])
cw=j['coder_workloads']
print('max_rows=$v', 'value', j['value'], 'enc', j['enc_ms_per_step'], 'dec', j['dec_ms_per_step'], '| b1 enc', j['b1']['enc_ms'], 'dec', j['b1']['dec_ms'], '| cal b8 dec', cw['b8_512x768']['calibrated']['dec_ms_per_step'], 'b1 dec', cw['b1_512x768']['calibrated']['dec_ms_per_step'])
"
done
done
