R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_hoist
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
for cfg in "2 1" "2 0" "1 1" "1 0" "2 1" "1 1"; do
  set -- $cfg
  LVAE_ENC_GROUPS=$1 LVAE_SIDE_STREAMS=$2 python bench.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 --qres-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('enc_groups $1 side $2:', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'], 'b1', j['b1']['enc_ms'], j['b1']['dec_ms'])" | tee -a $O/sweep.txt
done
