R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_stagger
mkdir -p $O
cd $R
for cfg in "2 0" "2 0.4" "2 0.8" "2 1.2" "2 1.6" "2 2.0" "3 0.5" "3 0.9" "4 0.4" "4 0.7" "2 0"; do
  set -- $cfg
  LVAE_DEC_GROUPS=$1 LVAE_DEC_STAGGER_MS=$2 python tools/dec_trace.py dec 30 8 2>&1 | grep "ms per step" | sed "s/^/groups $1 stagger $2: /" | tee -a $O/sweep.txt
done
