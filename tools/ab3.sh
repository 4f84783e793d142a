#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3
one() { python bench.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])"; }
for i in 1 2 3; do
  (cd $R/_old && python bench.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('OLDTREE', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])")
  (cd $R && one NEW)
  (cd $R && LVAE_SERIAL=0 LVAE_H2P_RULE=1 one NEW_bothoff)
done | tee $R/gpurun_out/r3/ab3.txt
