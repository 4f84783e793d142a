#!/bin/bash
# same-box A/B of environment variants of the current tree: tools/ab_env.sh "A=1" "LVAE_X=0" ...   (each run: default bench, no cpu baseline)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3
for i in 1 2; do
  for v in "$@"; do
    env $v python bench.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])"
  done
done | tee gpurun_out/r3/ab_env.txt
