#!/bin/bash
# same-box A/B of the in-tree library against other builds: tools/ab_lib.sh _bin/a/liblvae_hip.so [...]   (default bench, no cpu baseline)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])"; }
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 2>/dev/null | one in-tree
  for l in "$@"; do python tools/bench_with_lib.py $l --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 2>/dev/null | one $l; done
done | tee gpurun_out/r3/ab_lib.txt
