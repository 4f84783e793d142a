#!/bin/bash
# every MLP shape of the model at batch $1 (default 4) under each gemm_h2p tile ($TILES, default "21 22 41 42"): which tile wins where
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3
B=${1:-4}
for t in ${TILES:-21 22 41 42}; do echo "== tile $t"; LVAE_PREC=4 LVAE_H2P=$t LVAE_OUT_H2=1 timeout 300 python tools/microbench.py gemm $B 2>&1 | grep -v amdgpu | grep -v "^s32\|^s64\|total\|---"; done > gpurun_out/r3/tile_all_b$B.txt 2>&1
python - <<PY
import re,collections
rows=collections.OrderedDict(); t=None
for l in open('gpurun_out/r3/tile_all_b$B.txt'):
    m=re.match(r'== tile (\d+)',l)
    if m: t=m.group(1); continue
    m=re.match(r'(s\d+\s+fc\d) M=\s*(\d+) N=\s*(\d+) K=\s*(\d+)\s+([\d.]+) us',l)
    if m: rows.setdefault((m.group(1),m.group(2),m.group(3),m.group(4)),{})[t]=float(m.group(5))
for k,v in rows.items():
    best=min(v,key=v.get)
    print(k, ' '.join(f'{t}:{v[t]:6.1f}' for t in sorted(v)), 'best', best)
PY
