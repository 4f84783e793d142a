#!/bin/bash
# stride-4 MLP GEMMs (pre-split operands): tile sweep, or (arg "out") fp32 vs pre-split output stores
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/s4
if [ "$1" = "out" ]; then
  for o in 0 1 0 1; do echo "== out_h2 $o"; LVAE_PREC=4 LVAE_H2P=1 LVAE_OUT_H2=$o timeout 300 python tools/microbench.py gemms4 2>&1 | grep -v amdgpu | grep "epi=1"; done > gpurun_out/s4/out.txt 2>&1
  cat gpurun_out/s4/out.txt; exit 0
fi
for t in 21 22 41 42 1; do echo "== tile $t"; LVAE_PREC=4 LVAE_H2P=$t LVAE_OUT_H2=1 timeout 300 python tools/microbench.py gemms4 2>&1 | grep -v amdgpu; done > gpurun_out/s4/tiles.txt 2>&1
cat gpurun_out/s4/tiles.txt
