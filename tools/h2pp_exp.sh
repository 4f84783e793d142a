#!/bin/bash
# timing studies of the persistent GEMM's main loop: in-tree library vs the _bin/pp_* ablations (wrong results by construction)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/h2pp
run() { echo "== $*"; env "$@" LVAE_PREC=4 LVAE_H2P=${TILE:-92} timeout 600 python tools/microbench.py gemmx 2>&1 | grep -v amdgpu | grep -E "K= 4096 epi=0|N=  768 K=  384 epi=0"; }
{ run A=1; for n in "$@"; do run LVAE_LIB=_bin/$n/liblvae_hip.so; done; } > gpurun_out/h2pp/exp.txt 2>&1
cat gpurun_out/h2pp/exp.txt
