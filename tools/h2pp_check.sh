#!/bin/bash
# persistent f16x2 GEMM (tools/studies/gemm_h2pp.hip): bit-equality tests, then the microbench shapes per tile code on one box
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/h2pp
timeout 900 python -m pytest tests/test_gpu_f16x2.py -q -x -k "h2p" 2>&1 | tail -5 > gpurun_out/h2pp/tests.txt
cat gpurun_out/h2pp/tests.txt
run() { echo "== $*"; env "$@" LVAE_PREC=4 timeout 600 python tools/microbench.py gemm 2>&1 | grep -v amdgpu; env "$@" LVAE_PREC=4 timeout 600 python tools/microbench.py gemmx 2>&1 | grep -v amdgpu; }
for t in ${TILES:-22 92 42}; do run LVAE_H2P=$t LVAE_OUT_H2=1; done > gpurun_out/h2pp/bench.txt 2>&1
cat gpurun_out/h2pp/bench.txt
