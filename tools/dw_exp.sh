#!/bin/bash
# depthwise+LN kernel: in-tree library vs the _bin/dw_* ablations (wrong results by construction: timing only)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3
{ echo "== base"; python tools/dw_bench.py 2>&1 | grep -v amdgpu; for n in "$@"; do echo "== $n"; LVAE_LIB=_bin/$n/liblvae_hip.so python tools/dw_bench.py 2>&1 | grep -v amdgpu; done; } | tee gpurun_out/r3/dw_exp.txt
