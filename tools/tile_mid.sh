#!/bin/bash
# mid-size M (one pipeline group's stride-16 layers, B = 4 / 2 / 1): 128 x 128 against 128 x 64 tiles of gemm_h2p
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3
for shp in "6144 384 768 2" "6144 768 384 1" "6144 512 1024 2" "6144 1024 512 1" "3072 384 768 2" "3072 768 384 1" "1536 384 768 2" "1536 768 384 1" "1536 512 1024 2" "1536 1024 512 1" "12288 384 768 2" "12288 512 1024 2" "24576 192 384 2" "24576 384 192 1" "24576 128 192 2" "24576 192 128 1" "6144 256 448 2" "6144 448 256 1"; do
  for t in 22 21 42 41; do echo -n "tile $t: "; LVAE_PREC=4 LVAE_H2P=$t LVAE_OUT_H2=1 timeout 120 python tools/microbench.py gemm1 $shp 2>&1 | grep -v amdgpu; done
done | tee gpurun_out/r3/tile_mid.txt
