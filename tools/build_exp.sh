#!/bin/bash
# Build an experimental copy of liblvae_hip.so: tools/build_exp.sh <name> <file.hip> <extra hipcc flags...>
# -> _bin/<name>/liblvae_hip.so (only <file.hip> is recompiled; the other objects come from the in-tree build).
# Run a tool against it with LVAE_LIB=_bin/<name>/liblvae_hip.so (tools/microbench.py, tools/dw_bench.py honour it).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift 2
N=$R/lossy-vae_amd/lvae/_native
mkdir -p $R/_bin/$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLVAE_EXPERIMENTAL_BUILD "$@" -c $R/lossy-vae_amd/csrc/$src -o $R/_bin/$name/$src.o
objs=""
for o in gemm_f32.hip gemm_f32_patch2.hip gemm_f32_conv3.hip gemm_x3v2.hip gemm_h2.hip gemm_h2p.hip gemm_h2n.hip mlp_h2c.hip mlp_sk.hip gemm_lp.hip gemm_q8.hip pointwise.hip dwconv_cl.hip dwconv_cl_bf16.hip dwconv_cl_h2.hip dwconv_cl_q8.hip rans_host.cpp plan_runtime.cpp; do
  if [ "$o" = "$src" ]; then objs="$objs $R/_bin/$name/$src.o"; else objs="$objs $N/$o.o"; fi
done
# EXTRA_SRC="a.hip b.hip": study-only translation units (tools/studies/: e.g. gemm_h2pp.hip, which gemm_h2p.hip calls when built with
# -DLVAE_EXP_H2PP) that are not part of the product library, compiled with the same flags and linked in
for x in $EXTRA_SRC; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DLVAE_EXPERIMENTAL_BUILD "$@" -c $R/tools/studies/$x -o $R/_bin/$name/$x.o
  objs="$objs $R/_bin/$name/$x.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/_bin/$name/liblvae_hip.so $objs -lpthread
echo $R/_bin/$name/liblvae_hip.so
