#!/bin/bash
# Builds the micro-benchmarks / in-kernel timeline harnesses next to their sources (binaries are git-ignored; they travel to the
# GPU box with the gpurun snapshot).   bash tools/ubench/build.sh [name ...]
cd "$(dirname "$0")"
names=("$@"); [ ${#names[@]} -eq 0 ] && names=($(ls *.hip | sed 's/\.hip$//'))
for n in "${names[@]}"; do
  echo "hipcc $n.hip" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -o "$n" "$n.hip" || echo "  (failed: $n)"
done
