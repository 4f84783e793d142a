for shp in "49152 1536 384" "49152 384 1536" "196608 384 192"; do
for epi in 0 1; do LVAE_PREC=2 LVAE_X3V2_TN=3 python tools/microbench.py gemm1 $shp $epi; done
done
