# Round 6: three-phase start of gemm_h2p's first generation of workgroups (H2P_EXP_STAGGER3, experimental build): does breaking the lock-step
# of the co-resident workgroups make main loops and epilogues overlap?   bash tools/r6_stagger3.sh  (GPU box; the build happens here too)
R=$GRAFT_REPO_ROOT
cd $R
# the study code lives in tools/studies/gemm_h2p_stagger3_and_half_stage_pipeline.patch: apply it to a scratch copy of the tree's gemm_h2p.hip first
#   (git apply tools/studies/gemm_h2p_stagger3_and_half_stage_pipeline.patch; build; git checkout lossy-vae_amd/csrc/gemm_h2p.hip)
[ -f _bin/stg3/liblvae_hip.so ] || bash tools/build_exp.sh stg3 gemm_h2p.hip -DH2P_EXP_STAGGER3 > /dev/null
for shape in "24576 768 384 1" "24576 384 768 2" "24576 448 256 1" "24576 256 448 2" "98304 384 192 1" "98304 192 384 2" "49152 768 384 1"; do
  for s in 0 4 8 12 16 22 30; do
    echo -n "stagger $s: "
    LVAE_LIB=_bin/stg3/liblvae_hip.so LVAE_PREC=4 LVAE_H2P=1 LVAE_OUT_H2=1 LVAE_H2P_STAGGER=$s python tools/microbench.py gemm1 $shape 2>&1 | grep "TF/s"
  done
done
