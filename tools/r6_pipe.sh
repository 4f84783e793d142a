# Round 6: gemm_h2p's half-stage software pipeline (PIPE; experimental build -DH2P_EXP_PIPE, env LVAE_H2P_PIPE: 1 = plain launches, 2 = serial
# split-K launches, 3 = both): same bits?  faster?     bash tools/r6_pipe.sh   (GPU box)
R=$GRAFT_REPO_ROOT
cd $R
# the study code lives in tools/studies/gemm_h2p_stagger3_and_half_stage_pipeline.patch: apply it to a scratch copy of the tree's gemm_h2p.hip first
#   (git apply tools/studies/gemm_h2p_stagger3_and_half_stage_pipeline.patch; build; git checkout lossy-vae_amd/csrc/gemm_h2p.hip)
L=_bin/pipe/liblvae_hip.so
[ -f $L ] || bash tools/build_exp.sh pipe gemm_h2p.hip -DH2P_EXP_PIPE > /dev/null
echo "== equality tests with PIPE on (tests/test_gpu_f16x2.py against the experimental library)"
LVAE_LIB=$L LVAE_H2P_PIPE=3 timeout 900 python -m pytest tests/test_gpu_f16x2.py -x -q 2>&1 | tail -3
echo "== stand-alone launches (tools/microbench.py gemm1), us: PIPE off / on"
for shape in "24576 768 384 1" "24576 384 768 2" "24576 448 256 1" "24576 256 448 2" "98304 384 192 1" "98304 192 384 2" "6144 768 384 1" "6144 384 768 2" "6144 1024 512 1" "6144 512 1024 2" "1536 768 384 1" "1536 384 768 2"; do
  for p in 0 1 0 1; do
    echo -n "pipe $p: "
    LVAE_LIB=$L LVAE_PREC=4 LVAE_H2P=1 LVAE_OUT_H2=1 LVAE_H2P_PIPE=$p python tools/microbench.py gemm1 $shape 2>&1 | grep "TF/s"
  done
done
echo "== serial split-K launches (FOLD): 1536 x 512 x 1024 S=4, 384 x 512 x 1536 S=16 ... via tools/microbench.py gemmsk_b"
for p in 0 2 0 2; do echo "-- LVAE_H2P_PIPE=$p"; LVAE_LIB=$L LVAE_FOLD_LOADERS=0 LVAE_H2P_PIPE=$p python tools/microbench.py gemmsk_b 4 2>&1 | grep -v amdgpu | tail -12; done
