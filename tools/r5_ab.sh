# same-process A/B of encode variants (tools/ab_enc.py): pipeline groups x side stream [x row threshold of the (384, 768) fused MLP]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_ab
mkdir -p $O
cd $R
python tools/ab_enc.py "2,1" "2,1,24576" "2,0" "1,1" 2>&1 | grep -v amdgpu | grep "enc groups" | tee $O/ab_enc_b8_fused384.txt
