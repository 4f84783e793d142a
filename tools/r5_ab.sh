# same-process A/B of encode variants (tools/ab_enc.py): pipeline groups x side stream
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_ab
mkdir -p $O
cd $R
python tools/ab_enc.py "2,1" "2,0" "1,1" "1,0" 2>&1 | grep -v amdgpu | grep "enc groups" | tee $O/ab_enc_b8.txt
AB_BATCH=1 python tools/ab_enc.py "1,1" "1,0" 2>&1 | grep -v amdgpu | grep "enc groups" | tee $O/ab_enc_b1.txt
