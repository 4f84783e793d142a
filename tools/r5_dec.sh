R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_dec
mkdir -p $O
python $R/tools/dec_timeline.py 8 20 2>&1 | grep -v amdgpu | tee $O/dec_timeline_b8.txt
python $R/tools/dec_timeline.py 1 20 2>&1 | grep -v amdgpu | tee $O/dec_timeline_b1.txt
python $R/tools/dec_timeline.py 4 20 2>&1 | grep -v amdgpu | tee $O/dec_timeline_b4.txt
