R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/h2n
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_qres.py tests/test_gpu_configs.py -x -q 2>&1 | tail -12 | tee $O/pytest_model.txt
cd /tmp
python $R/tools/qres_speed.py qres34m 8 2>&1 | grep -v amdgpu | tail -1 | tee $O/qres_speed.txt
python $R/tools/qres_speed.py qres34m 1 2>&1 | grep -v amdgpu | tail -1 | tee -a $O/qres_speed.txt
LVAE_MODEL=qres34m OP_TIMES_TOP=25 python $R/tools/op_times.py 8 2>&1 | grep -v amdgpu > $O/op_times_qres34m_b8.txt; head -30 $O/op_times_qres34m_b8.txt | cut -c1-120
