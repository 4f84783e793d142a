R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/h2
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_f16x2.py -x -q > $O/pytest_f16x2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_f16x2.log
tail -5 $O/pytest_f16x2.log
LVAE_PREC=4 python tools/microbench.py gemm 8 2>&1 | grep -v amdgpu > $O/mb_base.txt
for n in "$@"; do LVAE_PREC=4 LVAE_LIB=_bin/$n/liblvae_hip.so python tools/microbench.py gemm 8 2>&1 | grep -v amdgpu > $O/mb_$n.txt; done
paste -d'|' $O/mb_base.txt $(for n in "$@"; do echo $O/mb_$n.txt; done) | cut -c1-250
OP_TIMES_PRECISION=f16x2 python tools/op_times.py 8 2>&1 | grep -v amdgpu > $O/op_times_b8_f16x2.txt
head -50 $O/op_times_b8_f16x2.txt
