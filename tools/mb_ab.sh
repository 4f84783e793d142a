R=$GRAFT_REPO_ROOT
echo "== OLD forced TN=2"; (cd $R/_old && LVAE_PREC=2 LVAE_X3V2_TN=2 python tools/microbench.py gemm 8 2>&1 | grep -v amdgpu | tail -24)
echo "== NEW forced TN=2 (3 WGs/CU)"; (cd $R && LVAE_PREC=2 LVAE_X3V2_TN=2 python tools/microbench.py gemm 8 2>&1 | grep -v amdgpu | tail -24)
