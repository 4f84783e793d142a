R=$GRAFT_REPO_ROOT
cd /tmp
LVAE_LIB=$R/_bin/h2c_TRACE/liblvae_hip.so timeout 200 python $R/tools/microbench.py mlptrace 2>&1 | grep -v amdgpu | tee $R/gpurun_out/h2c/trace.txt
