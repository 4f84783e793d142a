# PMC counter groups (separate passes, kernel-trace only) for the fused MLP kernel and the two launches it replaces: tools/microbench.py mlpf
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmcm_$i
  LVAE_PREC=4 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcm_$i -o p -- python $R/tools/microbench.py mlpf > /tmp/pmcm_$i.log 2>&1 || { echo "group $i failed"; tail -3 /tmp/pmcm_$i.log; continue; }
  python $R/tools/pmc_summary.py $(find /tmp/pmcm_$i -name "*.db" | head -1) mlp_h2f 2>&1
  python $R/tools/pmc_summary.py $(find /tmp/pmcm_$i -name "*.db" | head -1) gemm_h2p 2>&1
done | tee $R/gpurun_out/r3/pmc_mlpf.txt
