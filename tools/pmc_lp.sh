R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmcl_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcl_$i -o p -- python $R/tools/lp_bench.py 3 > /tmp/pmcl_$i.log 2>&1 || { echo "group $i failed"; tail -3 /tmp/pmcl_$i.log; continue; }
  python $R/tools/pmc_summary.py $(find /tmp/pmcl_$i -name "*.db" | head -1) gemm_lp 2>&1
done
