# PMC passes over one x3 GEMM shape (args: M N K epi).  Separate rocprofv3 runs per counter group, kernel-trace only.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  LVAE_PREC=${LVAE_PREC:-2} timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_$i -o p -- python $R/tools/microbench.py gemm1 "$@" > /tmp/pmc_$i.log 2>&1 || { echo "group $i failed"; tail -3 /tmp/pmc_$i.log; continue; }
  python $R/tools/pmc_summary.py $(find /tmp/pmc_$i -name "*.db" | head -1) gemm 2>&1
done
