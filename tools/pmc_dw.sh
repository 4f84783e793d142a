# PMC passes over one depthwise+LN shape (DW_ONLY index of tools/dw_bench.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmcd_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmcd_$i -o p -- env DW_ONLY=${DW_ONLY:-3} python $R/tools/dw_bench.py 3 > /tmp/pmcd_$i.log 2>&1 || { echo "group $i failed"; tail -3 /tmp/pmcd_$i.log; continue; }
  python $R/tools/pmc_summary.py $(find /tmp/pmcd_$i -name "*.db" | head -1) dwconv 2>&1
done
