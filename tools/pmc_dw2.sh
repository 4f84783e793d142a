# second PMC set for one depthwise+LN shape (DW_ONLY index of tools/dw_bench.py): instruction fetch, outstanding memory ops, scalar work
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_BUSY_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmce_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmce_$i -o p -- env DW_ONLY=${DW_ONLY:-3} python $R/tools/dw_bench.py 3 > /tmp/pmce_$i.log 2>&1 || { echo "group $i failed: $grp"; tail -3 /tmp/pmce_$i.log; continue; }
  python $R/tools/pmc_summary.py $(find /tmp/pmce_$i -name "*.db" | head -1) "${PMC_FILTER:-false>}" 2>&1 | grep -v "^void"
done
