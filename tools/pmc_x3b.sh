# second PMC set: latencies / levels / stall reasons for one x3 GEMM shape (args: M N K epi)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=100
for grp in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  LVAE_PREC=${LVAE_PREC:-2} timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_$i -o p -- python $R/tools/microbench.py gemm1 "$@" > /tmp/pmc_$i.log 2>&1 || { echo "group $i failed: $grp"; tail -2 /tmp/pmc_$i.log; continue; }
  python $R/tools/pmc_summary.py $(find /tmp/pmc_$i -name "*.db" | head -1) gemm 2>&1 | grep -v "^void"
done
