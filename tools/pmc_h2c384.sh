# PMC counter groups (separate passes, kernel-trace only) for mlp_h2c_kernel<384, 768, 128, 64, 3, 4> and <192, 384, 128, 128, 1, 3>: tools/microbench.py mlpf
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_h2c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for shape in 384,768:49152 192,384:196608; do
  sh=${shape%%:*}; ms=${shape##*:}
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
    i=$((i+1))
    rm -rf /tmp/pmch_$i
    LVAE_MLP_SHAPE=$sh LVAE_MLP_MS=$ms timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmch_$i -o p -- python $R/tools/microbench.py mlpf > /tmp/pmch_$i.log 2>&1 || { echo "group $i failed"; tail -3 /tmp/pmch_$i.log; continue; }
    python $R/tools/pmc_summary.py $(find /tmp/pmch_$i -name "*.db" | head -1) mlp_h2c 2>&1
  done
done | tee $O/pmc.txt
