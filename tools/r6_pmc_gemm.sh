# Round 6: where the waves of the pre-split MLP GEMMs (gemm_h2p) spend their cycles -- three PMC passes over tools/microbench.py gemm at batch 4
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export LVAE_PREC=4 LVAE_H2P=1 LVAE_OUT_H2=1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d /tmp/pg_a -o a -- python $R/tools/microbench.py gemm 4 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL -d /tmp/pg_b -o b -- python $R/tools/microbench.py gemm 4 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS -d /tmp/pg_c -o c -- python $R/tools/microbench.py gemm 4 > /dev/null 2>&1
for p in a b c; do echo "=== pass $p"; python $R/tools/pmc_dump.py $(find /tmp/pg_$p -name "*.db" | head -1) "gemm_h2p"; done > $O/pmc_gemm_h2p_wave_states.txt 2>&1
cat $O/pmc_gemm_h2p_wave_states.txt
