# mlp_h2c ablations (experimental builds, wrong results by construction) + PMC passes of the product kernel
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/h2c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in product NOGELU NOADMA NOWDMA NOMFMA NOEPI NODSR NOBAR; do
  if [ $v = product ]; then lib=""; else lib=$R/_bin/h2c_$v/liblvae_hip.so; fi
  echo "== $v"
  LVAE_LIB=$lib LVAE_MLP_SHAPE=192,384 timeout 200 python $R/tools/microbench.py mlpf 2>&1 | grep "M= 196608\|M=  98304" | sed 's/fc1 + fc2 launches//'
done | tee $O/ablate.txt
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmch_$i
  LVAE_MLP_SHAPE=192,384 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmch_$i -o p -- python $R/tools/microbench.py mlpf > /tmp/pmch_$i.log 2>&1 || { echo "group $i failed"; tail -3 /tmp/pmch_$i.log; continue; }
  python $R/tools/pmc_summary.py $(find /tmp/pmch_$i -name "*.db" | head -1) mlp_h2c 2>&1
done | tee $O/pmc_mlp_h2c.txt
