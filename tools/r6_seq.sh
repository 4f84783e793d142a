# launch-by-launch picture of ONE decode (kernel trace, last step) with the fused small-map MLP on / off: gpurun_out/r6_seq/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6_seq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=${SEQ_B:-1}
for v in 0 1024; do
  for what in dec enc; do
  LVAE_MLP_SK_MAX_ROWS=$v rocprofv3 --kernel-trace -d /tmp/sq_${what}_$v -o t -- python $R/tools/dec_trace.py $what 10 $B 2>/dev/null | grep "ms per step" | tee -a $O/summary_b$B.txt
  ms=$(tail -1 $O/summary_b$B.txt | awk '{printf "%.2f", $2 * 0.97}')
  python $R/tools/kernel_seq.py $(find /tmp/sq_${what}_$v -name "*.db" | head -1) $ms > $O/seq_${what}_b${B}_sk$v.txt
  tail -1 $O/seq_${what}_b${B}_sk$v.txt
  done
done
