# launch-by-launch picture of ONE B = 1 decode and encode (kernel trace, last step): gpurun_out/r5_seq/
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_seq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for what in dec enc; do
  rocprofv3 --kernel-trace -d /tmp/sq_$what -o t -- python $R/tools/dec_trace.py $what 10 ${SEQ_B:-1} 2>/dev/null | grep "ms per step" | tee -a $O/summary.txt
  ms=$(tail -1 $O/summary.txt | awk '{printf "%.2f", $2 * 0.97}')
  python $R/tools/kernel_seq.py $(find /tmp/sq_$what -name "*.db" | head -1) $ms > $O/seq_$what.txt
  tail -1 $O/seq_$what.txt
done
