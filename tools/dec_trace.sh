R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/dectrace
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for what in dec enc; do
  rocprofv3 --kernel-trace -d /tmp/dt_$what -o t -- python $R/tools/dec_trace.py $what 10 8 2>/dev/null | grep "ms per step" | tee -a $O/summary.txt
  ms=$(tail -1 $O/summary.txt | awk '{printf "%d", $2 * 10 - 2}')      # the measured loop = the last 10 steps of the trace
  python $R/tools/gpu_gaps.py $(find /tmp/dt_$what -name "*.db" | head -1) $ms 30 | tee $O/gaps_$what.txt | head -3 | tee -a $O/summary.txt
done
