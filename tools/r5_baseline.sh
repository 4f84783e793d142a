# Round-5 first call: baseline bench on this round's box + kernel timeline of the product configuration (two groups)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_base
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --fp32-steps 0 --config5-steps 0 --qres-steps 0 > $O/bench.json 2> $O/bench.err
tail -c 3000 $O/bench.json
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/r5a -o a -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 --b1-steps 0 --qres-steps 0 > $O/bench_prof.json 2>/dev/null
DB=$(find /tmp/r5a -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB 45 > $O/kernel_stats_2groups.txt
python $R/tools/step_timeline.py $DB 45 > $O/timeline_last45ms.txt
python $R/tools/gpu_gaps.py $DB 45 40 > $O/gaps_last45ms.txt
head -5 $O/gaps_last45ms.txt
bash $R/tools/dec_trace.sh > $O/dec_trace.txt 2>&1
cp $R/gpurun_out/dectrace/* $O/ 2>/dev/null
python $R/tools/op_times.py 8 2>&1 | grep -v amdgpu > $O/op_times_b8.txt
LVAE_GROUPS=2 python $R/tools/op_times.py 8 2>&1 | grep -v amdgpu > $O/op_times_b8_g2.txt
ls -la $O
