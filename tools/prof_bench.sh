R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_x3 -o x3 -- python $R/bench.py --no-cpu-baseline --no-kernel-timing --steps 5 --warmup 2 > $R/gpurun_out/prof_x3.log 2>&1
cd $R
find gpurun_out/prof_x3 -name "*.db" | head
python tools/rocpd_summary.py $(find gpurun_out/prof_x3 -name "*.db" | head -1) 30 2>&1 | tee gpurun_out/x3_summary.txt
find gpurun_out/prof_x3 -name "*.db" -delete
LVAE_TIMING=1 python bench.py --no-cpu-baseline --no-kernel-timing --steps 3 --warmup 2 2>&1 | tail -25 | cut -c1-300
