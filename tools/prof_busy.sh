# GPU busy fraction / idle gaps of the product configuration: rocprofv3 kernel trace of a bench run WITHOUT the roofline pass
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pr_busy
rocprofv3 --kernel-trace -d /tmp/pr_busy -o a -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" > /tmp/pr_busy.json 2>/dev/null
tail -1 /tmp/pr_busy.json | cut -c1-170
python $R/tools/gpu_gaps.py $(find /tmp/pr_busy -name "*.db" | head -1) 0.25 30
