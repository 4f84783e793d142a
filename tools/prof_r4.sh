# Round-4 profile set (GPU box, repo root): everything under gpurun_out/prof_r4/ -- copy what is to be judged into profiles/r04_*
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. default bench line (cpu_baseline, b1, bf16x3 / fp32 mode values, qres34m, config5)
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
# 2. rocprofv3 kernel stats, single stream (what roofline.avg_launch_us is compared with) and product configuration
LVAE_GROUPS=1 rocprofv3 --kernel-trace --stats -d /tmp/pr_b -o b -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --fp32-steps 0 --config5-steps 0 --b1-steps 0 --qres-steps 0 > $O/bench_prof_single_stream.json 2>/dev/null
python $R/tools/rocpd_summary.py $(find /tmp/pr_b -name "*.db" | head -1) 40 > $O/kernel_stats_single_stream.txt
rocprofv3 --kernel-trace --stats -d /tmp/pr_a -o a -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --fp32-steps 0 --config5-steps 0 --b1-steps 0 --qres-steps 0 > $O/bench_prof_2groups.json 2>/dev/null
python $R/tools/rocpd_summary.py $(find /tmp/pr_a -name "*.db" | head -1) 40 > $O/kernel_stats_2groups.txt
# 3. PMC passes (kernel-trace only, one counter group per run): MFMA utilisation of the dominant family at the bench workload, HBM traffic
LVAE_GROUPS=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pr_m -o m -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 --b1-steps 0 --qres-steps 0 > /dev/null 2>&1
python $R/tools/pmc_mfma_util.py $(find /tmp/pr_m -name "*.db" | head -1) > $O/pmc_gemm_h2p_mfma_util.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pr_f -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 --b1-steps 0 --qres-steps 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pr_w -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 --b1-steps 0 --qres-steps 0 > /dev/null 2>&1
python $R/tools/pmc_traffic.py $(find /tmp/pr_f -name "*.db" | head -1) $(find /tmp/pr_w -name "*.db" | head -1) $O/pmc_gemm_traffic.json > $O/pmc_hbm_traffic.txt
# 4. per-op tables
python $R/tools/op_times.py 8 2>&1 | grep -v amdgpu > $O/op_times_b8.txt
python $R/tools/op_times.py 1 2>&1 | grep -v amdgpu > $O/op_times_b1.txt
if [ "${PROF_FULL:-0}" = "1" ]; then
python $R/bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 --b1-steps 0 --qres-steps 0 > $O/bench_b1.json 2>/dev/null
python $R/bench.py --precision fp8 --no-cpu-baseline --batch 4 --height 1216 --width 1216 --steps 8 > $O/bench_fp8_b4_1216x1216.json 2>/dev/null
python $R/bench.py --no-cpu-baseline --batch 4 --height 1216 --width 1216 --steps 8 --fp32-steps 0 --b1-steps 0 --qres-steps 0 > $O/bench_b4_1216x1216.json 2>/dev/null
python $R/tools/dw_bench.py 2>&1 | grep -v amdgpu > $O/dw_bench.txt
fi
ls -la $O
