# Round-3 profile set (GPU box, repo root): everything under gpurun_out/prof_r3/ -- copy what is to be judged into profiles/r03_*
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. default bench line (cpu_baseline, config5_value, fp32 mode value included)
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
# 2. rocprofv3 kernel stats, product configuration (two pipeline groups) and single stream
rocprofv3 --kernel-trace --stats -d /tmp/pr_a -o a -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --fp32-steps 0 --config5-steps 0 > $O/bench_prof_2groups.json 2>/dev/null
python $R/tools/rocpd_summary.py $(find /tmp/pr_a -name "*.db" | head -1) 40 > $O/kernel_stats_2groups.txt
LVAE_GROUPS=1 rocprofv3 --kernel-trace --stats -d /tmp/pr_b -o b -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --fp32-steps 0 --config5-steps 0 > $O/bench_prof_single_stream.json 2>/dev/null
python $R/tools/rocpd_summary.py $(find /tmp/pr_b -name "*.db" | head -1) 40 > $O/kernel_stats_single_stream.txt
# 3. HBM traffic: separate PMC passes (kernel-trace only)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pr_f -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pr_w -o w -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 > /dev/null 2>&1
python $R/tools/pmc_traffic.py $(find /tmp/pr_f -name "*.db" | head -1) $(find /tmp/pr_w -name "*.db" | head -1) $O/pmc_gemm_traffic.json > $O/pmc_hbm_traffic.txt
# 4. reduced-precision mode (config 5)
python $R/bench.py --precision fp8 --no-cpu-baseline > $O/bench_fp8_b8_512x768.json 2>/dev/null
python $R/bench.py --precision fp8 --no-cpu-baseline --batch 4 --height 1216 --width 1216 --steps 8 > $O/bench_fp8_b4_1216x1216.json 2>/dev/null
python $R/bench.py --no-cpu-baseline --batch 4 --height 1216 --width 1216 --steps 8 --fp32-steps 0 > $O/bench_b4_1216x1216.json 2>/dev/null
LVAE_GROUPS=1 rocprofv3 --kernel-trace --stats -d /tmp/pr_c -o c -- python $R/bench.py --precision fp8 --batch 4 --height 1216 --width 1216 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench_prof_fp8_1216_single_stream.json 2>/dev/null
python $R/tools/rocpd_summary.py $(find /tmp/pr_c -name "*.db" | head -1) 30 > $O/kernel_stats_fp8_1216_single_stream.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pr_g -o g -- python $R/bench.py --precision fp8 --batch 4 --height 1216 --width 1216 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pr_h -o h -- python $R/bench.py --precision fp8 --batch 4 --height 1216 --width 1216 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
python $R/tools/pmc_traffic.py $(find /tmp/pr_g -name "*.db" | head -1) $(find /tmp/pr_h -name "*.db" | head -1) $O/pmc_gemm_traffic_fp8_1216.json > $O/pmc_hbm_traffic_fp8_1216.txt
# 5. other operating points
python $R/bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing --fp32-steps 0 > $O/bench_b1.json 2>/dev/null
python $R/bench.py --precision bf16x3 --no-cpu-baseline --config5-steps 0 > $O/bench_bf16x3_mode.json 2>/dev/null
python $R/bench.py --precision fp32 --no-cpu-baseline > $O/bench_fp32_mode.json 2>/dev/null
# 6. per-op tables
python $R/tools/op_times.py 8 2>&1 | grep -v amdgpu > $O/op_times_b8.txt
python $R/tools/op_times.py 1 2>&1 | grep -v amdgpu > $O/op_times_b1.txt
OP_TIMES_PRECISION=fp8 python $R/tools/op_times.py 8 2>&1 | grep -v amdgpu > $O/op_times_fp8.txt
OP_TIMES_PRECISION=bf16x3 python $R/tools/op_times.py 8 2>&1 | grep -v amdgpu > $O/op_times_bf16x3.txt
python $R/tools/dw_bench.py 2>&1 | grep -v amdgpu > $O/dw_bench.txt
ls -la $O
