# Round-5 profile set (GPU box, repo root): everything under gpurun_out/prof_r5/ -- copy what is to be judged into profiles/r05_*
# PROF_PART=a: bench line + kernel stats + op tables;  b: PMC passes;  c: config 5, B=1, dw_bench, qres
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BX="--no-cpu-baseline --fp32-steps 0 --config5-steps 0 --b1-steps 0 --qres-steps 0"
part=${PROF_PART:-a}
if [ "$part" = "a" ]; then
# 1. default bench line (cpu_baseline, b1, bf16x3 / fp32 mode values, qres34m, config5)
python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
# 2. rocprofv3 kernel stats: product configuration (two groups, concurrent) and the roofline pass's configuration (same plans, groups and
#    side-stream branches one after the other: what roofline.avg_launch_us is compared with)
rocprofv3 --kernel-trace --stats -d /tmp/pr_a -o a -- python $R/bench.py --steps 6 --warmup 2 --no-kernel-timing $BX > $O/bench_prof_2groups.json 2>/dev/null
python $R/tools/rocpd_summary.py $(find /tmp/pr_a -name "*.db" | head -1) 40 > $O/kernel_stats_2groups.txt
LVAE_SIDE_STREAMS=0 rocprofv3 --kernel-trace --stats -d /tmp/pr_b -o b -- python $R/bench.py --steps 6 --warmup 2 --serial-groups --no-kernel-timing $BX > $O/bench_prof_serial_groups.json 2>/dev/null
python $R/tools/rocpd_summary.py $(find /tmp/pr_b -name "*.db" | head -1) 40 > $O/kernel_stats_serial_groups.txt
python $R/bench.py --steps 10 --warmup 2 $BX > $O/bench_roofline_only.json 2>/dev/null
# 3. per-op tables: one 8-image group, one 4-image group (the launches of a product group)
python $R/tools/op_times.py 8 2>&1 | grep -v amdgpu > $O/op_times_b8.txt
python $R/tools/op_times.py 4 2>&1 | grep -v amdgpu > $O/op_times_b4.txt
python $R/tools/op_times.py 1 2>&1 | grep -v amdgpu > $O/op_times_b1.txt
fi
if [ "$part" = "b" ]; then
# 4. PMC passes (kernel-trace only, one counter group per run) on the roofline pass's configuration: MFMA utilisation, HBM traffic
PB="--steps 2 --warmup 1 --serial-groups --no-kernel-timing $BX"
LVAE_SIDE_STREAMS=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pr_m -o m -- python $R/bench.py $PB > /dev/null 2>&1
python $R/tools/pmc_mfma_util.py $(find /tmp/pr_m -name "*.db" | head -1) > $O/pmc_gemm_mfma_util.txt 2>&1
LVAE_SIDE_STREAMS=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pr_f -o f -- python $R/bench.py $PB > /dev/null 2>&1
LVAE_SIDE_STREAMS=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pr_w -o w -- python $R/bench.py $PB > /dev/null 2>&1
python $R/tools/pmc_traffic.py $(find /tmp/pr_f -name "*.db" | head -1) $(find /tmp/pr_w -name "*.db" | head -1) $O/pmc_gemm_traffic.json > $O/pmc_hbm_traffic.txt
fi
if [ "$part" = "c" ]; then
python $R/bench.py --batch 1 --steps 30 --warmup 5 --no-kernel-timing $BX > $O/bench_b1.json 2>/dev/null
python $R/bench.py --precision fp8 --no-cpu-baseline --batch 4 --height 1216 --width 1216 --steps 8 > $O/bench_fp8_b4_1216x1216.json 2>/dev/null
python $R/bench.py --no-cpu-baseline --batch 4 --height 1216 --width 1216 --steps 8 --fp32-steps 0 --b1-steps 0 --qres-steps 0 --config5-steps 0 > $O/bench_b4_1216x1216.json 2>/dev/null
python $R/tools/dw_bench.py 2>&1 | grep -v amdgpu > $O/dw_bench.txt
LVAE_MODEL=qres34m python $R/tools/op_times.py 8 2>&1 | grep -v amdgpu > $O/op_times_qres34m_b8.txt
python $R/tools/dec_timeline.py 8 20 2>&1 | grep -v amdgpu > $O/dec_timeline_b8.txt
python $R/tools/dec_timeline.py 1 20 2>&1 | grep -v amdgpu > $O/dec_timeline_b1.txt
python $R/tools/enc_tail.py 8 2>&1 | grep -v amdgpu | tail -1 > $O/enc_tail.txt
python $R/tools/enc_tail.py 1 2>&1 | grep -v amdgpu | tail -1 >> $O/enc_tail.txt
LVAE_TIMING=1 python $R/tools/host_overhead.py 8 30 2>&1 | grep -v "amdgpu\|^lvae:" > $O/host_overhead.txt
LVAE_TIMING=1 python $R/tools/host_overhead.py 1 30 2>&1 | grep -v "amdgpu\|^lvae:" >> $O/host_overhead.txt
fi
ls -la $O
