# Round-6 profile set (GPU box, repo root): everything under gpurun_out/prof_r6/ -- copy what is to be judged into profiles/r06_*
# PROF_PART=a: kernel stats (2 groups / serial groups) + roofline-only line + op tables;  b: PMC passes (MFMA util, FETCH, WRITE) on the
# headline and on config 5 + the WRITE_SIZE calibration;  c: dw_bench, decode timelines, kernel sequences
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BX="--no-cpu-baseline --fp32-steps 0 --config5-steps 0 --b1-steps 0 --qres-steps 0 --size-steps 0 --coder-steps 0"
part=${PROF_PART:-a}
if [ "$part" = "a" ]; then
rocprofv3 --kernel-trace --stats -d /tmp/pr_a -o a -- python $R/bench.py --steps 6 --warmup 2 --no-kernel-timing $BX > $O/bench_prof_2groups.json 2>/dev/null
python $R/tools/rocpd_summary.py $(find /tmp/pr_a -name "*.db" | head -1) 40 > $O/kernel_stats_2groups.txt
LVAE_SIDE_STREAMS=0 rocprofv3 --kernel-trace --stats -d /tmp/pr_b -o b -- python $R/bench.py --steps 6 --warmup 2 --serial-groups --no-kernel-timing $BX > $O/bench_prof_serial_groups.json 2>/dev/null
python $R/tools/rocpd_summary.py $(find /tmp/pr_b -name "*.db" | head -1) 40 > $O/kernel_stats_serial_groups.txt
python $R/bench.py --steps 10 --warmup 2 $BX > $O/bench_roofline_only.json 2>/dev/null
python $R/tools/op_times.py 4 2>&1 | grep -v amdgpu > $O/op_times_b4.txt
python $R/tools/op_times.py 1 2>&1 | grep -v amdgpu > $O/op_times_b1.txt
fi
if [ "$part" = "b" ]; then
PB="--steps 2 --warmup 1 --serial-groups --no-kernel-timing $BX"
LVAE_SIDE_STREAMS=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pr_m -o m -- python $R/bench.py $PB > /dev/null 2>&1
python $R/tools/pmc_mfma_util.py $(find /tmp/pr_m -name "*.db" | head -1) > $O/pmc_gemm_mfma_util.txt 2>&1
LVAE_SIDE_STREAMS=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pr_f -o f -- python $R/bench.py $PB > /dev/null 2>&1
LVAE_SIDE_STREAMS=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pr_w -o w -- python $R/bench.py $PB > /dev/null 2>&1
python $R/tools/pmc_traffic.py $(find /tmp/pr_f -name "*.db" | head -1) $(find /tmp/pr_w -name "*.db" | head -1) $O/pmc_gemm_traffic.json > $O/pmc_hbm_traffic.txt
# config 5 (fp8 mode, 4 x 1216x1216) on this tree
P5="--precision fp8 --batch 4 --height 1216 --width 1216 --steps 2 --warmup 1 --serial-groups --no-kernel-timing --no-cpu-baseline"
LVAE_SIDE_STREAMS=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pr_f5 -o f -- python $R/bench.py $P5 > /dev/null 2>&1
LVAE_SIDE_STREAMS=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pr_w5 -o w -- python $R/bench.py $P5 > /dev/null 2>&1
python $R/tools/pmc_traffic.py $(find /tmp/pr_f5 -name "*.db" | head -1) $(find /tmp/pr_w5 -name "*.db" | head -1) $O/pmc_gemm_traffic_fp8_1216.json > $O/pmc_hbm_traffic_fp8_1216.txt
# WRITE_SIZE / FETCH_SIZE against known byte counts in the library's own store patterns
export CALIB_OUT=$O
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pr_cw -o w -- python $R/tools/r6_write_calib.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pr_cf -o f -- python $R/tools/r6_write_calib.py > /dev/null 2>&1
python $R/tools/r6_write_calib_summary.py $O/write_calib_known.json $(find /tmp/pr_cw -name "*.db" | head -1) $(find /tmp/pr_cf -name "*.db" | head -1) > $O/write_size_calibration.txt 2>&1
fi
if [ "$part" = "c" ]; then
python $R/tools/dw_bench.py 2>&1 | grep -v amdgpu > $O/dw_bench.txt
python $R/tools/dec_timeline.py 8 20 2>&1 | grep -v amdgpu > $O/dec_timeline_b8.txt
python $R/tools/dec_timeline.py 1 20 2>&1 | grep -v amdgpu > $O/dec_timeline_b1.txt
fi
