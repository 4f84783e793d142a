# gpurun_out/prof_r4/* (tools/prof_r4.sh, PROF_FULL=1) -> profiles/r04_* (what is committed and judged)
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/prof_r4
cp $S/bench_default.json profiles/r04_bench_default_f16x2.json
cp $S/kernel_stats_single_stream.txt profiles/r04_bench_b8_512x768_kernel_stats_single_stream.txt
cp $S/kernel_stats_2groups.txt profiles/r04_bench_b8_512x768_kernel_stats_2groups.txt
cp $S/pmc_gemm_h2p_mfma_util.txt profiles/r04_pmc_gemm_h2p_mfma_util.txt
cp $S/pmc_gemm_traffic.json profiles/r04_pmc_gemm_traffic.json
cp $S/pmc_hbm_traffic.txt profiles/r04_pmc_hbm_traffic.txt
cp $S/op_times_b8.txt profiles/r04_op_times_b8_512x768_f16x2.txt
cp $S/op_times_b1.txt profiles/r04_op_times_b1_512x768_f16x2.txt
[ -f $S/bench_b1.json ] && cp $S/bench_b1.json profiles/r04_bench_b1.json
[ -f $S/bench_fp8_b4_1216x1216.json ] && cp $S/bench_fp8_b4_1216x1216.json profiles/r04_bench_fp8_b4_1216x1216.json
[ -f $S/bench_b4_1216x1216.json ] && cp $S/bench_b4_1216x1216.json profiles/r04_bench_b4_1216x1216.json
[ -f $S/dw_bench.txt ] && cp $S/dw_bench.txt profiles/r04_dw_bench_dwconv_cl.txt
[ -f gpurun_out/parity_report.json ] && cp gpurun_out/parity_report.json profiles/r04_parity_report.json
ls profiles/r04_* | wc -l
