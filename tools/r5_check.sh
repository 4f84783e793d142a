# Round-5 check: GPU tests (fail fast), smoke, then the bench line (short extras) alternating with the previous GELU's library when present
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_check
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest_gpu.txt
python __graft_entry__.py smoke 2>&1 | grep smoke | tee $O/smoke.txt
BX="--steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 --qres-steps 0"
for r in 1 2 3; do
  python bench.py $BX 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('product      :', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'], 'b1', j['b1']['enc_ms'], j['b1']['dec_ms'])" | tee -a $O/ab_lib.txt
  [ -f _bin/gelu_two_branch/liblvae_hip.so ] && python tools/bench_with_lib.py _bin/gelu_two_branch/liblvae_hip.so $BX 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('two-branch lib:', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'], 'b1', j['b1']['enc_ms'], j['b1']['dec_ms'])" | tee -a $O/ab_lib.txt
done
