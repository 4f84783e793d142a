# whole-bench A/B on one box: product library against a _bin/<name> build (default: epi_generic), alternating
R=$GRAFT_REPO_ROOT
V=${1:-epi_generic}
O=$R/gpurun_out/r5_epi_bench
mkdir -p $O
cd $R
ARGS="--no-cpu-baseline --no-kernel-timing --fp32-steps 0 --b1-steps 0 --qres-steps 0 --config5-steps 0 --steps 30"
P='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], j["value"], j["ms_per_step"], j["enc_ms_per_step"], j["dec_ms_per_step"])'
for i in 1 2 3; do
  python tools/bench_with_lib.py _bin/$V/liblvae_hip.so $ARGS 2>/dev/null | python -c "$P" $V | tee -a $O/ab_$V.txt
  python bench.py $ARGS 2>/dev/null | python -c "$P" product | tee -a $O/ab_$V.txt
done
