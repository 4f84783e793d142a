# config 5: straight-line gemm_q8 epilogues -- fp8 tests, then the fp8 bench (4 x 1216 x 1216) product | generic alternating
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_q8
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_fp8.py -x -q -m gpu 2>&1 | tail -2 | tee $O/tests.txt
ARGS="--precision fp8 --no-cpu-baseline --no-kernel-timing --batch 4 --height 1216 --width 1216 --steps 8"
P='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], j["value"], j["ms_per_step"], j["enc_ms_per_step"], j["dec_ms_per_step"])'
for i in 1 2 3; do
  python tools/bench_with_lib.py _bin/q8_generic/liblvae_hip.so $ARGS 2>/dev/null | python -c "$P" q8_generic | tee -a $O/ab.txt
  python bench.py $ARGS 2>/dev/null | python -c "$P" product | tee -a $O/ab.txt
done
