# zero-copy coder arrays (LVAE_ZERO_COPY: '' = copies, dec, both): kernel tests, model tests, decode timelines and the bench alternating
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_zero_copy
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
LVAE_ZERO_COPY=both python -m pytest tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O/tests.txt
for Z in none dec both; do
  echo "== LVAE_ZERO_COPY=$Z" | tee -a $O/dec_timeline.txt
  LVAE_ZERO_COPY=$Z python tools/dec_timeline.py 8 20 2>&1 | grep -v "amdgpu\|lvae:" | tee -a $O/dec_timeline.txt
done
ARGS="--no-cpu-baseline --no-kernel-timing --fp32-steps 0 --qres-steps 0 --config5-steps 0 --steps 30"
P='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], j["value"], j["ms_per_step"], j["enc_ms_per_step"], j["dec_ms_per_step"], j["b1"])'
for i in 1 2; do for Z in none dec both; do
  LVAE_ZERO_COPY=$Z python bench.py $ARGS 2>/dev/null | python -c "$P" $Z | tee -a $O/bench.txt
done; done
