# raster kernels (16-channel chunks) + zero-copy: tests, op table, decode timeline, bench
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_raster
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_qres.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
python tools/op_times.py 4 2>&1 | grep -v amdgpu | grep -E "total|prior_index|quantize" | tee $O/op_times_b4.txt
python tools/dec_timeline.py 8 20 2>&1 | grep -v "amdgpu\|lvae:" | tee $O/dec_timeline_b8.txt
python tools/dec_timeline.py 1 20 2>&1 | grep -v "amdgpu\|lvae:" | tee $O/dec_timeline_b1.txt
ARGS="--no-cpu-baseline --no-kernel-timing --fp32-steps 0 --qres-steps 0 --config5-steps 0 --steps 30"
P='import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["enc_ms_per_step"], j["dec_ms_per_step"], j["b1"]["enc_ms"], j["b1"]["dec_ms"])'
for i in 1 2 3; do python bench.py $ARGS 2>/dev/null | python -c "$P" | tee -a $O/bench.txt; done
