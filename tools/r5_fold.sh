# straight-line tail of the serial split-K form (gemm_h2p FOLD): tests, serial-vs-parallel micro-benchmark (product | epi_generic), bench A/B
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_fold
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_f16x2.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -2 | tee $O/tests.txt
for V in product epi_generic; do
  LIB=""; [ $V != product ] && LIB=$R/_bin/$V/liblvae_hip.so
  for B in 4 8 1; do
    echo "== $V B=$B" | tee -a $O/gemmsk.txt
    LVAE_LIB=$LIB python tools/microbench.py gemmsk_b $B 2>&1 | grep rows | tee -a $O/gemmsk.txt
  done
done
bash tools/r5_epi_bench.sh epi_generic
python tools/dec_timeline.py 8 20 2>&1 | grep -v "amdgpu\|lvae:" | head -3 | tee $O/dec.txt
python tools/dec_timeline.py 1 20 2>&1 | grep -v "amdgpu\|lvae:" | head -3 | tee -a $O/dec.txt
