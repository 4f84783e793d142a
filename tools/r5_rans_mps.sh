# Round 5: the host decoder's most-probable-symbol path (csrc/rans_host.cpp) against the library before it (_bin/rans_before_mps: the
# same kernels, the previous rans_host.cpp), alternating on one box -> profiles/r05_rans_mps_path.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_rans_mps
mkdir -p $O
cd $R
OLD=_bin/rans_before_mps/liblvae_hip.so
BX="--no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 --qres-steps 0"
pick='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=j.get("b1") or {}; print(sys.argv[1], "B=8: %.1f Mpixels/s = %.3f ms = enc %.3f + dec %.3f" % (j["value"], j["ms_per_step"], j["enc_ms_per_step"], j["dec_ms_per_step"]), "| B=1: enc %.3f + dec %.3f ms" % (b.get("enc_ms", 0), b.get("dec_ms", 0)))'
for i in 1 2 3; do
  python tools/bench_with_lib.py $OLD $BX 2>/dev/null | python -c "$pick" before >> $O/ab.txt
  python bench.py $BX 2>/dev/null | python -c "$pick" after >> $O/ab.txt
done
cat $O/ab.txt
for B in 8 1; do
  echo "== before (B=$B)" >> $O/timeline.txt; LVAE_LIB=$OLD python tools/dec_timeline.py $B 20 2>&1 | grep -v amdgpu >> $O/timeline.txt
  echo "== after (B=$B)" >> $O/timeline.txt; python tools/dec_timeline.py $B 20 2>&1 | grep -v amdgpu >> $O/timeline.txt
done
pick5='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "fp8 4x1216x1216: %.1f Mpixels/s = enc %.3f + dec %.3f ms" % (j["value"], j["enc_ms_per_step"], j["dec_ms_per_step"]))'
F8="--precision fp8 --no-cpu-baseline --batch 4 --height 1216 --width 1216 --steps 8 --no-kernel-timing"
for i in 1 2; do
  python tools/bench_with_lib.py $OLD $F8 2>/dev/null | python -c "$pick5" before >> $O/ab_fp8.txt
  python bench.py $F8 2>/dev/null | python -c "$pick5" after >> $O/ab_fp8.txt
done
cat $O/ab_fp8.txt
