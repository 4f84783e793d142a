# same-box A/B of the product library against _bin/<name>/liblvae_hip.so: tools/ab_lib2.sh <name> [bench args]; prints value, ms, enc, dec, b1
R=$GRAFT_REPO_ROOT; cd $R; n=$1; shift
show='import sys,json; j=json.loads(sys.stdin.read()); b=j.get("b1") or {}; print(sys.argv[1], j["value"], j["ms_per_step"], j["enc_ms_per_step"], j["dec_ms_per_step"], "b1", b.get("enc_ms"), b.get("dec_ms"))'
for i in 1 2 3; do
  python tools/bench_with_lib.py lossy-vae_amd/lvae/_native/liblvae_hip.so --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --qres-steps 0 --config5-steps 0 "$@" 2>/dev/null | python -c "$show" product
  python tools/bench_with_lib.py _bin/$n/liblvae_hip.so --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --qres-steps 0 --config5-steps 0 "$@" 2>/dev/null | python -c "$show" $n
done
