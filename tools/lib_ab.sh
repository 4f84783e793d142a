# same-box A/B of the product library vs an experimental build: tools/lib_ab.sh <name> [bench args]
R=$GRAFT_REPO_ROOT; cd $R; n=$1; shift
for i in 1 2 3; do
  python tools/bench_with_lib.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('product', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])"
  LVAE_LIB=_bin/$n/liblvae_hip.so python tools/bench_with_lib.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$n', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])"
done
