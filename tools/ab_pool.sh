R=$GRAFT_REPO_ROOT
cd $R
run() { # $1 = label, $2 = lib ('' = in-tree), rest = bench args
  lbl=$1; lib=$2; shift 2
  if [ -n "$lib" ]; then export LVAE_LIB=$lib; else unset LVAE_LIB; fi
  LVAE_TIMING=1 python tools/bench_with_lib.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 "$@" 2>/tmp/e.txt >/tmp/o.json
  grep "host phase" /tmp/e.txt | sed 's/.*warm-up)://' | cut -c1-150
  python -c "import sys,json; j=json.loads(open('/tmp/o.json').read()); print('$lbl', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])"
}
for i in 1 2; do run new ""; run old _bin/oldpool/liblvae_hip.so; done
run "B=1 new" "" --batch 1 --steps 30 --warmup 5
run "B=1 old" _bin/oldpool/liblvae_hip.so --batch 1 --steps 30 --warmup 5
