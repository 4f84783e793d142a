# the three headline bench lines (default B=8, B=1 latency, fp8 mode), no cpu baseline
R=$GRAFT_REPO_ROOT; cd $R
for a in "" "--batch 1 --steps 60" "--precision fp8" "--precision fp8 --batch 4 --height 1216 --width 1216 --steps 10"; do
  python bench.py --no-cpu-baseline --fp32-steps 0 $a 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$a |', j['value'], j['ms_per_step'], j.get('enc_ms_per_step'), j.get('dec_ms_per_step'), j['roofline'].get('frac'), j['roofline'].get('avg_launch_us'))"
done
