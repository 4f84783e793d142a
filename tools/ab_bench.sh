# A/B on ONE box: an earlier tree (_old/) vs the current tree, alternating, default bench without the CPU baseline.
# _old/ is scratch (git-excluded): git worktree add _old <commit> && (cd _old && python lossy-vae_amd/build_native.py)
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  (cd $R/_old && python bench.py --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('OLD', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])")
  (cd $R && python bench.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('NEW', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])")
done
(cd $R/_old && python bench.py --no-cpu-baseline --no-kernel-timing --batch 1 --steps 40 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('OLD b1', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])")
(cd $R && python bench.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --batch 1 --steps 40 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('NEW b1', j['value'], j['ms_per_step'], j['enc_ms_per_step'], j['dec_ms_per_step'])")
