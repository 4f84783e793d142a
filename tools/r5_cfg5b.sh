R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_cfg5b
mkdir -p $O
cd $R
for i in 1 2; do
python bench.py --no-cpu-baseline --fp32-steps 0 --b1-steps 0 --qres-steps 0 --steps 5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); c=j['config5']; print('default flow :', j['value'], c['value'], c['enc_ms_per_step'], c['dec_ms_per_step'], c['speedup_vs_fp32_class'])" | tee -a $O/out.txt
python bench.py --precision fp8 --no-cpu-baseline --batch 4 --height 1216 --width 1216 --steps 8 --no-kernel-timing 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('standalone   :', j['value'], j['enc_ms_per_step'], j['dec_ms_per_step'])" | tee -a $O/out.txt
python bench.py --no-cpu-baseline --fp32-steps 0 --b1-steps 0 --qres-steps 0 --steps 5 --no-kernel-timing 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); c=j['config5']; print('default, no roofline pass:', j['value'], c['value'], c['enc_ms_per_step'], c['dec_ms_per_step'], c['speedup_vs_fp32_class'])" | tee -a $O/out.txt
done
