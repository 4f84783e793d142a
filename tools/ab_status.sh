# same-box A/B: what the status word (non-finite guard) costs -- alternating bench runs with and without it
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab_status
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  for v in 0 1; do
    LVAE_NO_STATUS_CHECK=$v python $R/bench.py --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 --b1-steps 0 --qres-steps 0 --steps 30 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nostatus=$v', d['value'], d['enc_ms_per_step'], d['dec_ms_per_step'])" | tee -a $O/ab.txt
  done
done
for v in 0 1; do
LVAE_NO_STATUS_CHECK=$v python $R/bench.py --batch 1 --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing --fp32-steps 0 --config5-steps 0 --b1-steps 0 --qres-steps 0 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b1 nostatus=$v', d['ms_per_step'], d['enc_ms_per_step'], d['dec_ms_per_step'])" | tee -a $O/ab.txt
done
$R/tools/ubench/mfma_valu_interleave > $O/mfma_valu_interleave.txt 2>&1; cat $O/mfma_valu_interleave.txt
