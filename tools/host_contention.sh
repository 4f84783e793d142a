# 8-rank HOST contention rehearsal on a 1-GPU box: N ranks, one process each, every rank pinned to its own 1/N of
# the host cores (what NUMA pinning gives it on the 8-GPU node) with a coder pool of that size, all sharing cuda:0 over gloo.
# The GPU is oversubscribed N-fold, so GPU waits are meaningless here; what is compared with the 1-rank run on the SAME core budget is
# the host side: launch-thread time (enc_launch), rANS encode / decode wall (enc_rans / dec_rans) per step.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3
mkdir -p $O
cd $R
N=${1:-8}
BATCH=${2:-1}
STEPS=10
CORES=$(python -c "import os; print(len(os.sched_getaffinity(0)))")
PER=$((CORES / N))
echo "# host cores available: $CORES; ranks: $N; cores per rank: $PER; batch $BATCH x 512x768 per rank; $STEPS steps + 3 warm-up"
echo "## 1 rank alone on $PER cores (taskset), coder pool $PER"
LVAE_TIMING=1 taskset -c 0-$((PER - 1)) python bench.py --batch $BATCH --steps $STEPS --no-cpu-baseline --no-kernel-timing --fp32-steps 0 2>&1 >/tmp/one.json | grep "host phase"
python -c "import json; j=json.load(open('/tmp/one.json')); print('   1 rank: ms_per_step', j['ms_per_step'], 'enc', j['enc_ms_per_step'], 'dec', j['dec_ms_per_step'])"
echo "## $N ranks, each on its own $PER cores, all on cuda:0 (gloo)"
LVAE_TIMING=1 LVAE_BENCH_SINGLE_GPU_TEST=1 LVAE_BENCH_REHEARSE_HOST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus $N --batch $BATCH --steps $STEPS --no-cpu-baseline --no-kernel-timing --fp32-steps 0 2>&1 >/tmp/many.json | grep "host phase" | sort
python -c "import json; j=[json.loads(l) for l in open('/tmp/many.json') if l.startswith('{')][0]; print('   $N ranks sharing ONE GPU: ms_per_step', j['ms_per_step'], '(GPU oversubscribed: not a scaling number)')"
