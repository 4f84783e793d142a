R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_host
mkdir -p $O
python $R/tools/enc_tail.py 8 2>&1 | grep -v amdgpu | tee $O/enc_tail_b8.txt
python $R/tools/enc_tail.py 1 2>&1 | grep -v amdgpu | tee $O/enc_tail_b1.txt
LVAE_TIMING=1 python - <<'PY' 2>&1 | grep -v amdgpu | tee $O/timing.txt
import os, sys, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch, bench
dev = torch.device('cuda', 0)
model, sd = bench.build_model(dev)
model.coder_threads = max(8, len(os.sched_getaffinity(0)))
ims = bench.synth_batch(8, 512, 768, 0).to(dev)
for _ in range(5):
    s = model.compress_batch(ims); torch.cuda.synchronize(); o = model.decompress_batch(s); torch.cuda.synchronize()
model.timing.clear()
N = 30
te = td = 0
for _ in range(N):
    t0 = time.perf_counter(); s = model.compress_batch(ims); torch.cuda.synchronize(); t1 = time.perf_counter()
    o = model.decompress_batch(s); torch.cuda.synchronize(); t2 = time.perf_counter()
    te += t1 - t0; td += t2 - t1
print(f'enc {te/N*1e3:.3f} ms dec {td/N*1e3:.3f} ms per step')
for k, v in sorted(model.timing.items()):
    print(f'{k:20s} {v / N * 1e3:9.3f} ms per step (sum over groups)')
PY
