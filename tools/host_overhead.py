"""Host-side phase timers of compress_batch / decompress_batch (model.timing; bench workload): where the time before the first launch
and behind the last native call goes.   LVAE_TIMING=1 python tools/host_overhead.py [B=8] [steps=30]"""
import os
import sys
import time

os.environ.setdefault('LVAE_TIMING', '1')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device('cuda', 0)
model, sd = bench.build_model(dev)
model.coder_threads = max(8, len(os.sched_getaffinity(0)))
ims = bench.synth_batch(B, 512, 768, 0).to(dev)
for _ in range(4):
    s = model.compress_batch(ims); torch.cuda.synchronize(); model.decompress_batch(s); torch.cuda.synchronize()
model.timing = {}
te = td = 0.0
for _ in range(steps):
    t0 = time.perf_counter(); s = model.compress_batch(ims); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    model.decompress_batch(s); t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    te += t2 - t0; td += t4 - t2
T = model.timing
ng = len(model._groups(B, 'dec'))
print(f'B={B}: enc {te / steps * 1e3:.3f} ms, dec {td / steps * 1e3:.3f} ms per step ({ng} group(s); per-group timers are means over the groups)')
for k in sorted(T):
    per = T[k] / steps
    if k not in ('dec_head_parse', 'dec_head_setup', 'dec_groups_total', 'dec_calls'):
        per /= ng
    print(f'   {k:18s} {per * 1e3:8.3f} ms')
