"""Decode-only / encode-only loops of the bench workload for a rocprofv3 kernel trace (tools/dec_trace.sh):
    dec_trace.py dec|enc [steps=10] [batch=8]
prints the phase's wall time per step; the trace then says how busy the GPU is inside that phase alone."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402

what = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device('cuda', 0)
model, sd = bench.build_model(dev)
model.coder_threads = max(8, len(os.sched_getaffinity(0)))
model.set_gemm_precision('f16x2')
ims = bench.synth_batch(B, 512, 768, 0).to(dev)
for _ in range(3):
    strings = model.compress_batch(ims)
    out = model.decompress_batch(strings)
torch.cuda.synchronize(dev)
time.sleep(0.05)                                   # a visible gap in the trace before the measured phase
t0 = time.time()
for _ in range(steps):
    if what == 'dec':
        out = model.decompress_batch(strings)
    else:
        strings = model.compress_batch(ims)
    torch.cuda.synchronize(dev)
print(f'{what}: {(time.time() - t0) / steps * 1e3:.3f} ms per step (B={B})')
