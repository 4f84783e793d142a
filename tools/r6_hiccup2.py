"""Round 6: slow decode steps -- with the per-block trace on (tools/r6_hiccup.py saw none in 300 steps) and off (bench.py's loop sees one or two in
40), alternating blocks of 100 steps in one process.   python tools/r6_hiccup2.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
import bench
dev = torch.device('cuda', 0)
model, sd = bench.build_model(dev)
model.coder_threads = max(8, len(os.sched_getaffinity(0)))
ims = bench.synth_batch(8, 512, 768, 0).to(dev)
for _ in range(5):
    s = model.compress_batch(ims); torch.cuda.synchronize(); model.decompress_batch(s); torch.cuda.synchronize()
for rep in range(3):
    for trace in (False, True):
        enc, dec = [], []
        for i in range(100):
            t0 = time.time(); s = model.compress_batch(ims); torch.cuda.synchronize(dev); t1 = time.time()
            if trace: model.dec_trace = []
            out = model.decompress_batch(s); torch.cuda.synchronize(dev); t2 = time.time()
            enc.append(t1 - t0); dec.append(t2 - t1)
        model.dec_trace = None
        enc, dec = np.array(enc) * 1e3, np.array(dec) * 1e3
        print(f'trace {"on " if trace else "off"}: decode median {np.median(dec):.3f} mean {dec.mean():.3f} max {dec.max():.3f}, steps > 1.3 x median: {int((dec > 1.3 * np.median(dec)).sum())} | '
              f'encode median {np.median(enc):.3f} mean {enc.mean():.3f} max {enc.max():.3f}, > 1.1 x median: {int((enc > 1.1 * np.median(enc)).sum())}', flush=True)
