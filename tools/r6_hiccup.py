"""Round 6: where do the occasional slow decode steps of the bench loop (5.6 ms median, 9-13 ms once or twice in 40 steps) lose their time?
The bench's own step (compress_batch + sync + decompress_batch + sync) with the decode's per-block stamps recorded every step; prints the
median step's timeline and every step slower than 1.3 x the median next to it.   python tools/r6_hiccup.py [steps=300]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device('cuda', 0)
model, sd = bench.build_model(dev)
model.coder_threads = max(8, len(os.sched_getaffinity(0)))
ims = bench.synth_batch(8, 512, 768, 0).to(dev)
for _ in range(5):
    s = model.compress_batch(ims); torch.cuda.synchronize(); model.decompress_batch(s); torch.cuda.synchronize()
rows = []
for i in range(steps):
    te0 = time.monotonic()
    s = model.compress_batch(ims); torch.cuda.synchronize()
    te1 = time.monotonic()
    model.dec_trace = []
    t0 = time.monotonic()
    model.decompress_batch(s)
    t_ret = time.monotonic()
    torch.cuda.synchronize()
    t1 = time.monotonic()
    tr = sorted(model.dec_trace, key=lambda r: r[2][2])
    rows.append((te1 - te0, t1 - t0, t_ret - t0, [np.array([v - t0 for v in r[2][2:2 + 4 * r[1] + 1]]) * 1e3 for r in tr]))
model.dec_trace = None
dec = np.array([r[1] for r in rows]) * 1e3
enc = np.array([r[0] for r in rows]) * 1e3
med = float(np.median(dec))
print(f'{steps} steps: decode median {med:.3f} ms, mean {dec.mean():.3f}, max {dec.max():.3f}; encode median {np.median(enc):.3f}, mean {enc.mean():.3f}, max {enc.max():.3f}')
print(f'decode steps > 1.3 x median: {int((dec > 1.3 * med).sum())}; encode steps > 1.15 x median: {int((enc > 1.15 * np.median(enc)).sum())}')
imed = int(np.argsort(dec)[len(dec) // 2])
def show(i):
    r = rows[i]
    print(f'step {i}: decode {r[1] * 1e3:.3f} ms (returns at {r[2] * 1e3:.3f})')
    for g, a in enumerate(r[3]):
        nb = (len(a) - 1) // 4
        gw = ' '.join(f'{a[4 * b + 2] - a[4 * b]:.2f}' for b in range(nb))
        rn = ' '.join(f'{a[4 * b + 3] - a[4 * b + 2]:.2f}' for b in range(nb))
        print(f'   group {g}: first launch at {a[0]:.3f}; gpu-wait per block [{gw}]; rans per block [{rn}]; tail issued at {a[4 * nb]:.3f}')
print('-- median step'); show(imed)
print('-- slow steps')
for i in np.argsort(-dec)[:6]:
    if dec[i] > 1.3 * med: show(int(i))
