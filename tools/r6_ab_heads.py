"""Round 6: same-process A/B of the deferred head reduces (prior / posterior split-K planes summed by lvae_prior_index_sk_f32 /
lvae_quantize_sk_f32 instead of a reduce launch): two models, timed in alternation, medians.   python tools/r6_ab_heads.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
import bench
from lvae import engine
dev = torch.device('cuda', 0)
models = {}
for name, on in (('deferred', True), ('reduce launch', False)):
    engine.Plan.DEFER_HEAD_REDUCE = on
    m, _ = bench.build_model(dev)
    m.coder_threads = max(8, len(os.sched_getaffinity(0)))
    models[name] = m
    for B in (8, 1):
        ims = bench.synth_batch(B, 512, 768, 0).to(dev)
        for _ in range(3):
            s = m.compress_batch(ims); torch.cuda.synchronize(); m.decompress_batch(s); torch.cuda.synchronize()
        models[(name, B)] = (ims, s)
engine.Plan.DEFER_HEAD_REDUCE = True
assert all(models[('deferred', B)][1] == models[('reduce launch', B)][1] for B in (8, 1)), 'the two forms must write the same bytes'
def t(fn, n):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3
res = {}
for r in range(6):
    for name in ('deferred', 'reduce launch'):
        m = models[name]
        for B in (8, 1):
            ims, s = models[(name, B)]
            res.setdefault((name, B, 'enc'), []).append(t(lambda: m.compress_batch(ims), 12))
            res.setdefault((name, B, 'dec'), []).append(t(lambda: m.decompress_batch(s), 12))
for B in (8, 1):
    for what in ('enc', 'dec'):
        a, b = res[('deferred', B, what)], res[('reduce launch', B, what)]
        print(f'B={B} {what}: deferred {np.median(a):.3f} ms (rounds {" ".join(f"{v:.3f}" for v in a)}) | reduce launch {np.median(b):.3f} ms ({" ".join(f"{v:.3f}" for v in b)})')
