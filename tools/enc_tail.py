"""Where does the host time of compress_batch go around the native group calls?  (bench workload, f16x2, 2 groups)
    entry -> first group's native call | native calls (launch / wait / coder seconds from lvae_encode_blocks) | last native return -> return"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device('cuda', 0)
model, sd = bench.build_model(dev)
model.coder_threads = max(8, len(os.sched_getaffinity(0)))
model.set_gemm_precision('f16x2')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ims = bench.synth_batch(B, 512, 768, 0).to(dev)
rec = []
orig = model._encode_group_native


def wrapped(pl, cuts, offs, n, tables, nthreads, stream, T=None):
    TT = {}
    t0 = time.perf_counter()
    r = orig(pl, cuts, offs, n, tables, nthreads, stream, TT)
    rec.append((t0, time.perf_counter(), TT))
    return r


model._encode_group_native = wrapped
for _ in range(5):
    s = model.compress_batch(ims)
    torch.cuda.synchronize()
acc = {'head': 0.0, 'native': 0.0, 'tail': 0.0, 'total': 0.0, 'launch': 0.0, 'wait': 0.0, 'coder': 0.0, 'sync_after': 0.0}
N = 30
for _ in range(N):
    rec.clear()
    tA = time.perf_counter()
    s = model.compress_batch(ims)
    tD = time.perf_counter()
    torch.cuda.synchronize()
    tE = time.perf_counter()
    acc['head'] += min(r[0] for r in rec) - tA
    acc['native'] += max(r[1] for r in rec) - min(r[0] for r in rec)
    acc['tail'] += tD - max(r[1] for r in rec)
    acc['total'] += tD - tA
    acc['sync_after'] += tE - tD
    for r in rec:
        acc['launch'] += r[2].get('enc_launch', 0) / len(rec)
        acc['wait'] += r[2].get('enc_gpu_wait', 0) / len(rec)
        acc['coder'] += r[2].get('enc_rans', 0) / len(rec)
print(f'B={B}: ' + '  '.join(f'{k} {v / N * 1e3:.3f} ms' for k, v in acc.items()))
