"""Host-side timeline of decompress_batch (bench workload, product configuration), unprofiled: per pipeline group and latent block the
stamps lvae_decode_blocks records (segment launch begins / issued / indexes on the host / block decoded), relative to the call's entry,
averaged over the steps.   python tools/dec_timeline.py [B=8] [steps=20] [typical|calibrated]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

if os.environ.get('LVAE_LIB'):            # A/B against another build of the native library (tools/r5_rans_mps.sh)
    from lvae import _native
    _native.LIB_PATH = os.path.abspath(os.environ['LVAE_LIB'])

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda', 0)
model, sd = bench.build_model(dev)
model.coder_threads = max(8, len(os.sched_getaffinity(0)))
if os.environ.get('DEC_GROUPS'):          # number of pipeline groups (default: the product's rule)
    model.dec_groups = int(os.environ['DEC_GROUPS'])
ims = bench.synth_batch(B, 512, 768, 0).to(dev)
kind = sys.argv[3] if len(sys.argv) > 3 else 'typical'
if kind == 'calibrated':           # latents drawn from the model's own prior (lossy-vae_amd/coder_workloads.py)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'lossy-vae_amd'))
    import coder_workloads as cw  # noqa: E402
    s = cw.calibrated_strings(model, B, 8, 12, seed=1)[0]
else:
    s = model.compress_batch(ims)
for _ in range(4):
    torch.cuda.synchronize(); model.decompress_batch(s); torch.cuda.synchronize()
rows = []
tot = 0.0
for _ in range(steps):
    model.dec_trace = []
    t0 = time.monotonic()
    model.decompress_batch(s)
    t_ret = time.monotonic()
    torch.cuda.synchronize()
    t1 = time.monotonic()
    tot += t1 - t0
    tr = sorted(model.dec_trace, key=lambda r: r[2][2])          # groups by their first stamp
    rows.append((t_ret - t0, t1 - t0, [[v - t0 for v in r[2][2:2 + 4 * r[1] + 1]] for r in tr]))
model.dec_trace = None
print(f'B={B} ({kind} strings): decompress_batch + sync {tot / steps * 1e3:.3f} ms per step; returns at {np.mean([r[0] for r in rows]) * 1e3:.3f} ms')
ng = len(rows[0][2])
for g in range(ng):
    a = np.mean([r[2][g] for r in rows], axis=0) * 1e3
    nb = (len(a) - 1) // 4
    print(f'group {g} (mean over {steps} steps, ms from entry):  block: launch-begin  issued  idx-on-host  decoded | gpu-wait  rans')
    for b in range(nb):
        t0_, ti, t1_, t2_ = a[4 * b:4 * b + 4]
        print(f'   b{b}: {t0_:7.3f} {ti:7.3f} {t1_:7.3f} {t2_:7.3f} | {t1_ - t0_:6.3f} {t2_ - t1_:6.3f}')
    print(f'   tail issued {a[4 * nb]:7.3f}')
