"""Round 6: decode time of the calibrated and the typical strings against the number of pipeline groups: python tools/r6_dec_groups.py [B]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import torch
from lvae import _native
if os.environ.get('LVAE_LIB'):            # experimental build (tools/build_exp.sh)
    _native.LIB_PATH = os.path.abspath(os.environ['LVAE_LIB'])
import bench, coder_workloads as cw
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
model, _ = bench.build_model(dev)
model.coder_threads = max(8, len(os.sched_getaffinity(0)))
ims = bench.synth_batch(B, 512, 768, 0).to(dev)
typ = model.compress_batch(ims)
cal, xhat, st, _ = cw.calibrated_strings(model, B, 8, 12, seed=1)
def t_dec(strings, n=15):
    for _ in range(3): model.decompress_batch(strings); torch.cuda.synchronize(dev)
    t0 = time.time()
    for _ in range(n): model.decompress_batch(strings); torch.cuda.synchronize(dev)
    return (time.time() - t0) / n * 1e3
for rep in range(2):
    for g in (1, 2, 3, 4, 8):
        if g > B: continue
        model.dec_groups = g
        print(f'dec_groups={g}: typical {t_dec(typ):.3f} ms   calibrated {t_dec(cal):.3f} ms', flush=True)
