"""Round 6: decode pipeline groups started with an offset (model.group_stagger_s, study knob): with equal starts the groups run in phase (all
on the GPU, then all in the host coder); on calibrated streams the host phases are the longer ones, so offset groups could use the GPU
while the others decode.   python tools/r6_stagger_groups.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import numpy as np
import torch
import bench, coder_workloads as cw
B = 8
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
model, _ = bench.build_model(dev)
model.coder_threads = max(8, len(os.sched_getaffinity(0)))
ims = bench.synth_batch(B, 512, 768, 0).to(dev)
typ = model.compress_batch(ims)
cal = cw.calibrated_strings(model, B, 8, 12, seed=1)[0]
def t_dec(strings, n=15):
    for _ in range(3): model.decompress_batch(strings); torch.cuda.synchronize(dev)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); model.decompress_batch(strings); torch.cuda.synchronize(dev); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3
for rep in range(2):
    for g in (2, 4, 8):
        for stg in (0, 100, 200, 300, 450):
            model.dec_groups, model.group_stagger_s = g, stg * 1e-6
            print(f'groups {g} stagger {stg:3d} us: typical {t_dec(typ):.3f} ms   calibrated {t_dec(cal):.3f} ms', flush=True)
