"""Round 6: do stream PRIORITIES change the two-group pipeline?  With equal priorities the two groups' GPU segments run concurrently and the groups
stay in phase (both wait for the GPU, then both decode on the host); with group 0's stream at high priority its segments should go first and the
groups fall into anti-phase.  Typical and calibrated strings, batch 8:   python tools/r6_prio.py"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import torch
import bench, coder_workloads as cw
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
model, _ = bench.build_model(dev)
model.coder_threads = max(8, len(os.sched_getaffinity(0)))
ims = bench.synth_batch(B, 512, 768, 0).to(dev)
typ = model.compress_batch(ims)
cal, xhat, st, _ = cw.calibrated_strings(model, B, 8, 12, seed=1)
def t(fn, n=15):
    for _ in range(3): fn(); torch.cuda.synchronize(dev)
    ts = []
    for _ in range(n):
        t0 = time.time(); fn(); torch.cuda.synchronize(dev); ts.append(time.time() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else 'n/a')
for rep in range(2):
    for pr in ((0, 0), (-1, 0), (0, -1)):
        model._streams = [torch.cuda.Stream(device=dev, priority=p) for p in pr]
        print(f'priorities {pr}: enc {t(lambda: model.compress_batch(ims)):.3f}  dec typical {t(lambda: model.decompress_batch(typ)):.3f}  '
              f'dec calibrated {t(lambda: model.decompress_batch(cal)):.3f} ms (median of 15)', flush=True)
