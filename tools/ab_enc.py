"""Same-process A/B of encode configurations on the bench workload: every variant is its own model (own plans and streams); the variants
are timed in alternation (rounds x steps of compress_batch + sync each), medians over the rounds.
    python tools/ab_enc.py "groups,side" "groups,side" ...      e.g.  "2,1" "2,0" "1,1" """
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device('cuda', 0)
specs = sys.argv[1:] or ['2,1', '2,0']
B = int(os.environ.get('AB_BATCH', '8'))
ims = bench.synth_batch(B, 512, 768, 0).to(dev)
models = []
for sp in specs:
    g, side = (int(v) for v in sp.split(',')[:2])
    fused384 = int(sp.split(',')[2]) if len(sp.split(',')) > 2 and sp.split(',')[2] != '' else None      # third field: row threshold of the (384, 768) fused MLP while this variant's plans are built
    m, _ = bench.build_model(dev)
    m.coder_threads = max(8, len(os.sched_getaffinity(0)))
    m.enc_groups, m.side_streams = g, bool(side)
    if len(sp.split(',')) > 3:                              # fourth field: 0 = small plans (<= 2 x 512x768 pixels) do not hoist posterior0
        m.hoist_small = bool(int(sp.split(',')[3]))
    from lvae import engine
    saved_rows = dict(engine.Plan.FUSED_MLP_MIN_ROWS)
    if fused384 is not None:
        engine.Plan.FUSED_MLP_MIN_ROWS = {(384, 768): fused384}
    for _ in range(4):
        s = m.compress_batch(ims)
        torch.cuda.synchronize()
    engine.Plan.FUSED_MLP_MIN_ROWS = saved_rows
    models.append((sp, m, s))
assert all(s == models[0][2] for _, _, s in models), 'variants must produce the same bytes'
rounds, steps = int(os.environ.get('AB_ROUNDS', '8')), int(os.environ.get('AB_STEPS', '15'))
res = {sp: [] for sp, _, _ in models}
for r in range(rounds):
    for sp, m, _ in models:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m.compress_batch(ims)
            torch.cuda.synchronize()
        res[sp].append((time.perf_counter() - t0) / steps * 1e3)
for sp in res:
    a = np.array(res[sp])
    print(f'enc groups,side = {sp}: median {np.median(a):.3f} ms  min {a.min():.3f}  max {a.max():.3f}  (B={B}, {rounds} rounds x {steps} steps)')
