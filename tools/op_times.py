#!/usr/bin/env python
"""Per-launch timing of the qarv_base encode + decode plans (single stream, HIP events around every recorded launch), grouped by
(op, shape).  Shows where a step's GPU time goes and what each GEMM shape achieves.   python tools/op_times.py [B] [H] [W]
(OP_TIMES_PRECISION=f16x2|bf16x3|fp32|bf16|fp8 selects the GEMM arithmetic; default: the package default)"""
import ctypes
import os
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import torch  # noqa: E402

import bench  # noqa: E402
from lvae._native import GemmDesc  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 768
    dev = torch.device('cuda:0')
    name = os.environ.get('LVAE_MODEL', 'qarv_base')
    if name == 'qarv_base':
        model, _ = bench.build_model(dev)
    else:                                                # qres34m / qres17m / qres34m_lossless with seeded weights
        import lvae
        import seeded_init
        model = lvae.get_model(name)
        sd = model.state_dict()
        for k in list(sd.keys()):
            a = seeded_init.seeded_tensor(k, tuple(sd[k].shape), 0, profile='typical')
            if a is not None and 'discrete_gaussian' not in k:
                sd[k] = torch.from_numpy(a)
        model.load_state_dict(sd)
        model.compress_mode()
        model = model.to(dev).eval()
    if os.environ.get('OP_TIMES_PRECISION'):              # tool-side switch (the package itself has no environment override)
        model.set_gemm_precision(os.environ['OP_TIMES_PRECISION'])
    model.pipeline_groups = 1
    ims = bench.synth_batch(B, H, W, 0).to(dev)
    strings = model.compress_batch(ims)
    model.decompress_batch(strings)
    torch.cuda.synchronize()
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    reps = 3
    for key, pl in model._plans.items():
        if key[1] != B:
            continue
        s = torch.cuda.current_stream().cuda_stream
        sp = ctypes.c_void_p(s)
        for _ in range(reps):
            evs = []
            for fn, a, label, _side in pl.ops:
                if not callable(fn):
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn(*a, sp)
                e1.record()
                evs.append((e0, e1, fn, a, label))
            torch.cuda.synchronize()
            for e0, e1, fn, a, label in evs:
                name = getattr(fn, '__name__', str(fn))
                fl = 0.0
                if 'gemm' in name:
                    d = ctypes.cast(a[0], ctypes.POINTER(GemmDesc)).contents
                    k = (key[0], 'gemm', d.M, d.N, d.K, f'amode{d.a_mode} epi{d.epi} st{d.store}')
                    fl = 2.0 * d.M * d.N * d.K
                else:
                    if 'dwconv' in name:          # args: x, wt, bias, ln_w, ln_b, shift, scale1p, y, B, H, W, C, k
                        k = (key[0], 'dwconv_ln', f'{a[9]}x{a[10]}', int(a[11]), int(a[12]), '(H x W, C, k)')
                    else:
                        k = (key[0], name.replace('lvae_', ''), label.split('.')[-1], 0, 0, '')
                r = agg[k]
                r[0] += 1; r[1] += e0.elapsed_time(e1) * 1e3; r[2] += fl
    tot = sum(r[1] for r in agg.values()) / reps
    byplan = defaultdict(float)
    for k, r in agg.items():
        byplan[k[0]] += r[1] / reps
    print(f'total launch time per step (single stream, event-timed): {tot / 1e3:.2f} ms  ' + ', '.join(f'{k}: {v / 1e3:.2f} ms' for k, v in byplan.items()))
    print(f'{"plan":5s} {"op":14s} {"M":>7s} {"N":>5s} {"K":>5s} {"flags":18s} {"calls":>5s} {"us/call":>8s} {"ms/step":>8s} {"pct":>5s} {"TF/s":>6s}')
    top = int(os.environ.get('OP_TIMES_TOP', '45'))                # rows printed (the rest is summed in the last line)
    ranked = sorted(agg.items(), key=lambda kv: -kv[1][1])
    for k, r in ranked[:top]:
        calls = r[0] // reps
        us = r[1] / r[0]
        tf = r[2] / r[1] / 1e6 if r[2] else 0.0
        print(f'{k[0]:5s} {k[1]:14s} {str(k[2]):>7s} {k[3]:5d} {k[4]:5d} {k[5]:18s} {calls:5d} {us:8.1f} {r[1] / reps / 1e3:8.2f} {100 * r[1] / reps / tot:5.1f} {tf:6.1f}')
    if len(ranked) > top:
        rest = ranked[top:]
        print(f'(+ {len(rest)} more rows: {sum(r[0] for _, r in rest) // reps} calls, {sum(r[1] for _, r in rest) / reps / 1e3:.2f} ms/step)')


if __name__ == '__main__':
    main()
