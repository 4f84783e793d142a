#!/usr/bin/env python
"""Determinism soak of the product path on one MI355X: for `--seconds` of wall time, keep encoding and decoding the same seeded
inputs and compare every result with the first one, bit for bit --

  * compress_batch with model.side_streams = False (two pipeline groups, two streams; the reference strings come from this form)
                                                            -> byte strings == first run's
  * compress_batch with the side stream (the product default since round 5: two groups x (main + side stream), posterior0 of the
    stride-8 / 16 blocks hoisted)                           -> == the same strings
  * single-image compress of one image of the batch        -> == that image's string of the batch call (batch invariance)
  * single-image compress with model.side_streams = True   -> == the same string (fork/join plan)
  * decompress_batch                                        -> reconstruction bits == first run's
  * decompress of one string alone                          -> == that image of the batch decode

at two image sizes (512x768 and a ragged 320x448), batch 8.  One JSON line with the iteration counts and the number of mismatches
(`tools/soak.py --seconds 240 > gpurun_out/soak.json`).  Nothing here touches oracle/ or the reference: it is a property of the
product path alone."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'lossy-vae_amd'))

import bench  # noqa: E402  (build_model / synth_batch: the bench's own seeded model and images)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seconds', type=float, default=120.0)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--model', type=str, default='qarv_base', help='qarv_base (the bench model) or a qres model name (seeded weights)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    if args.model == 'qarv_base':
        model, _ = bench.build_model(dev)
    else:
        import lvae
        import seeded_init
        model = lvae.get_model(args.model)
        sd = model.state_dict()
        for k in list(sd.keys()):
            a = seeded_init.seeded_tensor(k, tuple(sd[k].shape), 0, profile='typical')
            if a is not None and 'discrete_gaussian' not in k:
                sd[k] = torch.from_numpy(a)
        model.load_state_dict(sd)
        model.compress_mode()
        model = model.to(dev).eval()
    cases = []
    for (H, W) in ((512, 768), (320, 448)):
        x = bench.synth_batch(args.batch, H, W, 0).to(dev)
        model.side_streams = False
        ref_strings = model.compress_batch(x)
        ref_rec = model.decompress_batch(ref_strings).clone()
        cases.append((H, W, x, ref_strings, ref_rec))
    counts = {'compress_batch': 0, 'compress_batch_side_streams': 0, 'compress_single': 0, 'compress_single_side_streams': 0, 'decompress_batch': 0, 'decompress_single': 0}
    bad = {k: 0 for k in counts}
    t0 = time.time()
    it = 0
    while time.time() - t0 < args.seconds:
        H, W, x, ref_strings, ref_rec = cases[it % len(cases)]
        i = it % args.batch
        model.side_streams = False
        s = model.compress_batch(x)
        counts['compress_batch'] += 1
        bad['compress_batch'] += int(s != ref_strings)
        model.side_streams = True
        sb = model.compress_batch(x)
        model.side_streams = False
        counts['compress_batch_side_streams'] += 1
        bad['compress_batch_side_streams'] += int(sb != ref_strings)
        s1 = model.compress(x[i:i + 1])
        counts['compress_single'] += 1
        bad['compress_single'] += int(s1 != ref_strings[i])
        model.side_streams = True
        s2 = model.compress(x[i:i + 1])
        model.side_streams = False
        counts['compress_single_side_streams'] += 1
        bad['compress_single_side_streams'] += int(s2 != ref_strings[i])
        r = model.decompress_batch(ref_strings)
        counts['decompress_batch'] += 1
        bad['decompress_batch'] += int(not torch.equal(r, ref_rec))
        r1 = model.decompress(ref_strings[i])
        counts['decompress_single'] += 1
        bad['decompress_single'] += int(not torch.equal(r1[0], ref_rec[i]))
        it += 1
    torch.cuda.synchronize()
    print(json.dumps({'model': args.model, 'seconds': round(time.time() - t0, 1), 'batch': args.batch, 'sizes': [[c[0], c[1]] for c in cases],
                      'precision': getattr(model, '_prec', None), 'iterations': it, 'calls': counts, 'mismatches': bad,
                      'all_identical': all(v == 0 for v in bad.values())}))
    return 0 if all(v == 0 for v in bad.values()) else 1


if __name__ == '__main__':
    sys.exit(main())
