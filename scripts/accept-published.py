#!/usr/bin/env python
"""Acceptance gate for the reference's PUBLISHED numbers (north_star: "identical bpp/PSNR on Kodak"): runs the drop-in evaluation
harness with the reference's trained checkpoints on a reference test set and compares every rate-distortion point with the published
one (tests/golden/published/published_rd.json <- /root/reference/results/<set>/<set>-<model>.json).

    python scripts/accept-published.py [-m qarv_base|qres34m] [-n kodak|clic2022-test|tecnick-rgb-1200]
                                       [--tol-bpp 0.005] [--tol-psnr 0.02] [--precision f16x2] [--config5]

Needs what this offline build does not have: the checkpoint(s) in torch.hub's cache ($TORCH_HOME/hub/checkpoints/
qarv_base-2022-dec-12.pt; qres34m-lmb{16..2048}.pt -- `get_model(..., pretrained=True)` resolves them without network when they are
there) and the image folder (lvae/paths.py: $LVAE_DATASETS/kodak, ...).  When either is absent the script prints `SKIP: <what is
missing>` and exits 0; otherwise it exits 1 on the first point outside tolerance:
    |bpp - published| <= tol_bpp * published   (default 0.5 %)      |PSNR - published| <= tol_psnr dB   (default 0.02)
qarv_base: the eval-var-rate.py sweep (:24-61) over the published lambdas; qres34m: eval-fix-rate.py's one-checkpoint-per-lambda loop
(:25-35, compress_mode() before .to()).  --config5 additionally reports, per lambda, the PSNR of the reduced-precision mode (bf16
storage + MX-fp8 GEMMs) against the image and against the fp32-class reconstruction -- the figure that decides whether config 5 is
usable at a trained model's 44 dB."""
import argparse
import json
import math
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
FIXTURE = os.path.join(REPO, 'tests', 'golden', 'published', 'published_rd.json')


def checkpoint_paths(model, lambdas):
    import torch
    d = os.path.join(torch.hub.get_dir(), 'checkpoints')
    if model == 'qarv_base':
        return [os.path.join(d, 'qarv_base-2022-dec-12.pt')]
    return [os.path.join(d, f'qres34m-lmb{int(l)}.pt') for l in lambdas]


def missing_inputs(model, dataset, case):
    from lvae.paths import known_datasets
    miss = [p for p in checkpoint_paths(model, case['lambdas']) if not os.path.isfile(p)]
    folder = known_datasets[dataset]
    if not folder.is_dir() or not any(folder.iterdir()):
        miss.append(str(folder))
    return miss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-m', '--model', default='qarv_base', choices=['qarv_base', 'qres34m'])
    ap.add_argument('-n', '--dataset_name', default='kodak', choices=['kodak', 'clic2022-test', 'tecnick-rgb-1200'])
    ap.add_argument('--tol-bpp', type=float, default=0.005)
    ap.add_argument('--tol-psnr', type=float, default=0.02)
    ap.add_argument('--precision', default=None, help='GEMM arithmetic (default: the package default)')
    ap.add_argument('--config5', action='store_true')
    ap.add_argument('-d', '--device', default='cuda:0')
    args = ap.parse_args()
    case = json.load(open(FIXTURE))['cases'][args.model][args.dataset_name]
    miss = missing_inputs(args.model, args.dataset_name, case)
    if miss:
        print('SKIP: not available offline: ' + ', '.join(miss))
        return 0
    import torch
    from lvae import get_model
    from lvae.evaluation import imcoding_evaluate
    dev = torch.device(args.device)
    bad = 0
    model = None
    for i, lmb in enumerate(case['lambdas']):
        if args.model == 'qarv_base':
            if model is None:
                model = get_model('qarv_base', pretrained=True).to(dev).eval()
                if args.precision:
                    model.set_gemm_precision(args.precision)
                model.compress_mode()
            model.default_lmb = lmb
        else:
            model = get_model('qres34m', lmb=int(lmb), pretrained=True)
            if args.precision:
                model.set_gemm_precision(args.precision)
            model.compress_mode()
            model = model.to(dev).eval()
        res = imcoding_evaluate(model, args.dataset_name)
        db, dp = res['bpp'] / case['bpp'][i] - 1, res['psnr'] - case['psnr'][i]
        ok = abs(db) <= args.tol_bpp and abs(dp) <= args.tol_psnr
        bad += not ok
        line = (f"lambda={lmb:9.3f}: bpp {res['bpp']:.6f} (published {case['bpp'][i]:.6f}, {100 * db:+.3f} %)  "
                f"PSNR {res['psnr']:.4f} dB (published {case['psnr'][i]:.4f}, {dp:+.4f})  {'ok' if ok else 'OUT OF TOLERANCE'}")
        if args.config5 and args.model == 'qarv_base':
            base = model._prec
            model.set_gemm_precision('fp8')
            r8 = imcoding_evaluate(model, args.dataset_name)
            model.set_gemm_precision(base)
            line += f"   config 5: bpp {r8['bpp']:.6f}, PSNR {r8['psnr']:.4f} dB ({r8['psnr'] - res['psnr']:+.4f} vs the fp32-class mode)"
        print(line, flush=True)
    print(f"{args.model} on {args.dataset_name}: {len(case['lambdas']) - bad} of {len(case['lambdas'])} points within "
          f"{100 * args.tol_bpp:g} % bpp / {args.tol_psnr:g} dB of {case['source']}")
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
