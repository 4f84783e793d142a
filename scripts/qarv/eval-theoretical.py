#!/usr/bin/env python
"""Estimated-rate evaluation (no entropy coding): model.self_evaluate over datasets, same CLI as the reference's
scripts/qarv/eval-theoretical.py (:8-31)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'lossy-vae_amd'))
import torch  # noqa: E402
import lvae  # noqa: E402


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-m', '--model', type=str, default='qarv_base')
    ap.add_argument('-a', '--model_args', type=str, default='pretrained=True')
    ap.add_argument('-l', '--lmb_range', type=float, default=[16, 2048], nargs='+')
    ap.add_argument('-s', '--steps', type=int, default=8)
    ap.add_argument('-n', '--datasets', type=str, default=['kodak', 'tecnick-rgb-1200', 'clic2022-test'], nargs='+')
    ap.add_argument('-d', '--device', type=str, default='cuda:0')
    args = ap.parse_args()
    model = lvae.get_model(args.model, **eval(f'dict({args.model_args})'))
    model = model.to(device=torch.device(args.device))
    model.eval()
    for name in args.datasets:
        img_dir = lvae.paths.known_datasets.get(name, name)
        stats = model.self_evaluate(img_dir, lmb_range=args.lmb_range, steps=args.steps)
        print(f'================ {name} ================')
        for k, vlist in stats.items():
            print(f'{k:<6s} = [' + ', '.join(f'{v:.12f}'[:7] for v in vlist) + ']')


if __name__ == '__main__':
    main()
