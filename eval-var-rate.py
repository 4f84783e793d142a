#!/usr/bin/env python
"""Variable-rate evaluation, same CLI and JSON schema as the reference's eval-var-rate.py (:10-66): one model, a sweep of
`steps` lambdas log-spaced over lmb_range, imcoding_evaluate per lambda, results dumped to runs/results/<set>-<model>.json."""
import argparse
import json
import math
import os
import platform
import sys
from pathlib import Path

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lossy-vae_amd'))
import torch  # noqa: E402
from lvae import get_model  # noqa: E402
from lvae.evaluation import imcoding_evaluate  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-m', '--model', type=str, default='qarv_base')
    ap.add_argument('-a', '--model_args', type=str, default='pretrained=True')
    ap.add_argument('-l', '--lmb_range', type=float, default=None, nargs='+')
    ap.add_argument('-s', '--steps', type=int, default=8)
    ap.add_argument('-n', '--dataset_name', type=str, default='kodak')
    ap.add_argument('-d', '--device', type=str, default='cuda:0')
    args = ap.parse_args()

    model = get_model(args.model, **eval(f'dict({args.model_args})'))
    model = model.to(device=torch.device(args.device))
    model.eval()
    model.compress_mode()
    start, end = args.lmb_range or model.lmb_range
    lambdas = torch.linspace(math.log(start), math.log(end), steps=args.steps).exp().tolist()

    save = Path(f'runs/results/{args.dataset_name}-{args.model}.json')
    save.parent.mkdir(parents=True, exist_ok=True)
    all_stats = {}
    for lmb in lambdas:
        if hasattr(model, 'default_lmb'):
            model.default_lmb = lmb
        res = imcoding_evaluate(model, args.dataset_name, progress=True)
        print(f'lambda={lmb:.2f}: {res}')
        for k, v in res.items():
            all_stats.setdefault(k, []).append(v)
    out = {'name': args.model, 'test-set': args.dataset_name, 'platform': platform.platform(),
           'device': str(torch.cuda.get_device_properties(torch.device(args.device))), 'lambdas': lambdas, 'results': all_stats}
    with open(save, 'w') as f:
        json.dump(out, f, indent=2)
    print(f'saved to {save}')


if __name__ == '__main__':
    main()
