#!/usr/bin/env python
"""Fixed-rate evaluation, same CLI and JSON schema as the reference's eval-fix-rate.py (:10-56): one model per lambda
(`get_model(name, lmb=..., pretrained=True)`), compress_mode() BEFORE .to(device) as the reference does (:30-31)."""
import argparse
import json
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
    ap.add_argument('-m', '--model', type=str, default='qres34m')
    ap.add_argument('-l', '--lambdas', type=int, default=[16, 32, 64, 128, 256, 512, 1024, 2048], nargs='+')
    ap.add_argument('-n', '--dataset_name', type=str, default='kodak')
    ap.add_argument('-d', '--device', type=str, default='cuda:0')
    args = ap.parse_args()

    save = Path(f'runs/results/{args.dataset_name}-{args.model}.json')
    save.parent.mkdir(parents=True, exist_ok=True)
    all_stats = {}
    for lmb in args.lambdas:
        model = get_model(args.model, lmb=lmb, pretrained=True)
        model.compress_mode()
        model = model.to(device=torch.device(args.device))
        model.eval()
        res = imcoding_evaluate(model, args.dataset_name, progress=True)
        print(f'lambda={lmb}: {res}')
        for k, v in res.items():
            all_stats.setdefault(k, []).append(v)
    out = {'name': args.model, 'test-set': args.dataset_name, 'platform': platform.platform(),
           'device': str(torch.cuda.get_device_properties(torch.device(args.device))), 'lambdas': args.lambdas, 'results': all_stats}
    with open(save, 'w') as f:
        json.dump(out, f, indent=2)
    print(f'saved to {save}')


if __name__ == '__main__':
    main()
