#!/usr/bin/env python
"""Rate targeting: bisect lambda in log space until compress_file(..., lmb=) hits a byte budget.
Same CLI as the reference's scripts/qarv/test-at-target-bytes.py (:56-76)."""
import argparse
import math
import os
import sys
from pathlib import Path

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'lossy-vae_amd'))
import torch  # noqa: E402
import lvae  # noqa: E402
from lvae.utils.coding import pil_to_tensor01  # noqa: E402


def log_mid(a, b):
    return math.exp(0.5 * (math.log(a) + math.log(b)))


def binary_search_lmb(model, img_path, bits_path, tgt_bytes, max_iter=50, tol=1):
    from PIL import Image
    lo, hi = model.lmb_range
    lmb = log_mid(lo, hi)
    real = pil_to_tensor01(Image.open(img_path)).unsqueeze(0)
    for it in range(max_iter):
        model.compress_file(img_path, bits_path, lmb=lmb)
        n_bytes = Path(bits_path).stat().st_size
        fake = model.decompress_file(bits_path).cpu()
        psnr = -10 * math.log10(torch.mean((fake - real) ** 2).item())
        print(f'iter {it}: lmb={lmb:.3f}, bytes={n_bytes}B, target={tgt_bytes}B, '
              f'bpp={n_bytes * 8 / (real.shape[2] * real.shape[3]):.3f}, PSNR={psnr:.3f}')
        if abs(n_bytes - tgt_bytes) <= tol:
            break
        if n_bytes > tgt_bytes:
            hi = lmb
        else:
            lo = lmb
        lmb = log_mid(lo, hi)
    return lmb


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-i', '--input', type=str, default='runs/lake720p.jpg')
    ap.add_argument('-b', '--bits', type=str, default='runs/lake720p.bits')
    ap.add_argument('-m', '--model', type=str, default='qarv_base')
    ap.add_argument('-a', '--model_args', type=str, default='pretrained=True')
    ap.add_argument('-t', '--target_bytes', type=int, default=1500)
    ap.add_argument('--search_device', type=str, default='cuda:0')
    args = ap.parse_args()
    model = lvae.get_model(args.model, **eval(f'dict({args.model_args})'))
    model = model.to(device=torch.device(args.search_device))
    model.eval()
    model.compress_mode(True)
    lmb = binary_search_lmb(model, args.input, args.bits, args.target_bytes)
    print(f'lambda = {lmb:.4f}')


if __name__ == '__main__':
    main()
