#!/usr/bin/env python
"""Progressive decoding (the reference's scripts/qarv/robust-decoding.py on this package): encode one image, then decode it
from its first k+1 latent blocks only, the remaining blocks set to their prior means (conditional_sample, t = 0); prints the
cumulative bpp of each prefix and saves the decodings side by side.

    python scripts/qarv/robust-decoding.py [--image PATH | --synthetic H W] [--lmb 16] [--mode progressive|exclude|reverse|single]
"""
import argparse
import math
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

import lvae  # noqa: E402
import seeded_init  # noqa: E402
from lvae.utils import coding  # noqa: E402


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--image', type=str, default=None)
    ap.add_argument('--synthetic', type=int, nargs=2, default=[256, 256], metavar=('H', 'W'))
    ap.add_argument('--lmb', type=float, default=16)
    ap.add_argument('--mode', type=str, default='progressive', choices=['progressive', 'exclude', 'reverse', 'single'])
    ap.add_argument('--weights', type=str, default=None, help='checkpoint (state dict with the reference key names); default: seeded random init')
    ap.add_argument('--out', type=str, default=None)
    args = ap.parse_args()
    device = torch.device('cuda:0')
    model = lvae.get_model('qarv_base', pretrained=args.weights if args.weights else False)
    if not args.weights:                                         # no network for checkpoints: seeded random init
        sd = model.state_dict()
        for k in list(sd.keys()):
            a = seeded_init.seeded_tensor(k, tuple(sd[k].shape), 0, profile='typical')
            if a is not None:
                sd[k] = torch.from_numpy(a)
        model.load_state_dict(sd)
    model = model.to(device).eval()
    if args.image:
        img = coding.pad_divisible_by(Image.open(args.image).convert('RGB'), div=model.max_stride)
        im = coding.pil_to_tensor01(img).unsqueeze(0).to(device)
        stem = os.path.splitext(os.path.basename(args.image))[0]
    else:
        h, w = args.synthetic
        u8 = seeded_init.synthetic_image_u8(h, w, 0)
        im = torch.from_numpy(u8).permute(2, 0, 1).float().div(255).unsqueeze(0).to(device)
        stem = f'synthetic{h}x{w}'
    nB, _, imH, imW = im.shape
    zs, nats = model.get_latents(im, args.lmb)                   # forward_end2end(..., get_latent=True) of the reference
    L = len(zs)
    outs, bpps = [], []
    for anchor in range(L):
        keep = {'progressive': lambda i: i <= anchor, 'exclude': lambda i: i != anchor, 'reverse': lambda i: i >= anchor,
                'single': lambda i: i == anchor}[args.mode]
        latents = [z if keep(i) else None for i, z in enumerate(zs)]
        x = model.conditional_sample(args.lmb, latents, bhw_repeat=(nB, imH // 64, imW // 64), t=0)
        bpp = sum(float(nats[i, 0]) for i in range(L) if keep(i)) / (imH * imW) * math.log2(math.e)
        psnr = -10 * math.log10(float((x - im).square().mean()))
        outs.append(x[0].clamp(0, 1).cpu())
        bpps.append(bpp)
        print(f'{args.mode}={anchor}, bpp={bpp:.4f}, psnr={psnr:.2f} dB')
    grid = torch.cat(outs, dim=2)                                 # side by side
    arr = (grid.permute(1, 2, 0).numpy() * 255).round().astype(np.uint8)
    out = args.out or os.path.join('runs', f'qarv-{args.mode}-lmb{int(args.lmb)}-{stem}.png')
    os.makedirs(os.path.dirname(out) or '.', exist_ok=True)
    Image.fromarray(arr).save(out)
    print(', '.join(f'{b:.3f} bpp' for b in bpps))
    print(out)


if __name__ == '__main__':
    main()
