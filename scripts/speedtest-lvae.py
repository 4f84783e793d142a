#!/usr/bin/env python
"""Latency protocol of the reference's scripts/speedtest-lvae.py (:13-44) on the MI355X build: per image, tensor already on
the device, time compress() and decompress() separately with a device sync after each, 4-image warm-up, mean over images.
Same CLI (-m/--models, -a/--kwargs, -d/--device, -w/--workers); `--synthetic N` uses N seeded 512x768 images and seeded
weights when Kodak / checkpoints are not on disk."""
import argparse
import os
import sys
from time import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lossy-vae_amd'))
import torch  # noqa: E402
from lvae.models.registry import get_model  # noqa: E402
from lvae.paths import known_datasets  # noqa: E402
from lvae.utils.coding import pil_to_tensor01  # noqa: E402


def load_images(synthetic):
    if synthetic:
        import numpy as np
        import seeded_init
        return [torch.from_numpy(seeded_init.synthetic_image_u8(512, 768, seed=i)).permute(2, 0, 1).float().div(255)
                for i in range(synthetic)]
    from PIL import Image
    return [pil_to_tensor01(Image.open(p)) for p in sorted(known_datasets['kodak'].rglob('*.*'))]


def speedtest(model, images, first=None):
    device = next(model.parameters()).device
    images = images[:first] if first else images
    enc = dec = 0.0
    for im in images:
        im = im.unsqueeze(0).to(device=device)
        t0 = time()
        obj = model.compress(im)
        torch.cuda.synchronize()
        t1 = time()
        model.decompress(obj)
        torch.cuda.synchronize()
        t2 = time()
        enc += t1 - t0
        dec += t2 - t1
    return enc / len(images), dec / len(images)


@torch.no_grad()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('-m', '--models', type=str, default=['qarv_base'], nargs='+')
    ap.add_argument('-a', '--kwargs', type=str, default='pretrained=True')
    ap.add_argument('-d', '--device', type=str, default='cuda:0')
    ap.add_argument('-w', '--workers', type=int, default=None, help='host coder threads')
    ap.add_argument('--synthetic', type=int, default=0)
    args = ap.parse_args()
    print(f'pytorch = {torch.__version__}, hip = {torch.version.hip}')
    device = torch.device(args.device)
    print(f'device = {torch.cuda.get_device_properties(device)}')
    images = load_images(args.synthetic)
    for name in args.models:
        kwargs = eval(f'dict({args.kwargs})')
        if args.synthetic and kwargs.get('pretrained') is True:
            kwargs['pretrained'] = False
        model = get_model(name, **kwargs)
        if args.synthetic:
            import seeded_init
            sd = model.state_dict()
            for k in sd:
                a = seeded_init.seeded_tensor(k, tuple(sd[k].shape), 0)
                if a is not None:
                    sd[k] = torch.from_numpy(a)
            model.load_state_dict(sd)
        n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
        model = model.to(device=device)
        model.eval()
        model.compress_mode()
        if args.workers is not None:
            model.coder_threads = args.workers
        print(f'{name}, {type(model)}, device={device}\nNumber of parameters: {n_params / 1e6:.3f} M')
        speedtest(model, images, first=4)
        enc_time, dec_time = speedtest(model, images)
        print(f'encode time={enc_time:.3f}s, decode time={dec_time:.3f}s')


if __name__ == '__main__':
    main()
