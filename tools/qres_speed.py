#!/usr/bin/env python
"""Throughput of the qres models (seeded weights, synthetic images):  python tools/qres_speed.py [model] [B] [H] [W]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import torch, lvae, seeded_init

name = sys.argv[1] if len(sys.argv) > 1 else 'qres34m'
B, H, W = (int(v) for v in (sys.argv[2:5] + ['8', '512', '768'][len(sys.argv[2:5]):]))
m = lvae.get_model(name)
sd = m.state_dict()
for k in list(sd.keys()):
    a = seeded_init.seeded_tensor(k, tuple(sd[k].shape), 0, profile='typical')
    if a is not None and 'discrete_gaussian' not in k:
        sd[k] = torch.from_numpy(a)
m.load_state_dict(sd)
m.compress_mode()
m = m.cuda().eval()
ims = torch.cat([torch.from_numpy(seeded_init.synthetic_image_u8(H, W, 900 + i)).permute(2, 0, 1).float().div(255).unsqueeze(0) for i in range(B)]).cuda()
for _ in range(3):
    objs = m.compress_batch(ims); x = m.decompress_batch(objs)
torch.cuda.synchronize()
t0 = time.time(); n = 10
for _ in range(n):
    objs = m.compress_batch(ims); torch.cuda.synchronize(); t1 = time.time(); x = m.decompress_batch(objs); torch.cuda.synchronize()
dt = (time.time() - t0) / n
nbytes = sum(sum(len(s[0]) for s in o if isinstance(s, list)) for o in objs)
print(f'{name} B={B} {H}x{W}: {dt * 1e3:.1f} ms/step enc+dec, {B * H * W / dt / 1e6:.1f} Mpixels/s, bpp {nbytes * 8 / (B * H * W):.3f}, '
      f'psnr {float(-10 * torch.log10((x - ims).square().mean())):.2f} dB')
