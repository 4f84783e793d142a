"""Debug: fp8 mode, producer-side quantisation (Q8 pipeline) vs the in-GEMM quantiser: reconstruction distance to the fp32-class path."""
import os, sys, math
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import numpy as np, torch
import bench, seeded_init
from lvae import engine
m, _ = bench.build_model(torch.device('cuda:0'))
def psnr(a, b): return -10 * math.log10(float((a - b).square().mean()))
for (H, W, B) in [(512, 768, 2), (1216, 1216, 2)]:
    ims = torch.stack([torch.from_numpy(seeded_init.synthetic_image_u8(H, W, 900 + i)).permute(2, 0, 1).float().div(255) for i in range(B)]).cuda()
    for lmb in (512.0,):
        m.set_gemm_precision('f16x2')
        xr = m.decompress_batch(m.compress_batch(ims, lmb))
        zs, _ = m.get_latents(ims, lmb)
        for q8 in (False, True):
            engine.Plan.use_q8_pipeline = q8
            m._plans = {}
            m.set_gemm_precision('fp8')
            s = m.compress_batch(ims, lmb)
            x8 = m.decompress_batch(s)
            # decoder-only distance: feed the fp32-class latents to the fp8-mode decoder
            xd = m.conditional_sample(lmb, zs)
            print(H, W, 'q8 pipeline' if q8 else 'in-GEMM quantiser', f'recon-vs-recon {psnr(x8, xr):.2f} dB; decoder on identical latents {psnr(xd, xr):.2f} dB; bpp {np.mean([len(t) for t in s]) * 8 / (H * W):.4f}', flush=True)
