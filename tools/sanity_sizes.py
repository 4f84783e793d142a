import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/lossy-vae_amd')
import torch, bench
dev = torch.device('cuda:0')
model, _ = bench.build_model(dev)
for (B, H, W) in [(1, 2048, 3072), (16, 512, 768), (2, 64, 64), (3, 832, 1216)]:
    ims = bench.synth_batch(B, H, W, 0).to(dev)
    t0 = time.time(); s = model.compress_batch(ims); torch.cuda.synchronize(); t1 = time.time()
    x = model.decompress_batch(s); torch.cuda.synchronize(); t2 = time.time()
    xe, _ = model.estimate(ims)
    same = bool(torch.equal(x, xe))
    print(B, H, W, 'bytes', sum(len(a) for a in s), f'enc {t1-t0:.3f}s dec {t2-t1:.3f}s', 'dec==estimate', same, 'psnr', float(-10*torch.log10((x-ims).square().mean())))
    assert same
