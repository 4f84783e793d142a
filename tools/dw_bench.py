"""Depthwise+LN kernel timings on model shapes (fp32 and bf16 storage).  python tools/dw_bench.py [reps]"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import torch
from lvae import _native
if os.environ.get('LVAE_LIB'):
    _native.LIB_PATH = os.path.abspath(os.environ['LVAE_LIB'])
L = _native.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shapes = [(8, 128, 192, 192, 1), (8, 128, 192, 192, 3), (8, 128, 192, 192, 5), (8, 128, 192, 192, 7), (8, 128, 192, 128, 7), (8, 64, 96, 384, 7), (8, 64, 96, 256, 7), (8, 32, 48, 384, 5), (8, 32, 48, 512, 5),
          (8, 16, 24, 512, 3), (1, 128, 192, 192, 7), (1, 64, 96, 384, 7)]
only = os.environ.get('DW_ONLY')
if only:
    shapes = [shapes[int(only)]]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (B, H, W, C, k) in shapes:
    x = torch.randn(B, H, W, C, device='cuda')
    wp = torch.randn(k * k, C, device='cuda') / k
    b = torch.randn(C, device='cuda'); sh = torch.randn(C, device='cuda'); sc = 1 + 0.3 * torch.randn(C, device='cuda')
    y = torch.empty_like(x)
    xb, yb = x.to(torch.bfloat16), torch.empty(B, H, W, C, device='cuda', dtype=torch.bfloat16)
    res = []
    for name, fn, xi, yo, esz in (('f32', L.lvae_dwconv_ln_f32, x, y, 4), ('bf16', L.lvae_dwconv_ln_bf16, xb, yb, 2)):
        for _ in range(3):
            fn(xi.data_ptr(), wp.data_ptr(), b.data_ptr(), None, None, sh.data_ptr(), sc.data_ptr(), yo.data_ptr(), B, H, W, C, k, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            rc = fn(xi.data_ptr(), wp.data_ptr(), b.data_ptr(), None, None, sh.data_ptr(), sc.data_ptr(), yo.data_ptr(), B, H, W, C, k, st)
        e1.record(); torch.cuda.synchronize()
        assert rc == 0
        us = e0.elapsed_time(e1) * 1e3 / reps
        res.append(f'{name} {us:7.1f} us {2 * x.numel() * esz / us / 1e6:5.2f} TB/s')
    print(f'B={B} {H}x{W} C={C} k={k}: ' + ' | '.join(res), flush=True)
