"""gemm_q8 (MX-fp8, pre-quantised operands) under each tile on the MLP shapes of config 5 (4 x 1216x1216) and of 8 / 4 x 512x768.
python tools/q8_tiles.py [reps]"""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'lossy-vae_amd'))
import torch
from lvae import _native
from lvae.models.base import pack_mxfp8_q8
L = _native.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
px = {4: 92416, 8: 23104, 16: 5776}          # pixels of one 1216x1216 image per stride
shapes = []
for (B, pxs) in ((4, px), (8, {4: 24576, 8: 6144, 16: 1536}), (4, {4: 24576, 8: 6144, 16: 1536})):
    for s, C, hid in ((4, 192, 384), (8, 384, 768), (16, 512, 1024), (8, 256, 448), (4, 128, 192)):
        if C % 64 or hid % 64:
            continue
        shapes += [(B * pxs[s], hid, C, 1), (B * pxs[s], C, hid, 2)]
for (M, N, K, epi) in shapes:
    A = pack_mxfp8_q8(torch.randn(M, K)).cuda()
    Wt = torch.randn(N, K) / K ** 0.5
    wq = pack_mxfp8_q8(Wt).cuda()
    bias, gamma = torch.randn(N, device='cuda'), torch.rand(N, device='cuda')
    res = torch.randn(M, N, device='cuda').to(torch.bfloat16)
    out = torch.empty(M * N + M * N // 16 + 64, device='cuda', dtype=torch.bfloat16)
    line = f'M={M:7d} N={N:5d} K={K:5d} epi={epi}:'
    for tile in (21, 22, 42, 0):
        d = _native.GemmDesc()
        d.A0, d.lda0, d.K0, d.Wt16, d.ldw = A.data_ptr(), K, K, wq.data_ptr(), K
        d.bias, d.gamma, d.res, d.ldres, d.out, d.ldo = bias.data_ptr(), gamma.data_ptr(), res.data_ptr(), N, out.data_ptr(), N
        d.M, d.N, d.K, d.epi, d.prec, d.a_bf16, d.out_bf16, d.a_h2, d.cfg = M, N, K, epi, 3, 1, 1, 1, tile
        d.out_h2 = 1 if epi == 1 else 0
        for _ in range(3):
            assert L.lvae_gemm_f32(ctypes.byref(d), st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            L.lvae_gemm_f32(ctypes.byref(d), st)
        e1.record(); torch.cuda.synchronize()
        line += f'  {tile if tile else "auto"}: {e0.elapsed_time(e1) * 1e3 / reps:6.1f}'
    print(line, flush=True)
